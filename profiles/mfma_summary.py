#!/usr/bin/env python
"""Summarise one rocprofv3 PMC pass (profiles/collect_mfma.sh) into MFMA utilisation per kernel.

usage: mfma_summary.py <dir of the pass> <out.json>

Per kernel, over its full-batch launches (GRBM_GUI_ACTIVE above 50 % of the kernel's maximum):
  mfma_busy      = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)   -- the gfx94x `MfmaUtil` formula (rocprofv3 falls back to
                   it on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots") with GRBM_GUI_ACTIVE divided by the 8 XCDs: the value rocprofv3
                   reports is the SUM over the XCDs' GRBMs (checked on mfma_pair_gram2_kernel: 111.9 M "active" for a 6.32 ms launch = 8 x 13.99 M
                   cycles at 2.21 GHz).  1.0 = every matrix core issuing back to back for the whole kernel.  Cross-check: BUSY_CYCLES equals
                   8 cycles x MOPS exactly for the f32 32x32x2 kernels, i.e. the 64-cycle issue of an instruction that is 8 MOPS
  mfma_flops     = 512 * (SQ_INSTS_VALU_MFMA_MOPS_F32 + _F64 + _BF16) per launch (one MOP = 512 flop; round 5: the bf16 x 3 kernels count under _BF16), to be compared with the
                   algorithmic flops the engine accounts for the kernel's class (bench.py "kernel_classes")
  mfma_tflops    = mfma_flops / kernel duration from the kernel trace of the same pass
The raw counter means are kept next to the derived values."""
import csv
import glob
import json
import re
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from buildid import build_id  # noqa: E402

NCU, NSIMD, NXCD = 256, 4, 8


def main():
    d, out = sys.argv[1:3]
    f = glob.glob(d + "/*/*counter_collection.csv")
    if not f:
        raise SystemExit(f"no counter_collection.csv under {d}")
    rows = defaultdict(lambda: defaultdict(dict))          # kernel -> dispatch id -> counter -> value
    dur = defaultdict(dict)
    for r in csv.DictReader(open(f[0])):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        rows[name][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        if "Start_Timestamp" in r and "End_Timestamp" in r:
            dur[name][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    kt = glob.glob(d + "/*/*kernel_trace.csv")
    if kt:
        for r in csv.DictReader(open(kt[0])):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            dur[name][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    res = {}
    for k, disp in rows.items():
        act = {i: c.get("GRBM_GUI_ACTIVE", 0.0) for i, c in disp.items()}
        if not act or max(act.values()) <= 0:
            continue
        big = [i for i, a in act.items() if a > 0.5 * max(act.values())]
        mean = lambda name: sum(disp[i].get(name, 0.0) for i in big) / len(big)
        gui, busy = mean("GRBM_GUI_ACTIVE"), mean("SQ_VALU_MFMA_BUSY_CYCLES")
        mops = mean("SQ_INSTS_VALU_MFMA_MOPS_F32") + mean("SQ_INSTS_VALU_MFMA_MOPS_F64") + mean("SQ_INSTS_VALU_MFMA_MOPS_BF16")
        ns = [dur[k][i] for i in big if i in dur.get(k, {})]
        t = sum(ns) / len(ns) * 1e-9 if ns else None
        res[k] = {"launches_full_batch": len(big), "mfma_busy": round(busy / (gui / NXCD * NCU * NSIMD), 4) if gui > 0 else None,
                  "mfma_flops_per_launch": 512.0 * mops, "mfma_tflops": round(512.0 * mops / t / 1e12, 2) if t else None,
                  "avg_ms": round(t * 1e3, 4) if t else None,
                  "clock_GHz": round(gui / NXCD / (t * 1e9), 3) if t else None,       # GRBM_GUI_ACTIVE of one XCD / duration: the clock the kernel ran at
                  "raw_means": {c: mean(c) for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32",
                                                     "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE")}}
    doc = {"command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_BF16 "
                      "SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- " + os.environ.get("PROFILE_CMD", "<command not recorded>") + " (one pass; profiles/collect_mfma.sh)",
           "build_id": build_id(),
           "formulas": "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 * 4); mfma_flops = 512 * MOPS; full-batch launches only",
           "kernels": res}
    json.dump(doc, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -(kv[1]["mfma_flops_per_launch"] or 0))[:14]:
        print(f"busy {v['mfma_busy']}  {v['mfma_tflops']} TFLOP/s executed  {v['avg_ms']} ms  ({v['launches_full_batch']:4d} launches)  {k}")


if __name__ == "__main__":
    main()
