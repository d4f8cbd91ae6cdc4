#!/usr/bin/env python
"""Per-kernel means of the counters of several rocprofv3 PMC passes (profiles/collect_stalls.sh): pmc_generic.py <out.json> <pass dir> ...
For every kernel, over its full-batch launches (duration above half of the kernel's longest launch): the mean of each counter as rocprofv3 reports
it (summed over XCDs / SEs / instances) and the mean launch duration of the pass.  Derived, where the inputs are there:
  waves_per_simd        = SQ_WAVE_CYCLES / (duration cycles * 256 CUs * 4 SIMDs)            resident waves per SIMD, time average
  wait_any_frac         = SQ_WAIT_ANY / SQ_WAVE_CYCLES                                         share of resident-wave cycles spent waiting for anything
  wait_inst_any_frac    = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES                                    ... waiting for an instruction to be issued / fetched
  wait_inst_lds_frac    = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES                                    ... waiting on the LDS instruction path
  active_inst_frac      = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  lds_bank_conflict_frac= SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS                            LDS cycles lost to bank conflicts
  tcp_pending_stall_frac= TCP_PENDING_STALL_CYCLES / (duration cycles * 256)                   vector-L1 cycles stalled on outstanding requests, per CU
  l2_hit_rate           = TCC_HIT / (TCC_HIT + TCC_MISS)
(clock: 2.4 GHz nominal is NOT assumed -- duration cycles = GRBM_GUI_ACTIVE / 8, the per-XCD active count of the same pass)"""
import csv, glob, json, os, re, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from buildid import build_id  # noqa: E402


def load(d):
    f = sorted(glob.glob(d + "/*/*counter_collection.csv"), key=os.path.getmtime, reverse=True)      # (the newest run: a merged scratch directory keeps older ones)
    if not f:
        return {}, {}
    rows = defaultdict(lambda: defaultdict(dict)); dur = defaultdict(dict)
    for r in csv.DictReader(open(f[0])):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        rows[name][r["Dispatch_Id"]][r["Counter_Name"]] = rows[name][r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for kt in [f[0].replace("counter_collection", "kernel_trace")]:
        for r in csv.DictReader(open(kt)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            dur[name][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return rows, dur


def main():
    out = sys.argv[1]; res = defaultdict(dict)
    for d in sys.argv[2:]:
        rows, dur = load(d)
        for k, disp in rows.items():
            if not any(t in k for t in ("mfma_", "x3_", "rowgemm", "jacobi_lds", "chol_kernel", "theta_svd_pre")):
                continue
            ds = {i: dur.get(k, {}).get(i, 0.0) for i in disp}
            if not ds or max(ds.values()) <= 0:
                continue
            big = [i for i, t in ds.items() if t > 0.5 * max(ds.values())]
            cnames = sorted({c for i in big for c in disp[i]})
            m = {c: sum(disp[i].get(c, 0.0) for i in big) / len(big) for c in cnames}
            res[k].setdefault("avg_ms_by_pass", {})[os.path.basename(d)] = round(sum(ds[i] for i in big) / len(big) * 1e-6, 4)
            res[k].setdefault("launches_full_batch", len(big))
            res[k].setdefault("counters", {}).update({c: round(v, 1) for c, v in m.items()})
            gui = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
            der = res[k].setdefault("derived", {})
            wc = m.get("SQ_WAVE_CYCLES")
            if wc and gui: der["waves_per_simd"] = round(wc / (gui * 256 * 4), 3)
            for num, den, name in (("SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "wait_inst_any_frac"),
                                   ("SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES", "active_inst_frac"), ("SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "lds_bank_conflict_frac")):
                if m.get(num) is not None and m.get(den): der[name] = round(m[num] / m[den], 4)
            if m.get("SQ_WAIT_INST_LDS") is not None and gui: der["wait_inst_lds_per_simd_cycle"] = round(m["SQ_WAIT_INST_LDS"] / (gui * 256 * 4), 4)
            if m.get("SQ_ACTIVE_INST_LDS") is not None and gui: der["lds_active_per_cu_cycle"] = round(m["SQ_ACTIVE_INST_LDS"] / (gui * 256), 4)
            if m.get("SQ_ACTIVE_INST_VMEM") is not None and gui: der["vmem_active_per_simd_cycle"] = round(m["SQ_ACTIVE_INST_VMEM"] / (gui * 256 * 4), 4)
            if m.get("SQ_ACTIVE_INST_VALU") is not None and gui: der["valu_active_per_simd_cycle"] = round(m["SQ_ACTIVE_INST_VALU"] / (gui * 256 * 4), 4)
            if m.get("TCP_PENDING_STALL_CYCLES") is not None and gui: der["tcp_pending_stall_frac"] = round(m["TCP_PENDING_STALL_CYCLES"] / (gui * 256), 4)
            if m.get("TCC_HIT") is not None and (m.get("TCC_HIT", 0) + m.get("TCC_MISS", 0)) > 0: der["l2_hit_rate"] = round(m["TCC_HIT"] / (m["TCC_HIT"] + m["TCC_MISS"]), 4)
    json.dump({"build_id": build_id(), "command": os.environ.get("PROFILE_CMD", "bench.py --steps 1 --warmup 1 --no-cpu-baseline"), "kernels": res}, open(out, "w"), indent=1)
    for k, v in res.items():
        print(k[:60], v.get("avg_ms_by_pass"), v.get("derived"))


if __name__ == "__main__":
    main()
