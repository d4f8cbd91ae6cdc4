#!/usr/bin/env python
"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) into HBM bytes per launch per kernel.

usage: pmc_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json>

Corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section, re-calibrated in every run on permute_kernel = a pure copy
of known size): the counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a coalesced stream (x2),
WRITE_SIZE is exact (x1).  Per-launch values are means over the FULL-BATCH launches of a kernel (launches whose counter
is above 50 % of that kernel's maximum), so that the small boundary-site launches do not dilute them."""
import csv
import glob
import json
import re
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from buildid import build_id  # noqa: E402


def load(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")
    if not f:
        raise SystemExit(f"no counter_collection.csv under {d}")
    per = defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name") != counter:
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        per[name].append(float(r["Counter_Value"]))
    return per


def main():
    fd, wd, out = sys.argv[1:4]
    fetch, write = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    cal = {}
    if "tnqs::permute_kernel<float>" in fetch:
        big = [v for v in fetch["tnqs::permute_kernel<float>"] if v > 0.5 * max(fetch["tnqs::permute_kernel<float>"])]
        cal["permute_fetch_KiB_mean"] = sum(big) / len(big)
    res = {}
    for k in sorted(set(fetch) | set(write)):
        fv, wv = fetch.get(k, []), write.get(k, [])
        fb = [v for v in fv if v > 0.5 * max(fv)] if fv and max(fv) > 0 else []
        wb = [v for v in wv if v > 0.5 * max(wv)] if wv and max(wv) > 0 else []
        rd = 2.0 * 1024.0 * (sum(fb) / len(fb)) if fb else 0.0
        wr = 1024.0 * (sum(wb) / len(wb)) if wb else 0.0
        res[k] = {"launches_full_batch": max(len(fb), len(wb)), "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                  "hbm_bytes_per_launch": rd + wr}
    doc = {"command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} --output-format csv -- "
                      + os.environ.get("PROFILE_CMD", "<command not recorded>") + " (two separate passes; profiles/collect.sh)",
           "build_id": build_id(),
           "corrections": "FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE x2 on gfx950 (calibration: permute_kernel copies 16384 KiB per launch of a "
                          "bulk site tensor, see 'calibration'); WRITE_SIZE x1.  Means over the full-batch launches (> 50 % of the kernel's maximum).",
           "calibration": cal, "kernels": res}
    json.dump(doc, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
        print(f"{v['hbm_bytes_per_launch'] / 1e9:9.3f} GB/launch  ({v['launches_full_batch']:4d} launches)  {k}")


if __name__ == "__main__":
    main()
