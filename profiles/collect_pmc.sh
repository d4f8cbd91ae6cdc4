#!/bin/bash
# One rocprofv3 PMC pass with an arbitrary counter list (kernel trace only), summarised per kernel over its full-batch launches:
#   gpurun --timeout 900 -- 'bash profiles/collect_pmc.sh TAG "SQ_WAIT_ANY SQ_WAIT_INST_ANY ..." ["<command>"]'
# -> gpurun_out/<TAG>_pmc.json.  At most 8 SQ counters per pass (MI355X_MICROARCH.md, "rocprofv3 PMC slots").
set -e
TAG=$1; CTRS=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CMD=${3:-"python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline"}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmc_$TAG
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/pmc_$TAG -- $CMD > $OUT/pmc_$TAG.log 2>&1 || true
python - "$OUT/pmc_$TAG" "$OUT/${TAG}_pmc.json" <<'PY'
import csv, glob, json, re, sys
from collections import defaultdict
d, out = sys.argv[1:3]
f = glob.glob(d + "/*/*counter_collection.csv")
rows = defaultdict(lambda: defaultdict(dict)); dur = defaultdict(dict)
for r in csv.DictReader(open(f[0])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    rows[name][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    if r.get("Start_Timestamp") and r.get("End_Timestamp"):
        dur[name][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for kt in glob.glob(d + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(kt)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        dur[name][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
res = {}
for k, disp in rows.items():
    t = {i: dur[k].get(i, 0.0) for i in disp}
    if not t or max(t.values()) <= 0: continue
    big = [i for i in disp if t[i] > 0.5 * max(t.values())]
    ctrs = sorted({c for i in big for c in disp[i]})
    res[k] = {"launches_full_batch": len(big), "avg_ms": round(sum(t[i] for i in big) / len(big) * 1e-6, 4),
              "total_ms": round(sum(t.values()) * 1e-6, 3), "means": {c: sum(disp[i].get(c, 0.0) for i in big) / len(big) for c in ctrs}}
json.dump({"kernels": dict(sorted(res.items(), key=lambda kv: -kv[1]["total_ms"]))}, open(out, "w"), indent=1)
PY
