O=gpurun_out/r5y; mkdir -p $O
python -m pytest tests -q -m gpu -x --tb=short 2>&1 | tail -15 > $O/suite.log
cat $O/suite.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_x3.json 2>> $O/err.txt
TNQS_NO_BP_SPLIT=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_x3_nosplit.json 2>> $O/err.txt
python - <<PY
import json
for f in ("bench_x3","bench_x3_nosplit"):
    d=json.load(open("$O/%s.json"%f)); print(f, d["ms_per_step"], {k:(v["ms"],v["launches"]) for k,v in d["kernel_classes"].items()})
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof -o out --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /tmp/prof.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob("/tmp/prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
cp /tmp/prof/*/*kernel_stats.csv $GRAFT_REPO_ROOT/$O/ 2>/dev/null || find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/$O/kernel_stats.csv \;
