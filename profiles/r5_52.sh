O=gpurun_out/r5f32; mkdir -p $O
TNQS_NO_BF16X3=1 python -m pytest tests -q -m gpu -x --tb=short > $O/suite_f32_full.log 2>&1
grep -E "passed|failed|error" $O/suite_f32_full.log | tail -3 > $O/suite_f32.log; cat $O/suite_f32.log
python bench.py --bp-order reference --steps 5 --warmup 2 --no-cpu-baseline --no-ab > $O/bench_reference_order.json 2>> $O/err.txt
python bench.py --evolved 14 --steps 5 --warmup 2 --no-cpu-baseline --no-ab > $O/bench_evolved.json 2>> $O/err.txt
python - <<PY
import json
for f in ("bench_reference_order","bench_evolved"):
    try:
        d=json.load(open("$O/%s.json"%f)); print(f, d["ms_per_step"], d["value"], d["config"]["bp_sweeps_per_step"])
    except Exception as e: print(f, "failed", e)
PY
