O=gpurun_out/r5bf; mkdir -p $O
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --bp-order reference > $O/bench_reforder.json 2>> $O/err.txt
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --evolved 14 > $O/bench_evolved.json 2>> $O/err.txt
python - <<PY
import json
d=json.load(open("$O/bench_reforder.json")); print("reference order", d["ms_per_step"], d["value"])
d=json.load(open("$O/bench_evolved.json")); print("evolved", d["ms_per_step"], d.get("evolved"))
PY
