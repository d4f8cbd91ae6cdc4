O=gpurun_out/r5bi; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -m gpu -x --tb=short -k "bp or message or layer" 2>&1 | tail -2
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --bp-order reference > $O/bench_reforder.json 2>> $O/err.txt
python - <<PY
import json
d=json.load(open("$O/bench_reforder.json")); print("reference order", d["ms_per_step"], d["value"], d["phases"])
PY
