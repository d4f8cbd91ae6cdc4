O=gpurun_out/r5bh; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "pair_gram" --tb=short 2>&1 | tail -3
python -m pytest tests/test_gpu_parity.py -q -m gpu -x --tb=short 2>&1 | tail -2
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --bp-order reference > $O/bench_reforder.json 2>> $O/err.txt
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2>> $O/err.txt
python - <<PY
import json
d=json.load(open("$O/bench_reforder.json")); print("reference order", d["ms_per_step"], d["value"])
d=json.load(open("$O/bench.json")); print("library order", d["ms_per_step"], d["value"])
PY
