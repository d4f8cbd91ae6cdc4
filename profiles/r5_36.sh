O=gpurun_out/r5ay; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -q -m gpu -x --tb=short 2>&1 | tail -3
for L in 7 20; do python bench.py --L $L --steps 10 --warmup 3 --no-cpu-baseline > $O/b_$L.json 2>> $O/err.txt; done
python - <<PY
import json
for f in ("b_7","b_20"):
    d=json.load(open("$O/%s.json"%f)); print(f, d["ms_per_step"], d["phases"]["bp_ms_per_step"], d["phases"]["gate_ms_per_step"])
PY
NREP=10 python profiles/shape_bench.py heavyhex | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('heavyhex', d['ms_per_layer'])"
