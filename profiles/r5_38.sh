O=gpurun_out/r5ba; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gauge_leg" --tb=short 2>&1 | tail -3
python -m pytest tests/test_gpu_parity.py tests/test_gpu_toggles.py -q -m gpu -x --tb=short 2>&1 | tail -2
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c2.json 2>> $O/err.txt
TNQS_NO_BF16X3=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c2_f32.json 2>> $O/err.txt
python - <<PY
import json
for f in ("bench_c2","bench_c2_f32"):
    d=json.load(open("$O/%s.json"%f)); print(f, d["ms_per_step"], {k:v["ms"] for k,v in d["kernel_classes"].items() if k.startswith("gate")})
PY
