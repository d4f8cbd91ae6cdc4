O=gpurun_out/r5x; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "double_pair" --tb=short 2>&1 | tail -3
for m in 0 1; do echo "== mode $m"; TNQS_X3_MODE=$m python profiles/plane_bench.py 100 5 2>&1 | grep -E "gram2"; done | tee $O/modes.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_x3.json 2>> $O/err.txt
python - <<PY
import json
d=json.load(open("$O/bench_x3.json")); print(d["ms_per_step"], {k:v["ms"] for k,v in d["kernel_classes"].items()})
PY
