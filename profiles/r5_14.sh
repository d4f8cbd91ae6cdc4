O=gpurun_out/r5s; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "pair" --tb=short 2>&1 | tail -8
for m in 0 1 2; do echo "== mode $m"; TNQS_X3_MODE=$m python profiles/plane_bench.py 100 5 2>&1 | grep gram2; done | tee $O/modes.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_x3.json 2>> $O/err.txt
python - <<PY
import json
for f in ("bench_x3",):
    try:
        d=json.load(open("$O/%s.json"%f)); print(f, d["ms_per_step"], {k:v["ms"] for k,v in d["kernel_classes"].items()})
    except Exception as e: print(f, "failed", e)
PY
tail -5 $O/err.txt
