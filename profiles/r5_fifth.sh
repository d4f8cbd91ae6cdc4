set -x
O=gpurun_out/r5e; mkdir -p gpurun_out/r5e
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "theta_svd" --tb=line 2>&1 | tail -12 > $O/kernels.log
python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "theta_svd_pre" 2>&1 | grep -E "^E  |assert|passed|failed" | head -20 > $O/kernels_first_failure.log
python profiles/svd_bench.py 24 20 > $O/svd_bench_24.txt 2>&1
for L in 7 20; do
  python bench.py --L $L --steps 10 --warmup 3 --no-cpu-baseline > $O/new_L$L.json 2>> $O/err.txt
done
python profiles/shard_proxy.py --ranks 1,2,4,8 > $O/shard_proxy.txt 2> $O/shard_proxy.err
cat $O/kernels.log; cat $O/kernels_first_failure.log | cut -c1-400; tail -n 1 $O/svd_bench_24.txt
for f in $O/*.json; do python - <<PY
import json
try:
    d=json.load(open("$f")); print("$f", d["ms_per_step"], d["config"].get("theta_svd_sweeps_per_gate"), d["config"].get("theta_svd_sweeps_slowest_gate"), d["kernel_classes"].get("jacobi"), d["kernel_classes"].get("small"))
except Exception as e: print("$f ERR", e)
PY
done
grep PROXY $O/shard_proxy.txt | cut -c1-900; tail -n 1 $O/shard_proxy.txt | cut -c1-400; tail -n 5 $O/shard_proxy.err
