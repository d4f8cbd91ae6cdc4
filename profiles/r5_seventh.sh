set -x
O=gpurun_out/r5j; mkdir -p gpurun_out/r5j
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "theta_svd" --tb=short 2>&1 | tail -30 > $O/kernels.log
python profiles/svd_bench.py 24 20 > $O/svd_bench_24.txt 2>&1
TNQS_DEBUG_SWEEPS=1 python bench.py --L 7 --steps 1 --warmup 3 --no-cpu-baseline 2>&1 | grep "tnqs sweeps" | awk '{print $5,$6,$7,$8,$9,$10,$11,$12,$13,$14,$15,$16,$17,$18,$19}' | sort | uniq -c | sort -k1 -n -r | head -20 > $O/sweeps_L7.txt
for L in 7 20; do
  python bench.py --L $L --steps 10 --warmup 3 --no-cpu-baseline > $O/new_L$L.json 2>> $O/err.txt
  TNQS_NO_PRECOND_SVD=1 python bench.py --L $L --steps 10 --warmup 3 --no-cpu-baseline > $O/old_L$L.json 2>> $O/err.txt
done
python bench.py --bp-order reference --steps 10 --warmup 3 --no-cpu-baseline > $O/new_L20_reforder.json 2>> $O/err.txt
python -m pytest tests/test_gpu_fullsize.py -q -m gpu 2>&1 | tail -12 > $O/fullsize.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_toggles.py tests/test_gpu_sharded.py -q -m gpu 2>&1 | tail -8 > $O/parity.log
python profiles/shard_proxy.py --ranks 1,2,4,8 > $O/shard_proxy.txt 2> $O/shard_proxy.err
cat $O/kernels.log | tail -15; tail -n 1 $O/svd_bench_24.txt | cut -c1-1500; cat $O/sweeps_L7.txt
for f in $O/*.json; do python - <<PY
import json
try:
    d=json.load(open("$f")); print("$f", d["ms_per_step"], d["config"].get("bp_sweeps_per_step"), d["config"].get("theta_svd_sweeps_per_gate"), d["config"].get("theta_svd_sweeps_slowest_gate"), d["kernel_classes"].get("jacobi"), d["kernel_classes"].get("small"))
except Exception as e: print("$f ERR", e)
PY
done
cat $O/fullsize.log $O/parity.log
grep PROXY $O/shard_proxy.txt | cut -c1-900; tail -n 1 $O/shard_proxy.txt | cut -c1-400; tail -n 5 $O/shard_proxy.err
