set -x
O=gpurun_out/r5d; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "theta_svd" 2>&1 | tail -8 > $O/kernels.log
python scratch/c3_diag.py > $O/c3_diag.txt 2>&1
TNQS_NO_PRECOND_SVD=1 python scratch/c3_diag.py > $O/c3_diag_old.txt 2>&1
python profiles/svd_bench.py 24 20 > $O/svd_bench_24.txt 2>&1
for L in 7 20; do
  python bench.py --L $L --steps 10 --warmup 3 --no-cpu-baseline > $O/new_L$L.json 2>> $O/err.txt
done
python profiles/shard_proxy.py --ranks 1,2,4,8 > $O/shard_proxy.txt 2> $O/shard_proxy.err
cat $O/kernels.log $O/c3_diag.txt $O/c3_diag_old.txt; tail -n 1 $O/svd_bench_24.txt
for f in $O/*.json; do python - <<PY
import json
try:
    d=json.load(open("$f")); print("$f", d["ms_per_step"], d["config"].get("theta_svd_sweeps_per_gate"), d["config"].get("theta_svd_sweeps_slowest_gate"), d["kernel_classes"].get("jacobi"))
except Exception as e: print("$f ERR", e)
PY
done
tail -n 12 $O/shard_proxy.txt; tail -n 5 $O/shard_proxy.err
