set -x
O=gpurun_out/r5c; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "theta_svd or jacobi" 2>&1 | tail -25 > $O/kernels.log
python profiles/svd_bench.py 24 20 > $O/svd_bench_24.txt 2>&1
for L in 7 20; do
  python bench.py --L $L --steps 10 --warmup 3 --no-cpu-baseline > $O/new_L$L.json 2>> $O/err.txt
  TNQS_NO_PRECOND_SVD=1 python bench.py --L $L --steps 10 --warmup 3 --no-cpu-baseline > $O/old_L$L.json 2>> $O/err.txt
done
python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "drift or c1_full or c3" 2>&1 | tail -25 > $O/fullsize.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -q -m gpu 2>&1 | tail -15 > $O/parity.log
python -m pytest tests/test_gpu_sharded.py -q -m gpu 2>&1 | tail -15 > $O/sharded.log
cat $O/kernels.log; tail -n 1 $O/svd_bench_24.txt
for f in $O/*.json; do python - <<PY
import json
try:
    d=json.load(open("$f")); print("$f", d["ms_per_step"], d["config"].get("theta_svd_sweeps_per_gate"), d["kernel_classes"].get("jacobi"))
except Exception as e: print("$f ERR", e)
PY
done
cat $O/fullsize.log $O/parity.log $O/sharded.log
