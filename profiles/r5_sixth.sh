set -x
O=gpurun_out/r5h; mkdir -p gpurun_out/r5h
python profiles/pre_debug.py 2>&1 | tail -5 > $O/dbg_x8.txt
TNQS_DBG_PRE_QUARTER=1 python profiles/pre_debug.py 2>&1 | tail -5 > $O/dbg_quarter.txt
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "theta_svd" --tb=line 2>&1 | tail -12 > $O/kernels.log
python profiles/svd_bench.py 24 20 > $O/svd_bench_24.txt 2>&1
for L in 7 20; do
  python bench.py --L $L --steps 10 --warmup 3 --no-cpu-baseline > $O/new_L$L.json 2>> $O/err.txt
done
python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "drift or c1_full or c3" 2>&1 | tail -12 > $O/fullsize.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_toggles.py -q -m gpu 2>&1 | tail -8 > $O/parity.log
python profiles/shard_proxy.py --ranks 1,2,4,8 > $O/shard_proxy.txt 2> $O/shard_proxy.err
cat $O/dbg_x8.txt $O/dbg_quarter.txt $O/kernels.log; tail -n 1 $O/svd_bench_24.txt
for f in $O/*.json; do python - <<PY
import json
try:
    d=json.load(open("$f")); print("$f", d["ms_per_step"], d["config"].get("theta_svd_sweeps_per_gate"), d["config"].get("theta_svd_sweeps_slowest_gate"), d["kernel_classes"].get("jacobi"), d["kernel_classes"].get("small"))
except Exception as e: print("$f ERR", e)
PY
done
cat $O/fullsize.log $O/parity.log
grep PROXY $O/shard_proxy.txt | cut -c1-900; tail -n 1 $O/shard_proxy.txt | cut -c1-400; tail -n 5 $O/shard_proxy.err
