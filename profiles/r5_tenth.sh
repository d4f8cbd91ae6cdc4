set -x
O=gpurun_out/r5m; mkdir -p gpurun_out/r5m
( time python -m pytest tests -q -m gpu -x --durations=15 ) > $O/suite.log 2>&1
for L in 7 20; do
  python bench.py --L $L --steps 10 --warmup 3 --no-cpu-baseline > $O/new_L$L.json 2>> $O/err.txt
  TNQS_NO_PRECOND_SVD=1 python bench.py --L $L --steps 10 --warmup 3 --no-cpu-baseline > $O/old_L$L.json 2>> $O/err.txt
done
python profiles/shard_proxy.py --ranks 1,2,4,8 > $O/shard_proxy.txt 2> $O/shard_proxy.err
python profiles/shape_bench.py c1 > $O/shape_c1.json 2>> $O/err.txt
python profiles/shape_bench.py heavyhex > $O/shape_hh.json 2>> $O/err.txt
tail -n 30 $O/suite.log
for f in $O/new_*.json $O/old_*.json; do python - <<PY
import json
try:
    d=json.load(open("$f")); print("$f", d["ms_per_step"], d["config"].get("theta_svd_sweeps_per_gate"), d["config"].get("theta_svd_sweeps_slowest_gate"), d["kernel_classes"].get("jacobi"), d["kernel_classes"].get("small"))
except Exception as e: print("$f ERR", e)
PY
done
grep PROXY $O/shard_proxy.txt | python -c "
import sys, json
for ln in sys.stdin:
    d=json.loads(ln[6:]); print(d['n_ranks'], d['partition']['bulk_sites'], d['ms_per_layer_by_rank'], 'bsp', d['ms_per_layer_bsp'], 'maxsw', d['svd_sweeps_slowest_gate'], 'MB', d['MB_gathered_per_layer_per_rank'])
"
tail -n 1 $O/shard_proxy.txt | cut -c1-300
tail -c 600 $O/shape_c1.json; echo; tail -c 600 $O/shape_hh.json
