set -x
O=gpurun_out/r5o; mkdir -p gpurun_out/r5o
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_toggles.py tests/test_gpu_sharded.py -q -m gpu --tb=line 2>&1 | tail -12 > $O/parity.log
python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "c3 or c1 or heavy" --tb=short 2>&1 | tail -12 > $O/fullsize.log
for m in heavyhex c1; do
  NREP=10 python profiles/shape_bench.py $m > $O/shape_$m.json 2>> $O/err.txt
  NREP=10 TNQS_NO_SMALL_SITE_BP=1 python profiles/shape_bench.py $m > $O/shape_${m}_nosmall.json 2>> $O/err.txt
done
python bench.py --L 7 --steps 10 --warmup 3 --no-cpu-baseline > $O/new_L7.json 2>> $O/err.txt
TNQS_NO_SMALL_SITE_BP=1 python bench.py --L 7 --steps 10 --warmup 3 --no-cpu-baseline > $O/nosmall_L7.json 2>> $O/err.txt
cat $O/parity.log $O/fullsize.log
for f in $O/shape_*.json; do python - <<PY
import json
d=json.load(open("$f")); print("$f", d["ms_per_layer"], {k:(v["ms"],v["launches"]) for k,v in d["classes"].items()})
PY
done
for f in $O/new_L7.json $O/nosmall_L7.json; do python - <<PY
import json
d=json.load(open("$f")); print("$f", d["ms_per_step"], d["kernel_classes"])
PY
done
