O=gpurun_out/r5cyc; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "random_graphs or default" 2>&1 | tail -8 > $O/parity.log
python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "cubic_4x4x4" 2>&1 | tail -5 > $O/full.log
for m in 0 1; do
  TNQS_NO_CYCLE_SETS=$m python bench.py --config c4 --L 3 --steps 3 --warmup 1 --no-cpu-baseline --no-ab > $O/c4_$m.json 2>> $O/err.txt
done
TNQS_NO_CYCLE_SETS=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ab > $O/c2_0.json 2>> $O/err.txt
cat $O/parity.log $O/full.log
python - <<PY
import json
for f in ("c4_0","c4_1","c2_0"):
    d=json.load(open("$O/%s.json"%f)); print(f, d["ms_per_step"], d["phases"]["bp_ms_per_step"], d["phases"]["gate_ms_per_step"], {k:(round(v["ms"]/d["steps"],1),v["launches"]) for k,v in d["kernel_classes"].items() if k.startswith("bp_")}, d["config"]["bp_partial_products"])
PY
tail -n 5 $O/err.txt
