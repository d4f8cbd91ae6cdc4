O=gpurun_out/r5cache; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "random_graphs or default or periodic or chi16 or degree6 or cubic" 2>&1 | tail -8 > $O/parity.log
python bench.py --config c4 --L 3 --steps 3 --warmup 1 --no-cpu-baseline --no-ab > $O/c4_L3.json 2>> $O/err.txt
python bench.py --config c4 --L 5 --steps 2 --warmup 1 --no-cpu-baseline --no-ab > $O/c4_L5.json 2>> $O/err.txt
cat $O/parity.log
python - <<PY
import json
for f in ("c4_L3","c4_L5"):
    d=json.load(open("$O/%s.json"%f)); print(f, d["ms_per_step"], d["phases"]["bp_ms_per_step"], d["phases"]["gate_ms_per_step"], {k:(round(v["ms"]/d["steps"],1),v["launches"]) for k,v in d["kernel_classes"].items() if k.startswith("bp_")}, d["config"]["bp_partial_products"], d["config"]["memory"])
PY
