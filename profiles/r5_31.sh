O=gpurun_out/r5ar; mkdir -p $O
for L in 7 20; do python bench.py --L $L --steps 10 --warmup 3 --no-cpu-baseline > $O/b_$L.json 2>> $O/err.txt; done
python - <<PY
import json
for f in ("b_7","b_20"):
    d=json.load(open("$O/%s.json"%f)); print(f, d["ms_per_step"], d["phases"]["bp_ms_per_step"], d["phases"]["gate_ms_per_step"])
PY
