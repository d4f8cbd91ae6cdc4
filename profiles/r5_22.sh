O=gpurun_out/r5ae; mkdir -p $O
python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "cubic or c4" --tb=short 2>&1 | tail -4
python bench.py --config c4 --L 3 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c4_L3.json 2>> $O/err.txt
python - <<PY
import json
d=json.load(open("$O/bench_c4_L3.json")); print("c4 L3", d["ms_per_step"], {k:(v["ms"],v["launches"]) for k,v in d["kernel_classes"].items()})
PY
bash profiles/collect.sh r5 > $O/collect.log 2>&1; tail -3 $O/collect.log
bash profiles/collect_mfma.sh r5 > $O/collect_mfma.log 2>&1; tail -12 $O/collect_mfma.log
python bench.py --steps 20 --warmup 3 > $O/bench_c2.json 2>> $O/err.txt
python - <<PY
import json
d=json.load(open("$O/bench_c2.json")); print("c2", d["ms_per_step"], d["value"], d["roofline"], d.get("cpu_baseline"))
PY
tail -3 $O/err.txt
