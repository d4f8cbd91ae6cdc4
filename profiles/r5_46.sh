O=gpurun_out/r5bj; mkdir -p $O
bash profiles/collect.sh r5 > $O/collect.log 2>&1; tail -1 $O/collect.log
bash profiles/collect_mfma.sh r5 > $O/collect_mfma.log 2>&1; head -5 $O/collect_mfma.log
bash profiles/collect_stalls.sh r5 > $O/collect_stalls.log 2>&1
python bench.py --steps 20 --warmup 3 > $O/bench_c2.json 2>> $O/err.txt
python bench.py --config c4 --L 5 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c4_L5.json 2>> $O/err.txt
python bench.py --config c5 --L 11 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c5_L11.json 2>> $O/err.txt
python - <<PY
import json
d=json.load(open("$O/bench_c2.json")); print("c2", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["mfma_busy"], d["roofline"]["from_profile"]["traffic"]["same_build"], d.get("ab_f32_matrix_instructions",{}).get("ms_per_step"))
for f in ("bench_c4_L5","bench_c5_L11"):
    try:
        d=json.load(open("$O/%s.json"%f)); print(f, d["ms_per_step"], d["config"]["memory"], {k:(v["ms"],v["TFLOPs"]) for k,v in d["kernel_classes"].items()})
    except Exception as e: print(f, "failed", e)
PY
tail -3 $O/err.txt
