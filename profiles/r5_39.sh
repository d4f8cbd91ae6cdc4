O=gpurun_out/r5bc; mkdir -p $O
bash profiles/collect.sh r5 > $O/collect.log 2>&1; tail -1 $O/collect.log
bash profiles/collect_mfma.sh r5 > $O/collect_mfma.log 2>&1; head -6 $O/collect_mfma.log
bash profiles/collect_stalls.sh r5 > $O/collect_stalls.log 2>&1
python bench.py --steps 20 --warmup 3 > $O/bench_c2.json 2>> $O/err.txt
python bench.py --config c4 --L 3 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c4_L3.json 2>> $O/err.txt
python bench.py --config c5 --L 5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c5_L5.json 2>> $O/err.txt
python bench.py --L 7 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_L7.json 2>> $O/err.txt
for m in heavyhex c1 c128; do NREP=5 python profiles/shape_bench.py $m > $O/shape_$m.json 2>> $O/err.txt; done
python - <<PY
import json
d=json.load(open("$O/bench_c2.json")); print("c2", d["ms_per_step"], d["value"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["mfma_busy"], d["roofline"]["from_profile"]["traffic"]["same_build"], d.get("ab_f32_matrix_instructions",{}).get("ms_per_step"), d["phases"]["bp_ms_per_step"], d["phases"]["gate_ms_per_step"])
for f in ("bench_c4_L3","bench_c5_L5","bench_L7"):
    d=json.load(open("$O/%s.json"%f)); print(f, d["ms_per_step"])
for m in ("heavyhex","c1","c128"):
    d=json.load(open("$O/shape_%s.json"%m)); print(m, d["ms_per_layer"])
PY
