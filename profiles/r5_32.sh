O=gpurun_out/r5at; mkdir -p $O
bash profiles/collect.sh r5 > $O/collect.log 2>&1; tail -2 $O/collect.log
bash profiles/collect_mfma.sh r5 > $O/collect_mfma.log 2>&1; head -4 $O/collect_mfma.log
bash profiles/collect_stalls.sh r5 > $O/collect_stalls.log 2>&1; tail -3 $O/collect_stalls.log
python bench.py --steps 20 --warmup 3 > $O/bench_c2.json 2>> $O/err.txt
python - <<PY
import json
d=json.load(open("$O/bench_c2.json")); print("c2", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["mfma_busy"], d["roofline"]["from_profile"], d.get("ab_f32_matrix_instructions"))
PY
for m in heavyhex c1 chi64 cubic16; do NREP=5 python profiles/shape_bench.py $m > $O/shape_$m.json 2>> $O/err.txt; done
python bench.py --L 7 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_L7.json 2>> $O/err.txt
python - <<PY
import json
for m in ("heavyhex","c1","chi64","cubic16"):
    d=json.load(open("$O/shape_%s.json"%m)); print(m, d["ms_per_layer"])
d=json.load(open("$O/bench_L7.json")); print("L7", d["ms_per_step"])
PY
