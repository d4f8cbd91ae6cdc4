# final collection of the round on one box: profiles of the default bench command, the bench lines of every BASELINE shape, latency shapes, the sharded proxy
O=gpurun_out/r5final; mkdir -p $O
bash profiles/collect.sh r5 > $O/collect.log 2>&1; tail -1 $O/collect.log
bash profiles/collect_mfma.sh r5 > $O/collect_mfma.log 2>&1; head -3 $O/collect_mfma.log
bash profiles/collect_stalls.sh r5 > $O/collect_stalls.log 2>&1
python bench.py --steps 20 --warmup 3 > $O/bench_c2.json 2>> $O/err.txt
python bench.py --L 7 --steps 10 --warmup 3 --no-cpu-baseline --no-ab > $O/bench_L7.json 2>> $O/err.txt
python bench.py --config c4 --L 3 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c4_L3.json 2>> $O/err.txt
python bench.py --config c4 --L 5 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c4_L5.json 2>> $O/err.txt
python bench.py --config c5 --L 5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c5_L5.json 2>> $O/err.txt
python bench.py --config c5 --L 11 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c5_L11.json 2>> $O/err.txt
for w in heavyhex c1 cubic16 chi64; do NREP=5 python profiles/shape_bench.py $w > $O/shape_$w.json 2>> $O/err.txt; done
python profiles/shard_proxy.py --ranks 1,2,4,8 --steps 3 --warmup 2 > $O/shard_proxy.txt 2>> $O/err.txt
python - <<PY
import json
d=json.load(open("$O/bench_c2.json")); print("c2", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["mfma_busy"], d["roofline"]["from_profile"]["traffic"]["same_build"], d.get("ab_f32_matrix_instructions",{}).get("ms_per_step"), d["phases"]["bp_ms_per_step"], d["phases"]["gate_ms_per_step"])
for f in ("bench_L7","bench_c4_L3","bench_c4_L5","bench_c5_L5","bench_c5_L11"):
    try:
        d=json.load(open("$O/%s.json"%f)); print(f, d["ms_per_step"], d["config"]["memory"].get("measured_peak_GiB"), d["config"].get("bp_partial_products"), {k:(round(v["ms"]/d["steps"],1),v["TFLOPs"]) for k,v in d["kernel_classes"].items()})
    except Exception as e: print(f, "failed", e)
for w in ("heavyhex","c1","cubic16","chi64"):
    try:
        d=json.loads(open("$O/shape_%s.json"%w).read().strip().splitlines()[-1]); print(w, d["ms_per_layer"])
    except Exception as e: print(w, "failed", e)
for l in open("$O/shard_proxy.txt"):
    if l.startswith('PROXY'):
        d=json.loads(l[6:]); print(d['n_ranks'], d['ms_per_layer_by_rank'], d['ms_per_layer_heaviest_rank'], d['ms_per_layer_bsp'])
    elif l.startswith('{'):
        d=json.loads(l); print(d['fit_ms'], d['speedup_bsp_before_communication'], d['speedup_heaviest_rank_before_communication'])
PY
tail -3 $O/err.txt
