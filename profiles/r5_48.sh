O=gpurun_out/r5bn; mkdir -p $O
python profiles/shard_proxy.py --ranks 1,2,4,8 --steps 3 --warmup 2 > $O/shard_proxy.txt 2>> $O/err.txt
python - <<PY
import json
for l in open("$O/shard_proxy.txt"):
    if l.startswith('PROXY'):
        d=json.loads(l[6:]); print(d['n_ranks'], d['ms_per_layer_by_rank'], d['ms_per_layer_heaviest_rank'], d['ms_per_layer_bsp'])
    elif l.startswith('{'):
        d=json.loads(l); print(d['fit_ms'], d['speedup_bsp_before_communication'], d['speedup_heaviest_rank_before_communication'])
PY
