O=gpurun_out/r5aj; mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/err.txt
python - <<PY
import json
d=json.load(open("$O/bench_default.json")); print(d["ms_per_step"], d["value"], d.get("ab_f32_matrix_instructions"), d["phases"])
PY
python profiles/shard_proxy.py --ranks 1,2,4,8 --steps 3 --warmup 2 > $O/shard_proxy.txt 2>> $O/err.txt
tail -1 $O/shard_proxy.txt | cut -c1-600
grep -c PROXY $O/shard_proxy.txt
