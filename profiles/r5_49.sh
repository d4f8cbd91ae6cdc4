O=gpurun_out/r5bp; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time python bench.py ) > $O/bench_default.json 2> $O/time.txt
tail -4 $O/time.txt
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_default.json") if l.startswith("{")][-1]); print(d["ms_per_step"], d["value"], d["steps"], d["roofline"]["frac"], d["roofline"]["from_profile"]["traffic"]["same_build"], d["ab_f32_matrix_instructions"]["ms_per_step"], d["cpu_baseline"]["value"])
PY
