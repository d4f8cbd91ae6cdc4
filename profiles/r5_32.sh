O=gpurun_out/r5mix; mkdir -p $O
TNQS_X3_MIX=1 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "double_pair or bf16x3 or pair_gram" 2>&1 | tail -8 > $O/kernels.log
for m in 0 1; do
  TNQS_X3_MIX=$m python profiles/plane_bench.py 100 5 > $O/plane_$m.txt 2>&1
  TNQS_X3_MIX=$m python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ab > $O/b_$m.json 2>> $O/err.txt
done
TNQS_X3_MIX=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ab > $O/b_1b.json 2>> $O/err.txt
TNQS_X3_MIX=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ab > $O/b_0b.json 2>> $O/err.txt
cat $O/kernels.log; tail -n 4 $O/plane_0.txt $O/plane_1.txt
python - <<PY
import json
for f in ("b_0","b_1","b_0b","b_1b"):
    d=json.load(open("$O/%s.json"%f)); print(f, d["ms_per_step"], d["phases"]["bp_ms_per_step"], d["phases"]["gate_ms_per_step"], d["kernel_classes"].get("pair_gram"))
PY
