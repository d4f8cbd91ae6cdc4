O=gpurun_out/r5an; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "pair or bf16x3" --tb=short 2>&1 | tail -2
for L in 7 20; do
  python bench.py --L $L --steps 10 --warmup 3 --no-cpu-baseline > $O/bal_$L.json 2>> $O/err.txt
  TNQS_NO_SPW_BALANCE=1 python bench.py --L $L --steps 10 --warmup 3 --no-cpu-baseline > $O/nobal_$L.json 2>> $O/err.txt
done
python - <<PY
import json
for f in ("bal_7","nobal_7","bal_20","nobal_20"):
    d=json.load(open("$O/%s.json"%f)); print(f, d["ms_per_step"], {k:v["ms"] for k,v in d["kernel_classes"].items() if k in ("bp_pair","bp_pairgram","gate_modeprod")})
PY
