O=gpurun_out/r5ax; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "test_gram or fiber_gemm" --tb=short 2>&1 | tail -4
python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "chi64 or c5" --tb=short 2>&1 | tail -3
NREP=5 python profiles/shape_bench.py chi64 > $O/shape_chi64.json 2>> $O/err.txt
NREP=5 TNQS_NO_BF16X3=1 python profiles/shape_bench.py chi64 > $O/shape_chi64_f32.json 2>> $O/err.txt
python - <<PY
import json
for f in ("shape_chi64","shape_chi64_f32"):
    d=json.load(open("$O/%s.json"%f)); print(f, d["ms_per_layer"], {k:(v["ms"],v.get("tflops")) for k,v in d["classes"].items() if not k.startswith("phase")})
PY
