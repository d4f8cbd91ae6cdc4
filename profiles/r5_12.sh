set -x
O=gpurun_out/r5p; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "pair" --tb=short 2>&1 | tail -15 > $O/kernels.log
cat $O/kernels.log
python profiles/x3_error.py > $O/err_x3.json 2>> $O/err.txt
TNQS_NO_BF16X3=1 python profiles/x3_error.py > $O/err_f32_3m.json 2>> $O/err.txt
TNQS_NO_BF16X3=1 TNQS_NO_3M=1 python profiles/x3_error.py > $O/err_f32_4m.json 2>> $O/err.txt
python profiles/plane_bench.py 100 5 > $O/plane_x3.txt 2>> $O/err.txt
TNQS_NO_BF16X3=1 python profiles/plane_bench.py 100 5 > $O/plane_f32.txt 2>> $O/err.txt
cat $O/plane_x3.txt $O/plane_f32.txt
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_x3.json 2>> $O/err.txt
TNQS_NO_BF16X3=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_f32.json 2>> $O/err.txt
python - <<PY
import json
for f in ("bench_x3","bench_f32"):
    try:
        d=json.load(open("$O/%s.json"%f)); print(f, d["ms_per_step"], {k:v["ms"] for k,v in d["kernel_classes"].items()})
    except Exception as e: print(f, "failed", e)
for f in ("err_x3","err_f32_3m","err_f32_4m"):
    try:
        d=json.load(open("$O/%s.json"%f)); print(f, {k:(("%.1e"%v["max_over_max"]),("%.1e"%v["rms_over_rms"])) for k,v in d.items() if isinstance(v,dict)})
    except Exception as e: print(f, "failed", e)
PY
tail -5 $O/err.txt
