O=gpurun_out/r5g16; mkdir -p $O
timeout 300 python profiles/plane16_bench.py 12 3 01,12,23,35,45 > $O/plane16.txt 2>&1; cat $O/plane16.txt | tail -12
timeout 600 python -m pytest tests/test_gpu_toggles.py -q -m gpu -k "chi16_plane" 2>&1 | tail -4
timeout 300 python bench.py --config c4 --L 3 --steps 3 --warmup 1 --no-cpu-baseline --no-ab > $O/c4_L3.json 2>> $O/err.txt
python - <<PY
import json
d=json.load(open("$O/c4_L3.json")); print("c4 L3", d["ms_per_step"], {k:(round(v["ms"]/d["steps"],1),v["TFLOPs"]) for k,v in d["kernel_classes"].items() if k.startswith("bp_")})
PY
