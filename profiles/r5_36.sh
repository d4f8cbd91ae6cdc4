O=gpurun_out/r5small; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "small_sites_with_16 or heavy_hex or periodic_lattices" 2>&1 | tail -15 > $O/parity.log
python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "c3" 2>&1 | tail -8 > $O/full.log
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "theta_svd or jacobi" 2>&1 | tail -5 > $O/kern.log
for m in 0 1; do
  TNQS_NO_SMALL_SITE_MFMA=$m NREP=5 python profiles/shape_bench.py heavyhex > $O/hh_$m.json 2>> $O/err.txt
  TNQS_NO_SMALL_SITE_MFMA=$m NREP=5 python profiles/shape_bench.py heavyhex > $O/hh_${m}b.json 2>> $O/err.txt
done
cat $O/parity.log $O/full.log $O/kern.log
python - <<PY
import json
for f in ("hh_0","hh_0b","hh_1","hh_1b"):
    d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["ms_per_layer"], {k:(v["ms"],v["launches"]) for k,v in d["classes"].items() if k in ("bp_fused","jacobi","small","phase_bp_update","phase_gate_batch")})
PY
