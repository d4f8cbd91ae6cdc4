TNQS_HOST_TIMING=1 NREP=10 python profiles/shape_bench.py heavyhex 2>&1 | grep -v amdgpu.ids | tail -14
