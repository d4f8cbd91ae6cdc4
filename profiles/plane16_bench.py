"""Launch timing of the chi = 16 plane kernels on device-resident degree-6 site tensors (2 x 16^6 elements = 268 MB each), per leg pair
(include/tnqs_debug.h, tnqs_dbg_bench_plane which = 2 / 3):   python profiles/plane16_bench.py [nsites] [reps]
prints ms per launch and TB/s (pair16: read + write of the tensor; pair_gram2x16: read of X and Y)."""
import ctypes as C
import itertools
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "..", "tensornetworkquantumsimulator.jl_amd", "libtnqs_hip.so"))
lib.tnqs_dbg_bench_plane.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
nsites = int(sys.argv[1]) if len(sys.argv) > 1 else 12
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pairs = [tuple(int(c) for c in p) for p in sys.argv[3].split(",")] if len(sys.argv) > 3 else list(itertools.combinations(range(6), 2))
n = 2 * 16 ** 6
for which, name in ((2, "pair16"), (3, "pair_gram2x16")):
    for lx, ly in pairs:
        ms = C.c_double(0)
        rc = lib.tnqs_dbg_bench_plane(which, nsites, lx, ly, reps, C.byref(ms))
        if rc != 0:
            print(name, (lx, ly), "rc", rc)
            continue
        print(f"{name:14s} legs ({lx},{ly})  {ms.value:8.3f} ms  {2.0 * n * 8 * nsites / ms.value / 1e9:6.2f} TB/s", flush=True)
