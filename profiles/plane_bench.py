"""Launch timing of the chi = 32 plane kernels on device-resident site tensors (include/tnqs_debug.h, tnqs_dbg_bench_plane):
    python profiles/plane_bench.py [nsites] [reps]
prints ms per launch and algorithmic TFLOP/s (8 flop per complex multiply-add) per kernel and leg pair."""
import ctypes as C
import os
import sys

here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "..", "tensornetworkquantumsimulator.jl_amd", "libtnqs_hip.so"))
lib.tnqs_dbg_bench_plane.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
nsites = int(sys.argv[1]) if len(sys.argv) > 1 else 100
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = 2 * 32 ** 4
for which, name, flops in ((0, "pair", 2 * 8.0 * n * 32), (1, "pair_gram2", 4 * 8.0 * n * 32)):
    for lx, ly in ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)):
        ms = C.c_double(0)
        rc = lib.tnqs_dbg_bench_plane(which, nsites, lx, ly, reps, C.byref(ms))
        if rc != 0:
            print(name, (lx, ly), "rc", rc)
            continue
        print(f"{name:11s} legs ({lx},{ly})  {ms.value:8.3f} ms  {flops * nsites / ms.value / 1e9:7.1f} TFLOP/s")
