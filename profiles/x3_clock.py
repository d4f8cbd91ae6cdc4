"""one leg pair of the both-messages pair-Gram, a few launches (for rocprofv3 --pmc GRBM_GUI_ACTIVE: cycles / duration = effective clock)"""
import ctypes as C, os, sys
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "..", "tensornetworkquantumsimulator.jl_amd", "libtnqs_hip.so"))
lib.tnqs_dbg_bench_plane.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
ms = C.c_double(0)
which = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rc = lib.tnqs_dbg_bench_plane(which, 100, 1, 2, 5, C.byref(ms))
print(rc, ms.value)
