"""One Cholesky factorisation + inverse per launch at n = 64 / 96 / 128 through tnqs_dbg_chol (single workgroup: pure latency).  Run under
`rocprofv3 --kernel-trace` and read the kernel durations from the trace."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tnqs_amd as tn
lib = tn._lib.lib
lib.tnqs_dbg_chol.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_double]
for n in (64, 96, 128):
    rng = np.random.default_rng(n)
    a = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    g = np.asfortranarray(a @ a.conj().T + n * np.eye(n))
    L = np.zeros((n, n), dtype=np.complex128, order="F"); W = np.zeros((n, n), dtype=np.complex128, order="F"); fail = C.c_int(-1)
    for _ in range(3):
        lib.tnqs_dbg_chol(n, g.ctypes.data_as(C.c_void_p), L.ctypes.data_as(C.c_void_p), W.ctypes.data_as(C.c_void_p), C.byref(fail), 1e-12)
        lib.tnqs_dbg_chol(n, g.ctypes.data_as(C.c_void_p), L.ctypes.data_as(C.c_void_p), None, C.byref(fail), 1e-12)
