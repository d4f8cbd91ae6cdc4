"""theta SVD of a colour batch, kernel level: the plain LDS-resident Jacobi on the 128 x 64 low-rank factor against the preconditioned kernel
(kernels.hip theta_svd_pre_kernel) on the same factors -- R factors harvested from oracle runs (tests/golden/theta_factors.npz), `copies` gates per launch.
    python profiles/svd_bench.py [copies] [reps]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import tnqs_amd as tn
lib = C.CDLL(tn.LIB_PATH)
from test_gpu_kernels import low_rank_factors, theta_svd_pre

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
z = np.load(os.path.join(ROOT, "tests", "golden", "theta_factors.npz"))
out = []
for k in sorted({k.rsplit("_", 1)[0] for k in z.files}):
    r1, r2, gate = z[k + "_r1"], z[k + "_r2"], z[k + "_gate"]
    if r1.shape[0] < r2.shape[0]:
        continue
    M, Q, theta = low_rank_factors(r1, r2, gate)
    A = np.asfortranarray(M.astype(np.complex64))
    ms = C.c_double(0.0); sw = C.c_int(0)
    rc = lib.tnqs_dbg_time_jacobi_f32(A.shape[0], A.shape[1], A.ctypes.data_as(C.c_void_p), copies, reps, C.byref(ms), C.byref(sw)); assert rc == 0
    _, _, sw2, ms2 = theta_svd_pre(M, Q, copies=copies, reps=reps)
    out.append({"factor": k, "shape": list(A.shape), "plain_ms": round(ms.value, 4), "plain_sweeps": sw.value, "precond_ms": round(ms2, 4), "precond_sweeps": sw2, "precond_phases_us": theta_svd_pre.phases_us})
    print(out[-1], flush=True)
print(json.dumps({"copies": copies, "reps": reps, "results": out}))
