import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from test_gpu_kernels import theta_svd_pre, rnd
for (m, n, nq, rank) in [(128, 64, 128, 20), (128, 64, 128, 64), (128, 40, 100, 10), (64, 64, 64, 40)]:
    rng = np.random.default_rng(m + 7 * rank + n)
    dec = np.exp(-np.arange(rank) * (8.0 / max(rank, 1)))
    q1, _ = np.linalg.qr(rnd(rng, (m, rank), np.complex128)); q2, _ = np.linalg.qr(rnd(rng, (n, rank), np.complex128))
    M = (q1 * dec) @ q2.conj().T
    Q, _ = np.linalg.qr(rnd(rng, (nq, n), np.complex128))
    A, V, sw, _ = theta_svd_pre(M, Q)
    nrm = np.linalg.norm(A.astype(np.complex128), axis=0)
    sref = np.linalg.svd(M.astype(np.complex64).astype(np.complex128), compute_uv=False)
    print((m, n, nq, rank), "variant", "quarter" if os.environ.get("TNQS_DBG_PRE_QUARTER") else "x8", "sweeps", sw, "NaN cols A", int(np.isnan(nrm).sum()), "NaN cols V", int(np.isnan(V).any(axis=0).sum()),
          "sv err", float(np.nanmax(np.abs(np.sort(np.nan_to_num(nrm))[::-1] - sref)) / sref[0]), "phases", theta_svd_pre.phases_us, flush=True)
