import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
os.environ["TNQS_DBG_PRE_DUMP_L"] = "1"
from test_gpu_kernels import theta_svd_pre, rnd
def clamp_chol(G, tau=1e-13):
    n=len(G); A=np.tril(G).copy(); dmax=np.real(np.diag(G)).max(); pt=tau*dmax; piv=np.zeros(n)
    for k in range(n):
        d=A[k,k].real; d=d if d>pt else pt; piv[k]=d
        for j in range(k+1,n): A[j:,j]-=A[j:,k]*np.conj(A[j,k])/d
    L=np.tril(A)/np.sqrt(piv)[None,:]
    for k in range(n): L[k,k]=np.sqrt(piv[k])
    return L, piv, pt
for (m, n, nq, rank) in [(128, 64, 128, 20), (128, 64, 128, 64)]:
    rng = np.random.default_rng(m + 7 * rank + n)
    dec = np.exp(-np.arange(rank) * (8.0 / max(rank, 1)))
    q1, _ = np.linalg.qr(rnd(rng, (m, rank), np.complex128)); q2, _ = np.linalg.qr(rnd(rng, (n, rank), np.complex128))
    M = (q1 * dec) @ q2.conj().T
    Q, _ = np.linalg.qr(rnd(rng, (nq, n), np.complex128))
    A, V, sw, _ = theta_svd_pre(M, Q)
    Ldev = V[:n * n // nq + (1 if (n * n) % nq else 0), :].T if False else np.frombuffer(np.asfortranarray(V).tobytes(order="F"), dtype=np.complex64)[: n * n].reshape(n, n, order="F")
    M32 = M.astype(np.complex64).astype(np.complex128)
    fro = np.linalg.norm(M32); kexp = -(int(np.floor(np.log2(fro * fro))) // 2); Ms = M32 * 2.0 ** kexp
    order = np.argsort(-np.linalg.norm(M32, axis=0), kind="stable"); Msort = Ms[:, order]
    Lref, piv, pt = clamp_chol(Msort.conj().T @ Msort)
    bad = ~np.isfinite(Ldev)
    print((m, n, rank), "non-finite entries in device L:", int(bad.sum()), "columns:", np.where(bad.any(axis=0))[0][:12], "rows:", np.where(bad.any(axis=1))[0][:12])
    print("  collapsed pivots (ref):", int((piv <= pt).sum()), " max |Ldev - Lref| over finite:", float(np.max(np.abs(np.where(bad, 0, Ldev - Lref)))), " max|Lref|", float(np.abs(Lref).max()))
    k = int(np.argmax(np.abs(np.where(bad, 0, Ldev - Lref)).max(axis=0)))
    print("  worst column", k, "ref diag", Lref[k, k], "dev diag", Ldev[k, k], "ref col max", float(np.abs(Lref[:, k]).max()), "dev col max", float(np.nanmax(np.abs(Ldev[:, k]))))
    print("  dev diag[18:26]", np.real(np.diag(Ldev))[18:26], "ref", np.real(np.diag(Lref))[18:26])
