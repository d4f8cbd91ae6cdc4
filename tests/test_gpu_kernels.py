"""Kernel-level GPU tests through include/tnqs_debug.h: each HIP kernel against a float64 numpy evaluation of the
same contraction (transposition-detecting random inputs, odd dimensions, ragged tiles)."""
import ctypes as C
import os

import numpy as np
import pytest

import tnqs_amd as tn

pytestmark = pytest.mark.gpu
lib = C.CDLL(tn.LIB_PATH)
CDT = {0: np.complex64, 1: np.complex128}
EPS = {0: 2e-6, 1: 1e-14}


def rnd(rng, shape, dt):
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(dt)


def jacobi(a, dtype):
    m, n = a.shape
    A = np.asfortranarray(a.astype(CDT[dtype]))
    V = np.zeros((n, n), dtype=CDT[dtype], order="F")
    sw = C.c_int()
    rc = lib.tnqs_dbg_jacobi(dtype, m, n, A.ctypes.data_as(C.c_void_p), V.ctypes.data_as(C.c_void_p), C.byref(sw))
    assert rc == 0, lib.tnqs_last_error()
    return A, V, sw.value


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("shape", [(3, 3), (6, 6), (12, 12), (12, 6), (17, 9), (40, 40), (64, 64), (128, 128), (130, 70), (1, 1), (5, 2), (256, 256)])
def test_jacobi_svd(dtype, shape):
    rng = np.random.default_rng(sum(shape) + dtype)
    a = rnd(rng, shape, np.complex128)
    A, V, sw = jacobi(a, dtype)
    eps = EPS[dtype] * max(shape)
    s_ref = np.linalg.svd(a, compute_uv=False)
    s = np.sort(np.linalg.norm(A, axis=0))[::-1]
    assert sw < 60      # rows >= columns is required (wide matrices are handled through their adjoint)
    assert np.max(np.abs(s[:len(s_ref)] - s_ref)) < eps * s_ref[0], (s[:5], s_ref[:5])
    assert np.max(np.abs(V.conj().T @ V - np.eye(shape[1]))) < 4 * eps                  # V unitary (recovered V: eps*cond)
    assert np.max(np.abs(A @ V.conj().T - a)) < 4 * eps * s_ref[0]                      # A_in = (U S) V^dagger (V may be recovered: eps*cond)
    G = A.conj().T @ A
    off = G - np.diag(np.diag(G))
    assert np.max(np.abs(off)) < eps * s_ref[0] ** 2                                    # columns orthogonal


@pytest.mark.parametrize("n,rank", [(6, 3), (12, 5), (64, 20), (8, 8)])
def test_jacobi_hermitian_psd_rank_deficient(n, rank):
    rng = np.random.default_rng(n)
    b = rnd(rng, (rank, n), np.complex128)
    g = b.conj().T @ b
    A, V, sw = jacobi(g, 1)
    lam = np.real(np.sum(V.conj() * A, axis=0))                                          # Rayleigh quotients
    ref = np.linalg.eigvalsh(g)
    assert np.max(np.abs(np.sort(lam) - ref)) < 1e-12 * ref[-1]
    keep = lam > 1e-12 * lam.max()
    assert keep.sum() == rank
    rec = (V[:, keep] * lam[keep]) @ V[:, keep].conj().T
    assert np.max(np.abs(rec - g)) < 1e-12 * ref[-1]


def fiber_ref(x, X, D, PA, K, PB, Do, No):
    t = x.reshape(PB, K, PA, D).transpose(3, 2, 1, 0)            # [s, a, k, b]
    Xm = X.reshape(No, Do, K, D).transpose(3, 2, 1, 0)           # [s, k, s', n]
    out = np.einsum("sakb,sktn->tanb", t.astype(np.complex128), Xm.astype(np.complex128))
    return out.transpose(3, 2, 1, 0).reshape(-1)                 # memory order: s' fastest, then a, n, b


@pytest.mark.parametrize("dtype,mfma", [(0, 0), (1, 0), (0, 1), (1, 1)])       # (1, 1): ComplexF64 on the f64 matrix cores (kernels_f64.hip), shapes it covers
@pytest.mark.parametrize("D,PA,K,PB,Do,No", [(2, 64, 64, 8, 2, 64), (1, 128, 64, 16, 1, 64), (2, 4, 64, 70, 2, 40), (1, 64, 32, 64, 1, 32), (2, 32, 32, 32, 2, 32), (2, 128, 32, 4, 2, 17), (1, 2, 32, 200, 1, 32), (2, 16, 16, 70, 2, 5),
                                             (1, 2, 3, 5, 1, 3), (1, 100, 10, 7, 1, 10), (1, 1, 7, 130, 1, 7), (2, 9, 5, 4, 2, 3),
                                             (2, 1, 1, 1, 2, 1), (2, 300, 1, 1, 2, 1), (1, 64, 32, 33, 1, 32), (2, 16, 16, 16, 2, 16),
                                             (3, 5, 4, 6, 3, 2), (2, 70, 8, 3, 2, 11),
                                             # D K = 64 with fewer than 32 fibers below the leg (rows of a tile = several b-indices): the register-direct kernels' second addressing mode
                                             (2, 1, 32, 64, 2, 32), (2, 8, 32, 16, 2, 32), (1, 16, 64, 8, 1, 64), (1, 4, 64, 32, 1, 40), (2, 2, 32, 48, 2, 9)])
def test_fiber_gemm(dtype, mfma, D, PA, K, PB, Do, No):
    rng = np.random.default_rng(D + PA + K + PB)
    dt = CDT[dtype]
    x = rnd(rng, D * PA * K * PB, dt)
    X = rnd(rng, D * K * Do * No, dt)
    out = np.zeros(Do * PA * No * PB, dtype=dt)
    n2 = C.c_double()
    rc = lib.tnqs_dbg_fiber_gemm(dtype, D, PA, K, PB, Do, No, x.ctypes.data_as(C.c_void_p), X.ctypes.data_as(C.c_void_p),
                                 out.ctypes.data_as(C.c_void_p), C.byref(n2), mfma)
    assert rc == 0, lib.tnqs_last_error()
    ref = fiber_ref(x, X, D, PA, K, PB, Do, No)
    scale = np.max(np.abs(ref))
    assert np.max(np.abs(out - ref)) < EPS[dtype] * D * K * scale
    assert abs(n2.value - np.sum(np.abs(ref) ** 2)) < 1e-5 * np.sum(np.abs(ref) ** 2)


@pytest.mark.parametrize("dtype,acc64,mfma", [(0, 0, 0), (0, 1, 0), (1, 1, 0), (0, 0, 1), (0, 1, 1), (1, 1, 1)])       # (1, 1, 1): ComplexF64 operands, f64 matrix cores
@pytest.mark.parametrize("D,PA,K,PB,same", [(2, 64, 64, 16, 1), (2, 2, 64, 300, 1), (1, 64, 64, 32, 0), (2, 1024, 32, 32, 1), (2, 32, 32, 1000, 1), (2, 1, 32, 1024, 1), (2, 77, 19, 11, 1), (1, 64, 32, 1024, 0), (1, 2048, 32, 32, 0), (1, 2, 32, 700, 1), (1, 70, 17, 9, 0),
                                           (1, 2, 3, 5, 0), (1, 100, 10, 7, 0), (1, 1, 7, 130, 1), (2, 9, 5, 40, 1), (2, 30, 1, 1, 0),
                                           (1, 64, 32, 33, 0), (2, 16, 16, 16, 1), (2, 512, 32, 8, 1), (3, 5, 5, 6, 0)])
def test_gram(dtype, acc64, mfma, D, PA, K, PB, same):
    if mfma and not acc64 and D * K > 32:
        pytest.skip('f32 MFMA Gram covers D*K <= 32')
    if mfma and acc64 and dtype == 0 and not (same and 16 <= D * K <= 64):
        pytest.skip('f64 MFMA Gram covers X == Y, 16 <= D*K <= 64')
    rng = np.random.default_rng(D + PA + K + PB)
    dt = CDT[dtype]
    x = rnd(rng, D * PA * K * PB, dt)
    y = x if same else rnd(rng, D * PA * K * PB, dt)
    KK = D * K
    odt = np.complex128 if (acc64 or dtype == 1) else np.complex64
    out = np.zeros(KK * KK, dtype=odt)
    rc = lib.tnqs_dbg_gram(dtype, D, PA, K, PB, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p),
                           out.ctypes.data_as(C.c_void_p), acc64, mfma)
    assert rc == 0, lib.tnqs_last_error()
    tx = x.reshape(PB, K, PA, D).transpose(3, 1, 2, 0).reshape(D, K, -1).astype(np.complex128)   # [s, k, (a,b)]
    ty = y.reshape(PB, K, PA, D).transpose(3, 1, 2, 0).reshape(D, K, -1).astype(np.complex128)
    mx = tx.transpose(1, 0, 2).reshape(KK, -1)        # row index s + D*k  -> reshape from [k, s]
    my = ty.transpose(1, 0, 2).reshape(KK, -1)
    ref = (mx @ my.conj().T).T.reshape(-1)            # out[i + KK*j]
    tol = (1e-13 if odt == np.complex128 and dtype == 1 else (1e-6 if acc64 else 3e-5)) * np.max(np.abs(ref))
    assert np.max(np.abs(out - ref)) < tol * max(1.0, np.sqrt(PA * PB / 64))


@pytest.mark.parametrize("chi,bleg", [((16, 16, 16, 16), 0), ((16, 16, 16, 16), 1), ((16, 16, 16, 16), 3), ((16, 16, 16), 2), ((16, 16, 8, 16, 4), 3),
                                      ((32, 32, 32, 32), 0), ((32, 32, 32, 32), 2), ((32, 32, 32, 32), 3), ((32, 32, 32), 1)])
def test_gauge_leg_inside_the_f64_gram(chi, bleg):
    """the gate path's fused kernels (kernels_gate.hip): the last gauge leg r (the lowest leg that is not the bond) absorbed inside the f64
    Gram over the (s, bond) columns -- mfma_gauge_gram64_kernel (chi = 32) and the wave-private mfma_gauge_gram32_kernel (chi = 16), bond leg 0
    (lanes along the bond index) and >= 1 -- against numpy in f64: G = X'^dagger X' of X' = X x_r M.  The kernels round X' to f32 and add
    the squares up in f64, so the difference to the all-f64 reference is the f32 rounding of X' (three-multiplication product: normwise)."""
    lib.tnqs_dbg_gauge_gram.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(sum(chi) + bleg)
    z = len(chi); K = chi[bleg]; KK = 2 * K; r = 1 if bleg == 0 else 0
    n = 2 * int(np.prod(chi))
    x = rnd(rng, n, np.complex64); m = rnd(rng, chi[r] * chi[r], np.complex64)
    out = np.zeros(KK * KK, dtype=np.complex128)
    rc = lib.tnqs_dbg_gauge_gram(z, (C.c_int * z)(*chi), bleg, x.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert rc == 0, lib.tnqs_last_error()
    X = x.reshape((2,) + tuple(chi), order="F").astype(np.complex128)              # [s, l0, l1, ...]
    mm = m.reshape(chi[r], chi[r], order="F").astype(np.complex128)                # M[r, r'] at r + chi r'
    Xp = np.moveaxis(np.tensordot(X, mm, axes=([1 + r], [0])), -1, 1 + r)          # X x_r M
    cols = np.moveaxis(Xp, 1 + bleg, 1).reshape(KK, -1, order="F")                 # rows i = s + 2 k
    ref = (cols @ cols.conj().T).reshape(-1, order="F")                            # out[i + KK j] = sum X'[i, .] conj(X'[j, .])
    err = np.max(np.abs(out - ref)) / np.max(np.abs(ref))
    print(f"chi {chi} bond leg {bleg}: fused gauge + Gram against f64 numpy, relative {err:.1e}")
    assert err < 1e-6            # measured 4e-9 ... 3e-8


@pytest.mark.parametrize("PA,K,PB", [(64, 32, 64), (2048, 32, 8), (2, 32, 32 * 33), (64, 17, 40), (2, 9, 64)])
def test_gram_fused_mode_product(PA, K, PB):
    """fused (X x_r M) + Gram: r = first row leg; rows of a 64-fiber tile are (s:2, i_r:32)"""
    rng = np.random.default_rng(PA + K + PB)
    dt = np.complex64
    x = rnd(rng, PA * K * PB, dt); y = rnd(rng, PA * K * PB, dt); m = rnd(rng, 32 * 32, dt)
    out = np.zeros(K * K, dtype=dt)
    rc = lib.tnqs_dbg_gram_fused(PA, K, PB, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p),
                                 m.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert rc == 0, lib.tnqs_last_error()
    mm = m.reshape(32, 32).T.astype(np.complex128)                      # M[i, j] stored at i + 32 j
    if PA >= 64:        # element (a, k, b), a = s + 2*i + 64*a'
        tx = x.reshape(PB, K, PA // 64, 32, 2).astype(np.complex128)    # [b, k, a', i, s]
        ty = y.reshape(PB, K, PA // 64, 32, 2).astype(np.complex128)
        tx = np.einsum("bkais,ij->bkajs", tx, mm)
        ref = np.einsum("bkajs,bqajs->kq", tx, ty.conj())
    else:               # PA == 2 (a = s), first row leg = fastest part of b
        tx = x.reshape(PB // 32, 32, K, 2).astype(np.complex128)        # [b', i, k, s]
        ty = y.reshape(PB // 32, 32, K, 2).astype(np.complex128)
        tx = np.einsum("biks,ij->bjks", tx, mm)
        ref = np.einsum("bjks,bjqs->kq", tx, ty.conj())
    got = out.reshape(K, K).T                                            # out[i + K j]
    assert np.max(np.abs(got - ref)) < 2e-5 * np.max(np.abs(ref)) * max(1.0, np.sqrt(PA * PB / 64))


@pytest.mark.parametrize("C0,NMID,NHI", [(64, 1, 4), (16, 1, 1), (64, 32, 1), (2048, 1, 1), (32, 3, 2)])
def test_pair_mode_products(C0, NMID, NHI):
    """two slow 32-dimensional legs absorbed in one pass (16 companions per workgroup)"""
    rng = np.random.default_rng(C0 + NMID + NHI)
    dt = np.complex64
    n = C0 * 32 * NMID * 32 * NHI
    x = rnd(rng, n, dt); mx = rnd(rng, 1024, dt); my = rnd(rng, 1024, dt)
    out = np.zeros(n, dtype=dt)
    rc = lib.tnqs_dbg_pair(C0, NMID, NHI, x.ctypes.data_as(C.c_void_p), mx.ctypes.data_as(C.c_void_p), my.ctypes.data_as(C.c_void_p),
                           out.ctypes.data_as(C.c_void_p))
    assert rc == 0, lib.tnqs_last_error()
    t = x.reshape(NHI, 32, NMID, 32, C0).astype(np.complex128)             # [hi, iy, mid, ix, c]
    Mx = mx.reshape(32, 32).T.astype(np.complex128); My = my.reshape(32, 32).T.astype(np.complex128)     # M[i, j] at i + 32 j
    ref = np.einsum("hymxc,xa,yb->hbmac", t, Mx, My).reshape(-1)
    assert np.max(np.abs(out - ref)) < 3e-5 * np.max(np.abs(ref))


def _site(rng, d, chi):
    """site tensor [d][chi_0]..[chi_{z-1}] column-major as a numpy array indexed [s, i_0, ..., i_{z-1}]"""
    shp = (d,) + tuple(chi)
    flat = rnd(rng, int(np.prod(shp)), np.complex64)
    return flat, flat.reshape(shp[::-1]).transpose(*range(len(shp) - 1, -1, -1)).astype(np.complex128)


PAIR_SHAPES = [((32, 32, 32), 1, 2), ((32, 32, 32), 0, 1), ((32, 32, 32), 0, 2), ((32, 32, 32), 2, 0), ((8, 32, 32), 1, 2), ((32, 32, 8), 0, 1),
               ((32, 32, 8, 4), 0, 1), ((32, 16, 32, 2), 0, 2), ((32, 32, 4, 32), 1, 3), ((32, 8, 32, 32), 3, 0), ((32, 32, 32, 32), 0, 3), ((32, 32, 32, 32), 2, 1)]


@pytest.mark.parametrize("chi,lx,ly", PAIR_SHAPES)
def test_pair_mode_products_on_arbitrary_legs(chi, lx, ly):
    """pair kernel with the general companion geometry (leg 0 included: only the 2-dim site index is faster)"""
    rng = np.random.default_rng(sum(chi) + 7 * lx + ly)
    z = len(chi)
    flat, t = _site(rng, 2, chi)
    mx = rnd(rng, 1024, np.complex64); my = rnd(rng, 1024, np.complex64)
    out = np.zeros_like(flat)
    cchi = (C.c_int * z)(*chi)
    rc = lib.tnqs_dbg_pair_legs(2, z, cchi, lx, ly, flat.ctypes.data_as(C.c_void_p), mx.ctypes.data_as(C.c_void_p), my.ctypes.data_as(C.c_void_p),
                                out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    Mx = mx.reshape(32, 32).T.astype(np.complex128); My = my.reshape(32, 32).T.astype(np.complex128)     # M[i, j] at i + 32 j
    ref = np.moveaxis(np.tensordot(t, Mx, axes=([1 + lx], [0])), -1, 1 + lx)
    ref = np.moveaxis(np.tensordot(ref, My, axes=([1 + ly], [0])), -1, 1 + ly)
    got = out.reshape((2,) + tuple(chi), order="F")
    assert np.max(np.abs(got - ref)) < 3e-5 * np.max(np.abs(ref))


@pytest.mark.parametrize("chi,lx,ly", PAIR_SHAPES)
def test_pair_gram_on_arbitrary_legs(chi, lx, ly):
    """last absorption (leg lx) + Gram keeping leg ly, X and Y different tensors"""
    rng = np.random.default_rng(sum(chi) + 5 * lx + ly)
    z = len(chi)
    fx, tx = _site(rng, 2, chi); fy, ty = _site(rng, 2, chi)
    m = rnd(rng, 1024, np.complex64)
    out = np.zeros(1024, dtype=np.complex64)
    cchi = (C.c_int * z)(*chi)
    rc = lib.tnqs_dbg_pair_gram(2, z, cchi, lx, ly, fx.ctypes.data_as(C.c_void_p), fy.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p),
                                out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    M = m.reshape(32, 32).T.astype(np.complex128)
    xm = np.moveaxis(np.tensordot(tx, M, axes=([1 + lx], [0])), -1, 1 + lx)
    axes = [a for a in range(z + 1) if a != 1 + ly]
    ref = np.tensordot(xm, ty.conj(), axes=(axes, axes))               # [b, b']
    got = out.reshape(32, 32).T                                          # out[b + 32 b']
    assert np.max(np.abs(got - ref)) < 3e-5 * np.max(np.abs(ref)) * max(1.0, np.sqrt(tx.size / 65536))


@pytest.mark.parametrize("chi,lx,ly", PAIR_SHAPES)
def test_double_pair_gram(chi, lx, ly):
    """both messages of a plane from one pass over (X, Y): the second one reads the same LDS planes transposed"""
    rng = np.random.default_rng(sum(chi) + 3 * lx + ly)
    z = len(chi)
    fx, tx = _site(rng, 2, chi); fy, ty = _site(rng, 2, chi)
    mx = rnd(rng, 1024, np.complex64); my = rnd(rng, 1024, np.complex64)
    oy = np.zeros(1024, dtype=np.complex64); ox = np.zeros(1024, dtype=np.complex64)
    cchi = (C.c_int * z)(*chi)
    rc = lib.tnqs_dbg_pair_gram2(2, z, cchi, lx, ly, fx.ctypes.data_as(C.c_void_p), fy.ctypes.data_as(C.c_void_p), mx.ctypes.data_as(C.c_void_p),
                                 my.ctypes.data_as(C.c_void_p), oy.ctypes.data_as(C.c_void_p), ox.ctypes.data_as(C.c_void_p))
    assert rc == 0
    Mx = mx.reshape(32, 32).T.astype(np.complex128); My = my.reshape(32, 32).T.astype(np.complex128)
    scale = max(1.0, np.sqrt(tx.size / 65536))
    for (absorbed, kept, M, got) in ((lx, ly, Mx, oy), (ly, lx, My, ox)):
        xm = np.moveaxis(np.tensordot(tx, M, axes=([1 + absorbed], [0])), -1, 1 + absorbed)
        axes = [a for a in range(z + 1) if a != 1 + kept]
        ref = np.tensordot(xm, ty.conj(), axes=(axes, axes))
        assert np.max(np.abs(got.reshape(32, 32).T - ref)) < 3e-5 * np.max(np.abs(ref)) * scale


@pytest.mark.parametrize("D,PA,K,PB", [(2, 64, 32, 64), (1, 128, 64, 32), (2, 1, 32, 2048)])
def test_bf16x3_fiber_gemm_and_gram_are_f32_accurate(D, PA, K, PB):
    """the other kernels of the bf16 x 3 route (csrc/kernels_x3.hip: x3_rowgemm64_kernel -- chi = 32 epilogue with the site index folded in, chi = 64 mode
    product -- and x3_gram64_kernel) against f64 references, held to f32-class bounds: 1e-6 of the largest entry for the 64-term products, 3e-6 for the Gram's
    sums of PA PB terms (the generic tests above allow EPS D K = 4e-5 resp. 3e-5)"""
    rng = np.random.default_rng(D + PA + K + PB)
    x = rnd(rng, D * PA * K * PB, np.complex64)
    X = rnd(rng, D * K * D * K, np.complex64)
    out = np.zeros(D * PA * K * PB, dtype=np.complex64)
    n2 = C.c_double()
    assert lib.tnqs_dbg_fiber_gemm(0, D, PA, K, PB, D, K, x.ctypes.data_as(C.c_void_p), X.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.byref(n2), 1) == 0
    ref = fiber_ref(x, X, D, PA, K, PB, D, K)
    assert np.max(np.abs(out - ref)) < 1e-6 * np.max(np.abs(ref))
    if D * K == 64 and PA * PB >= 64:
        y = rnd(rng, D * PA * K * PB, np.complex64)
        KK = D * K
        g = np.zeros(KK * KK, dtype=np.complex64)
        assert lib.tnqs_dbg_gram(0, D, PA, K, PB, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), g.ctypes.data_as(C.c_void_p), 0, 1) == 0
        tx = x.reshape(PB, K, PA, D).transpose(3, 1, 2, 0).reshape(D, K, -1).astype(np.complex128)
        ty = y.reshape(PB, K, PA, D).transpose(3, 1, 2, 0).reshape(D, K, -1).astype(np.complex128)
        mx = tx.transpose(1, 0, 2).reshape(KK, -1); my = ty.transpose(1, 0, 2).reshape(KK, -1)
        gref = (mx @ my.conj().T).T.reshape(-1)
        assert np.max(np.abs(g - gref)) < 3e-6 * np.max(np.abs(gref))


@pytest.mark.parametrize("lx,ly", [(1, 2), (0, 3), (2, 0)])
@pytest.mark.parametrize("scale", [1e-36, 1e-12, 1e-3, 1e6])
def test_bf16x3_plane_kernels_are_f32_accurate(lx, ly, scale):
    """round 5: the chi = 32 plane kernels multiply on the bf16 matrix cores -- every f32 operand split EXACTLY into three bf16 pieces, six of the nine piece
    products kept (the dropped ones are below one f32 rounding of the product), f32 accumulation (csrc/kernels_x3.hip).  That is f32 arithmetic, not bf16
    arithmetic: against an f64 reference the error has to be what the f32 matrix instructions give (measured on the same inputs, profiles/x3_error.py:
    3.0-5.6e-7 of the largest entry for the split route, 2.6-6.2e-7 for v_mfma_f32_32x32x2_f32; a bf16-rounded operand alone would give 4e-3), at any
    scale of the data (bf16 has the exponent range of f32: no scaling anywhere).  Bound: 1e-6 of the largest entry, thirty times tighter than the generic
    kernel tests above."""
    rng = np.random.default_rng(11 * lx + ly)
    chi = (32, 32, 32, 32); z = 4
    cchi = (C.c_int * z)(*chi)
    fx, tx = _site(rng, 2, chi); fy, ty = _site(rng, 2, chi)
    fx *= np.float32(scale); tx = tx * float(np.float32(scale)); fy *= np.float32(scale); ty = ty * float(np.float32(scale))
    # data at 1e-36 (round-5 verdict: the untested corner): the LOW bf16 piece of an operand below ~2^-110 is a denormal and the matrix cores flush it -- the kernels
    # themselves lose 16-bit instead of 24-bit operands there (measured 1.5e-5; asserted as the documented bound, 1e-4).  Data of that size are outside the domain
    # of the ComplexF32 path on ANY route (test_gate_path_is_scale_invariant_where_f32_is below)
    tol = 1e-4 if scale < 1e-30 else 1e-6
    mx = rnd(rng, 1024, np.complex64); my = rnd(rng, 1024, np.complex64)
    Mx = mx.reshape(32, 32).T.astype(np.complex128); My = my.reshape(32, 32).T.astype(np.complex128)
    out = np.zeros_like(fx)
    assert lib.tnqs_dbg_pair_legs(2, z, cchi, lx, ly, fx.ctypes.data_as(C.c_void_p), mx.ctypes.data_as(C.c_void_p), my.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)) == 0
    ref = np.moveaxis(np.tensordot(tx, Mx, axes=([1 + lx], [0])), -1, 1 + lx)
    ref = np.moveaxis(np.tensordot(ref, My, axes=([1 + ly], [0])), -1, 1 + ly)
    assert np.max(np.abs(out.reshape((2,) + chi, order="F") - ref)) < tol * np.max(np.abs(ref))
    if scale < 1e-30:
        return          # (the Gram of two tensors at 1e-36 underflows in f32 on any route: 1e-72)
    oy = np.zeros(1024, dtype=np.complex64); ox = np.zeros(1024, dtype=np.complex64)
    assert lib.tnqs_dbg_pair_gram2(2, z, cchi, lx, ly, fx.ctypes.data_as(C.c_void_p), fy.ctypes.data_as(C.c_void_p), mx.ctypes.data_as(C.c_void_p),
                                   my.ctypes.data_as(C.c_void_p), oy.ctypes.data_as(C.c_void_p), ox.ctypes.data_as(C.c_void_p)) == 0
    for (absorbed, kept, M, got) in ((lx, ly, Mx, oy), (ly, lx, My, ox)):
        xm = np.moveaxis(np.tensordot(tx, M, axes=([1 + absorbed], [0])), -1, 1 + absorbed)
        axes = [a for a in range(z + 1) if a != 1 + kept]
        ref = np.tensordot(xm, ty.conj(), axes=(axes, axes))
        assert np.max(np.abs(got.reshape(32, 32).T - ref)) < 2 * tol * np.max(np.abs(ref))


@pytest.mark.parametrize("scale", [1e-7, 1e-3, 1e4])
def test_gate_path_is_scale_invariant_where_f32_is(scale):
    """where does the corner above begin to matter?  Nowhere the ComplexF32 path itself works.  theta = gate . R1 R2 carries the product of the two site tensors'
    norms and the truncation rule squares its singular values IN THE DATA'S PRECISION (NDTensors truncate! on P = S^2; gate_finish_kernel does the same on
    purpose): with tensor norms of 1e-12, P = 1e-48 is zero in f32 and every bond is cut to dimension 1 -- in the reference's ComplexF32 arithmetic as much as
    here (measured on this path: chi' = 1 and error 0 at 1e-12, NaN at 1e12) -- long before an entry reaches 2^-110 = 8e-34.  Inside that domain the bf16 split
    is exact: one colour group of Rzz gates with unset (identity) messages and normalize_tensors on tensors at 1e-7, 1e-3 and 1e4 against the same tensors at 1:
    the gauge-invariant results -- spectrum (the bond messages diag(S)), truncation errors, <Z> on the gate vertices -- agree to f32 rounding."""
    g = tn.named_grid((4, 4)); chi = 32
    rng = np.random.default_rng(5)
    tens = {v: rnd(rng, (2,) + (chi,) * g.degree(v), np.complex64) / np.float32(np.sqrt(2.0 * chi ** g.degree(v))) for v in g.vertices}
    grp = tn.edge_color(g, 4)[0]
    layer = [("Rzz", [a, b], 0.3) for (a, b) in grp]
    outs = []
    for sc in (1.0, scale):
        b = tn.BeliefPropagationCache(tn.tensornetworkstate(np.complex64, lambda v: "↑", g))
        for v in g.vertices:
            b._set_tensor(v, tens[v] * np.float32(sc))
        b, errs = tn.apply_gates(layer, b, apply_kwargs=dict(maxdim=chi, cutoff=1e-10, normalize_tensors=True), update_cache=False)
        outs.append(([np.diag(b.message(e)).real for e in grp], errs, tn.expect_all(b, "Z").real, [b.bond_dim(a, c) for (a, c) in grp]))
    (s1, e1, z1, d1), (s2, e2, z2, d2) = outs
    assert d1 == d2 == [chi] * len(grp)
    assert max(np.max(np.abs(a - b)) / np.max(a) for a, b in zip(s1, s2)) < 3e-6
    assert np.max(np.abs(e1 - e2)) < 2e-5 * float(np.max(e1))
    assert np.max(np.abs(z1 - z2)) < 1e-5


@pytest.mark.parametrize("shape,rank", [((64, 64), 64), ((128, 128), 64), ((144, 144), 72), ((200, 120), 120)])
@pytest.mark.parametrize("scale", [1e-18, 1e-10, 1e-5, 1.0, 1e12])
def test_jacobi_svd_is_scale_invariant(shape, rank, scale):
    """ComplexF32: the sweeps square inner products, which underflows for matrices of small magnitude (singular values of 1e-5 are
    enough: the rotation phases of the smaller columns lose their unit modulus).  Both kernels (LDS-resident up to 136 rows here, global
    memory beyond) scale the matrix by an exact power of two first.  Before that fix the global-memory kernel returned singular values
    off by 20-40 % for theta matrices of chi >= 36 whose tensors carried a small norm; the LDS kernel skipped rotations below 1e-36."""
    rng = np.random.default_rng(shape[0] + rank)
    m, n = shape
    b0 = (rnd(rng, (m, rank), np.complex128) @ rnd(rng, (rank, n), np.complex128)) / rank
    b = (b0 * scale).astype(np.complex64)
    s_ref = np.linalg.svd(b.astype(np.complex128), compute_uv=False)
    A, V, sw = jacobi(b, 0)
    s = np.sort(np.linalg.norm(A.astype(np.complex128), axis=0))[::-1]
    assert sw < 60
    assert np.max(np.abs(s[:len(s_ref)] - s_ref)) < 3e-5 * s_ref[0]
    assert np.all(np.isfinite(V))
    if scale >= 1e-15:                                   # V is recovered by an f32 GEMM of theta0 with U Sigma: its products underflow below ~1e-18;
        rec = (A.astype(np.complex128) @ V.conj().T.astype(np.complex128))      # inside the engine theta is scaled to O(1) first (theta_scale_kernel)
        assert np.max(np.abs(rec - b)) < 2e-5 * s_ref[0]


@pytest.mark.parametrize("shape,rank", [((128, 64), 64), ((128, 64), 20), ((128, 40), 40), ((64, 64), 64), ((100, 33), 33), ((256, 64), 48), ((72, 8), 8),
                                        ((130, 70), 70), ((40, 40), 3)])
@pytest.mark.parametrize("scale", [1e-9, 1.0, 1e6])
def test_theta_svd_kernel(shape, rank, scale, monkeypatch):
    """the engine's theta SVD route at the shapes a gate batch produces (one-sided f32 sweeps on A in LDS, V not accumulated but recovered from
    the unrotated copy): singular values against LAPACK relative to the largest one, the reconstruction A = (U Sigma) V^dagger, and -- what
    the V recovery depends on -- the orthogonality of the columns of U Sigma relative to their OWN norms, for full-rank, rank-deficient and
    badly scaled inputs with spectra spread over 3.5 decades"""
    monkeypatch.setenv("TNQS_DBG_THETA_SVD", "1")
    rng = np.random.default_rng(shape[0] + 7 * rank)
    m, n = shape
    dec = np.exp(-np.arange(rank) * (8.0 / max(rank, 1)))                    # singular values spread over 3.5 decades, as a truncating theta has them
    q1, _ = np.linalg.qr(rnd(rng, (m, rank), np.complex128)); q2, _ = np.linalg.qr(rnd(rng, (n, rank), np.complex128))
    b = ((q1 * dec) @ q2.conj().T * scale).astype(np.complex64)
    s_ref = np.linalg.svd(b.astype(np.complex128), compute_uv=False)
    A, V, sw = jacobi(b, 0)
    A = A.astype(np.complex128)
    nrm = np.linalg.norm(A, axis=0)
    s = np.sort(nrm)[::-1]
    assert sw < 60
    assert np.max(np.abs(s[:len(s_ref)] - s_ref)) < 1e-5 * s_ref[0], (s[:4], s_ref[:4])
    big = nrm > 1e-4 * s_ref[0]                                             # columns that carry signal: mutually orthogonal relative to their own norms
    U = A[:, big] / nrm[big]
    assert np.max(np.abs(U.conj().T @ U - np.eye(U.shape[1]))) < 5e-6
    rec = A @ V.conj().T.astype(np.complex128)
    assert np.max(np.abs(rec - b)) < 2e-5 * s_ref[0]


@pytest.mark.parametrize("n", [1, 2, 5, 16, 33, 64, 65, 96, 128])
@pytest.mark.parametrize("cond", [1e2, 1e10])
def test_cholesky_kernels(n, cond):
    """chol_kernel (n <= 96: one barrier per column, unscaled trailing updates, four lanes per column of the inverse) and the packed kernel
    (n <= 128): G = L L^dagger, W = (L^-1)^dagger, against numpy in f64, for well- and ill-conditioned Gram matrices; a numerically singular
    matrix raises the failure flag instead of producing NaNs"""
    rng = np.random.default_rng(n)
    q, _ = np.linalg.qr(rnd(rng, (n, n), np.complex128))
    lam = np.logspace(0, -np.log10(cond), n)
    g = np.asfortranarray((q * lam) @ q.conj().T)
    L = np.zeros((n, n), dtype=np.complex128, order="F"); W = np.zeros((n, n), dtype=np.complex128, order="F"); fail = C.c_int(-1)
    rc = lib.tnqs_dbg_chol(n, g.ctypes.data_as(C.c_void_p), L.ctypes.data_as(C.c_void_p), W.ctypes.data_as(C.c_void_p), C.byref(fail), C.c_double(1e-12))
    assert rc == 0, tn._lib.lib.tnqs_last_error()
    assert fail.value == 0
    assert np.max(np.abs(np.triu(L, 1))) == 0 and np.all(np.diag(L).real > 0) and np.max(np.abs(np.diag(L).imag)) == 0
    assert np.max(np.abs(L @ L.conj().T - g)) < 1e-13 * n
    ref = np.linalg.cholesky(g)
    assert np.max(np.abs(L - ref)) < 1e-11 * cond ** 0.5
    assert np.max(np.abs(W.conj().T @ L - np.eye(n))) < 1e-13 * n * cond ** 0.5            # W^dagger = L^-1
    # singular: rank n - 1 (n >= 2) -> flagged, finite output
    if n >= 2:
        lam2 = lam.copy(); lam2[-1] = 0.0
        g2 = np.asfortranarray((q * lam2) @ q.conj().T)
        rc = lib.tnqs_dbg_chol(n, g2.ctypes.data_as(C.c_void_p), L.ctypes.data_as(C.c_void_p), W.ctypes.data_as(C.c_void_p), C.byref(fail), C.c_double(1e-12))
        assert rc == 0 and (fail.value == 1 or cond > 1e9) and np.all(np.isfinite(L))


# ---- preconditioned theta SVD kernel (round 5) ----------------------------------------------------------------------------------------
def theta_svd_pre(a, q=None, copies=1, reps=0, cap=0):
    """q given: a is the low-rank factor of theta = a q^T, V (nq x n) = right singular vectors of theta; q None: a is theta itself, V is n x n"""
    m, n = a.shape; nq = q.shape[0] if q is not None else n
    A = np.asfortranarray(a.astype(np.complex64)); Q = np.asfortranarray(q.astype(np.complex128)) if q is not None else None
    V = np.zeros((nq, n), dtype=np.complex64, order="F")
    sw = C.c_int(); ms = C.c_double(0.0); ph = (C.c_double * 6)()
    rc = lib.tnqs_dbg_theta_svd_pre(m, n, nq, A.ctypes.data_as(C.c_void_p), Q.ctypes.data_as(C.c_void_p) if Q is not None else None, V.ctypes.data_as(C.c_void_p), C.byref(sw),
                                    copies, reps, C.byref(ms), ph, cap)
    assert rc == 0, lib.tnqs_last_error()
    theta_svd_pre.phases_us = [round(x, 1) for x in ph]          # load, Gram, Cholesky + conversion, sweeps, U Sigma, V (workgroup 0)
    return A, V, sw.value, ms.value


def low_rank_factors(r1, r2, gate):
    """the engine's low-rank route in numpy (f64): operator-sum factors of the gate by elimination with complete pivoting, theta = A B^T, B = Q L^dagger
    (Cholesky QR), M = A conj(L) -- returns (M, Q, theta)"""
    d1 = d2 = 2
    g4 = gate.reshape(d1, d2, d1, d2)
    O = np.transpose(g4, (0, 2, 1, 3)).reshape(d1 * d1, d2 * d2).astype(np.complex128).copy()
    fa, fb, amax = [], [], np.abs(O).max()
    for _ in range(min(O.shape)):
        i, j = np.unravel_index(np.argmax(np.abs(O)), O.shape)
        if not np.abs(O[i, j]) > 1e-13 * amax:
            break
        col, row = O[:, j].copy(), O[i, :] / O[i, j]
        O -= np.outer(col, row); fa.append(col.reshape(d1, d1)); fb.append(row.reshape(d2, d2))
    R1, R2 = r1.astype(np.complex128), r2.astype(np.complex128)
    A = np.concatenate([np.einsum("xs,rsb->rxb", a, R1).reshape(R1.shape[0] * d1, -1) for a in fa], axis=1)
    B = np.concatenate([np.einsum("ys,rsb->ryb", b, R2).reshape(R2.shape[0] * d2, -1) for b in fb], axis=1)
    L = np.linalg.cholesky(B.conj().T @ B)
    Q = B @ np.linalg.inv(L).conj().T
    return A @ np.conj(L), Q, A @ B.T


def check_theta_svd_pre(M, Q, theta, sv_tol=4e-6, cap=0):
    A, V, sw, _ = theta_svd_pre(M, Q, cap=cap)
    A = A.astype(np.complex128); V = V.astype(np.complex128)
    M32 = M.astype(np.complex64).astype(np.complex128)
    s_ref = np.linalg.svd(M32, compute_uv=False)
    nrm = np.linalg.norm(A, axis=0)
    assert 0 < sw < 30
    assert np.max(np.abs(np.sort(nrm)[::-1] - s_ref)) < sv_tol * s_ref[0], (np.sort(nrm)[::-1][:4], s_ref[:4])      # EVERY singular value (the truncation error needs them all)
    n = M.shape[1]
    order = np.argsort(-nrm, kind="stable")
    keep = order[: (cap if 0 < cap < n else n)]                                               # the columns whose vectors must be formed
    # the others leave as sigma_j e_0 -- except those within 1e-4 of the cap-th singular value (ties at the cap are formed too: the consumer ranks again)
    other = np.array([j for j in order[len(keep):] if nrm[j] < nrm[keep[-1]] * (1 - 2e-4)], dtype=int)
    if len(other):
        assert np.all(A[1:, other] == 0) and np.allclose(A[0, other].real, nrm[other]) and np.all(A[0, other].imag == 0)
        assert np.all(V[:, other] == 0)                                                       # never garbage
    full = M32 @ Q.T if Q is not None else M32                                                # theta
    u, sv, vh = np.linalg.svd(full, full_matrices=False)
    k = len(keep)
    best = (u[:, :k] * sv[:k]) @ vh[:k]                                                       # best rank-k approximation
    rec = A[:, keep] @ V[:, keep].conj().T                                                    # (U Sigma) V^dagger over the kept triplets
    assert np.max(np.abs(rec - best)) < 3e-6 * s_ref[0] * np.sqrt(n), float(np.max(np.abs(rec - best)) / s_ref[0])
    if theta is not None and k == n:
        assert np.max(np.abs(rec - theta)) < 3e-6 * s_ref[0] * np.sqrt(n)
    big = nrm[keep] > 1e-5 * s_ref[0]                                       # V: orthonormal to f32 rounding whatever the singular value (no division by Sigma^2)
    Vb = V[:, keep][:, big]
    assert np.max(np.abs(Vb.conj().T @ Vb - np.eye(Vb.shape[1]))) < 3e-6
    nb = nrm[keep][big]
    U = A[:, keep][:, big] / nb                                             # U Sigma = M U_L: absolute accuracy eps * sigma_max per column (like LAPACK's)
    assert np.max(np.abs((U.conj().T @ U - np.eye(U.shape[1])) * np.minimum.outer(nb, nb))) < 3e-6 * s_ref[0]
    return sw


@pytest.mark.parametrize("shape,rank", [((128, 64, 128), 64), ((128, 64, 128), 20), ((128, 40, 100), 40), ((64, 64, 64), 64), ((100, 33, 80), 33), ((72, 8, 40), 8),
                                        ((40, 40, 40), 3), ((128, 64, 64), 64), ((6, 2, 4), 2)])
@pytest.mark.parametrize("scale", [1e-9, 1.0, 1e6])
def test_theta_svd_pre_kernel(shape, rank, scale):
    """synthetic factors with spectra spread over 3.5 decades, full rank and rank deficient, badly scaled: singular values against LAPACK, the reconstruction
    theta = (U Sigma) V^dagger, V orthonormal, U Sigma orthogonal to eps sigma_max"""
    m, n, nq = shape
    rng = np.random.default_rng(m + 7 * rank + n)
    dec = np.exp(-np.arange(rank) * (8.0 / max(rank, 1)))
    q1, _ = np.linalg.qr(rnd(rng, (m, rank), np.complex128)); q2, _ = np.linalg.qr(rnd(rng, (n, rank), np.complex128))
    M = (q1 * dec) @ q2.conj().T * scale
    Q, _ = np.linalg.qr(rnd(rng, (nq, n), np.complex128))
    check_theta_svd_pre(M, Q, None)
    if rank == n and n >= 8:
        check_theta_svd_pre(M, Q, None, cap=n // 2)           # only the n/2 largest triplets are formed (the bond dimension cap of a gate)
    check_theta_svd_pre(M, None, None, cap=(n // 2 if n >= 8 else 0))      # theta itself (a corner gate: no low-rank route): V = its right singular vectors


@pytest.mark.parametrize("with_q", [False, True])
def test_theta_svd_pre_kernel_on_degenerate_input(with_q):
    """round-5 advisor finding: theta identically ZERO used to come back with sigma_j = 1 on the unformed columns (largest pivot 0 -> every pivot replaced by 1) --
    weight in S and in the truncation error that the matrix does not have; a NaN entry used to be sorted to the end as a zero column instead of being reported.
    Now: zero in, zero out (U Sigma = 0, V = 0, no sweeps); NaN in, NaN kept in A (gate_finish reports TNQS_ERR_NUMERIC, like the plain Jacobi route) and V = NaN."""
    m, n, nq = 128, 64, 128
    rng = np.random.default_rng(3)
    Q = np.linalg.qr(rnd(rng, (nq, n), np.complex128))[0] if with_q else None
    A, V, sw, _ = theta_svd_pre(np.zeros((m, n), dtype=np.complex64), Q)
    assert sw == 0 and np.all(A == 0) and np.all(V == 0)
    M = rnd(rng, (m, n), np.complex64); M[5, 7] = np.nan
    A, V, sw, _ = theta_svd_pre(M, Q)
    assert sw == 0 and np.isnan(A[5, 7]) and np.all(np.isnan(V.real))
    np.testing.assert_array_equal(np.delete(A.ravel(order="F"), 5 + m * 7), np.delete(np.asfortranarray(M).ravel(order="F"), 5 + m * 7))      # nothing else was touched


def test_theta_svd_pre_kernel_on_harvested_factors():
    """R factors of two-site gates harvested from oracle runs (tests/golden/make_theta_factors.py: a random chi = 32 state after three TFIM layers, a chi = 16
    state evolved from the product state): the engine's low-rank factor M and Q built in numpy, then the kernel against LAPACK.  Sweeps: the numpy model of
    the kernel (scratch notes in DESIGN.md 4.26) needs 3-7 where the plain Jacobi on M needs 7-9"""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "theta_factors.npz"))
    keys = sorted({k.rsplit("_", 1)[0] for k in z.files})
    sweeps = {}
    for k in keys:
        r1, r2, gate = z[k + "_r1"], z[k + "_r2"], z[k + "_gate"]
        if r1.shape[0] * 2 < r2.shape[0] * 2:           # theta wider than tall: the engine stores its adjoint and runs no low-rank route
            continue
        M, Q, theta = low_rank_factors(r1, r2, gate)
        if not (M.shape[1] <= 64 and M.shape[0] <= 128):
            continue
        sweeps[k] = check_theta_svd_pre(M, Q, theta)
    print("sweeps of the preconditioned kernel:", sweeps)
    assert len(sweeps) >= 6 and max(sweeps.values()) <= 8

