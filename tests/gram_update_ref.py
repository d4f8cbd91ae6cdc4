"""numpy restatement of the algorithm the HIP library uses for a two-site gate (DESIGN.md "simple update on the
device"): instead of the reference's thin QR of the gauged tensors (src/Apply/simple_update.jl:45-48) it forms
G_i = psi~_i^dagger psi~_i (f64), factorises G_i = W L W^dagger (R_i = L^1/2 W^dagger is a valid "R factor": any
psi~ = Q R with orthonormal Q gives the same gauge-invariant result), builds theta from R_1, R_2 and the gate, SVDs
and truncates it exactly like the reference, and applies X_i = R_i^+ (U sqrt S | sqrt S V^dagger) to the UN-gauged
tensor (the gauge M^1/2 and un-gauge M^-1/2 cancel to the projector onto the support of the message).
Test infrastructure: used on CPU to (a) prove the substitution against the oracle's QR path and (b) simulate the
vertex-sharded exchange protocol with gloo (tests/test_sharding_cpu.py).  split into the per-rank pieces:
    site_gram(...)      runs on the owner of the site            -> G        (exchanged: all-gather #1)
    gate_algebra(...)   runs on the owner of the first vertex    -> chi', truncerr, S, X1, X2   (S, X2: all-gather #2)
    site_apply(...)     runs on the owner of the site            -> new site tensor
"""
import numpy as np

import tnqs_oracle as o

RANK_TAU = 1e-12


def _gauged(psi, envs, sqrt_cutoff):
    t, projs = psi, []
    for ax, m in envs:
        ms, mi = o.pseudo_sqrt_inv_sqrt(m, sqrt_cutoff)
        t = o._absorb(t, ax, ms)
        projs.append((ax, ms.astype(np.complex128) @ mi.astype(np.complex128)))
    return t, projs


def _as_matrix(t, bax):
    outer = [i for i in range(t.ndim) if i not in (0, bax)]
    tm = np.transpose(t, outer + [bax, 0])               # columns (b, s) -> index s + d*b (s fastest)
    d, chi = t.shape[0], t.shape[bax]
    return tm.reshape(-1, chi * d), outer


def site_gram(psi, bax, envs, sqrt_cutoff):
    t, projs = _gauged(psi, envs, sqrt_cutoff)
    a, _ = _as_matrix(t.astype(np.complex128), bax)
    return a.conj().T @ a, projs


def _factor(g):
    lam, w = np.linalg.eigh(g)
    keep = lam > RANK_TAU * lam.max()
    return lam[keep], w[:, keep]


def gate_algebra(gate, g1, g2, d1, d2, chi, maxdim, cutoff, normalize, real_dtype):
    l1, w1 = _factor(g1)
    l2, w2 = _factor(g2)
    r1 = (np.sqrt(l1)[:, None] * w1.conj().T).reshape(len(l1), chi, d1)      # R1[a, b, s]
    r2 = (np.sqrt(l2)[:, None] * w2.conj().T).reshape(len(l2), chi, d2)
    theta = np.einsum("abs,cbt->asct", r1, r2)
    g4 = np.asarray(gate, dtype=np.complex128).reshape(d1, d2, d1, d2)
    theta = np.einsum("xyst,asct->axcy", g4, theta).reshape(len(l1) * d1, len(l2) * d2)
    u, s, vh = np.linalg.svd(theta.astype(np.complex64 if real_dtype == np.float32 else np.complex128), full_matrices=False)
    n, err = o.truncate_spectrum(s.astype(real_dtype) ** 2, maxdim, cutoff)
    u, s, vh = u[:, :n].astype(np.complex128), s[:n].astype(np.float64), vh[:n].astype(np.complex128)
    L = (u * np.sqrt(s)).reshape(len(l1), d1, n)                               # [a, s1', u]
    R = (np.sqrt(s)[:, None] * vh).reshape(n, len(l2), d2)                     # [u, c, s2']
    x1 = np.einsum("ka,asu->ksu", w1 / np.sqrt(l1), L)                         # [(b,s) as kk, s1', u]
    x2 = np.einsum("kc,uct->ktu", w2 / np.sqrt(l2), R)
    sv = s / np.linalg.norm(s) if normalize else s
    return n, err, sv, x1, x2


def site_apply(psi, bax, projs, x, normalize):
    t = psi.astype(np.complex128)
    for ax, p in projs:
        if np.max(np.abs(p - np.eye(p.shape[0]))) > 1e-12:
            t = o._absorb(t, ax, p)
    a, outer = _as_matrix(t, bax)
    d, chi = psi.shape[0], psi.shape[bax]
    n = x.shape[2]
    out = (a @ x.reshape(chi * d, d * n)).reshape([psi.shape[i] for i in outer] + [d, n])    # outer..., s', u
    cur = list(outer) + [0, bax]
    out = np.transpose(out, [cur.index(i) for i in range(psi.ndim)])
    if normalize:
        out = out / np.linalg.norm(out)
    return out.astype(psi.dtype)


def x_as_columns(x):
    """[kk = s + d*b, s', u] with kk ordered (b slow, s fast) as produced by _as_matrix"""
    return x


def apply_gate_gram(bpc, gate, v1, v2, maxdim=None, cutoff=None, normalize_tensors=True):
    """single-process composition of the three pieces: mutates bpc like oracle.apply_gate"""
    g = bpc.g
    dt = bpc.tns.dtype
    real = np.float32 if dt == np.complex64 else np.float64
    sc = 10 * np.finfo(real).eps
    p1, p2 = bpc.tns.tensors[v1], bpc.tns.tensors[v2]
    b1, b2 = g.leg(v1, v2), g.leg(v2, v1)
    env1 = [(g.leg(v1, k), bpc.message((k, v1))) for k in g.nbrs[v1] if k != v2]
    env2 = [(g.leg(v2, k), bpc.message((k, v2))) for k in g.nbrs[v2] if k != v1]
    g1, pr1 = site_gram(p1, b1, env1, sc)
    g2, pr2 = site_gram(p2, b2, env2, sc)
    n, err, sv, x1, x2 = gate_algebra(gate, g1, g2, p1.shape[0], p2.shape[0], p1.shape[b1], maxdim, cutoff, normalize_tensors, real)
    bpc.tns.tensors[v1] = site_apply(p1, b1, pr1, x1, normalize_tensors)
    bpc.tns.tensors[v2] = site_apply(p2, b2, pr2, x2, normalize_tensors)
    md = np.diag(sv).astype(dt)
    bpc.messages[(v1, v2)] = md.copy()
    bpc.messages[(v2, v1)] = md.copy()
    return err
