"""Independent dense state-vector simulator + exact tensor-network contraction.

TEST INFRASTRUCTURE ONLY (see oracle/tnqs_oracle.py header).  Shares no code with the
simple-update / BP restatement: gates are applied to the full 2^n amplitude tensor, and a
TensorNetworkState is contracted exactly by one einsum.  Used to pin the oracle on the
reference's docstring claim "exact if no truncation is performed"
(/root/reference/src/Apply/simple_update.jl:4) and on test/test_apply.jl:20,53
(`norm_sqr(psi; alg = "exact") ≈ 1`).
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np


def product_statevector(g, vecs: Dict) -> np.ndarray:
    """amplitude tensor of shape (d,)*n, axis i = g.vertices[i]"""
    psi = np.ones((), dtype=complex)
    for v in g.vertices:
        psi = np.multiply.outer(psi, np.asarray(vecs[v], dtype=complex))
    return psi


def apply_gate_statevector(psi: np.ndarray, g, mat: np.ndarray, verts: Sequence) -> np.ndarray:
    axes = [g.pos[v] for v in verts]
    k = len(axes)
    d = psi.shape[axes[0]]
    m = np.asarray(mat, dtype=complex).reshape((d,) * (2 * k))      # [out..., in...]
    psi = np.tensordot(m, psi, axes=(list(range(k, 2 * k)), axes))  # out axes first
    return np.moveaxis(psi, list(range(k)), axes)


def run_circuit_statevector(g, vecs: Dict, circuit: List) -> np.ndarray:
    from tnqs_oracle import resolve_gate
    psi = product_statevector(g, vecs)
    for gate in circuit:
        mat, verts = resolve_gate(gate)
        psi = apply_gate_statevector(psi, g, mat, verts)
    return psi


def tns_to_statevector(tns) -> np.ndarray:
    """exact contraction of all virtual bonds -> (d,)*n amplitude tensor (axis i = vertex i)"""
    g = tns.g
    n = len(g.vertices)
    elabel = {}
    nxt = n
    for (a, b) in g.edges:
        elabel[(a, b)] = nxt
        elabel[(b, a)] = nxt
        nxt += 1
    args = []
    for i, v in enumerate(g.vertices):
        args.append(np.asarray(tns.tensors[v], dtype=complex))
        args.append([i] + [elabel[(v, w)] for w in g.nbrs[v]])
    args.append(list(range(n)))
    return np.einsum(*args, optimize="greedy")


def expect_statevector(psi: np.ndarray, g, op: np.ndarray, v) -> complex:
    ax = g.pos[v]
    t = np.tensordot(np.asarray(op, dtype=complex), psi, axes=([1], [ax]))
    t = np.moveaxis(t, 0, ax)
    return complex(np.vdot(psi, t) / np.vdot(psi, psi))


def expect_statevector_multi(psi: np.ndarray, g, ops: Dict) -> complex:
    """<psi| prod_v op_v |psi> / <psi|psi> for operators on several vertices (op[s', s])"""
    t = psi
    for v, op in ops.items():
        ax = g.pos[v]
        t = np.moveaxis(np.tensordot(np.asarray(op, dtype=complex), t, axes=([1], [ax])), 0, ax)
    return complex(np.vdot(psi, t) / np.vdot(psi, psi))


def rdm_statevector(psi: np.ndarray, g, v) -> np.ndarray:
    ax = g.pos[v]
    m = np.moveaxis(psi, ax, 0).reshape(psi.shape[ax], -1)
    rho = m @ m.conj().T
    return rho / np.trace(rho)


def fidelity(a: np.ndarray, b: np.ndarray) -> float:
    return float(abs(np.vdot(a, b)) ** 2 / (np.vdot(a, a).real * np.vdot(b, b).real))
