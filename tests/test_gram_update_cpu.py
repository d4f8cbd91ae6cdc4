"""CPU proof that the library's Gram/eigen formulation of the two-site update (tests/gram_update_ref.py) is
equivalent to the reference's QR formulation (oracle.simple_update) on gauge-invariant quantities."""
import numpy as np
import pytest

import tnqs_oracle as o
import statevector as sv
import gram_update_ref as gr

Z = np.diag([1.0, -1.0]).astype(complex)


@pytest.mark.parametrize("lattice,chi", [("grid3x3", 3), ("hex22", 2), ("line4", 3)])
@pytest.mark.parametrize("maxdim", [None, 2])
def test_gram_formulation_equals_qr_formulation(lattice, chi, maxdim):
    g = {"grid3x3": lambda: o.named_grid((3, 3)), "hex22": lambda: o.named_hexagonal_lattice_graph(2, 2),
         "line4": lambda: o.named_grid((4,))}[lattice]()
    psi = o.random_state(np.complex128, g, chi, seed=21)
    bpc = o.update(o.BeliefPropagationCache(psi), maxiter=60, tolerance=None)
    gate = o.gate_matrix("Rxx", 0.7) @ np.kron(o.gate_matrix("Rz", 0.3), o.gate_matrix("Rx", 0.2))
    for (a, b) in g.edges[:5]:
        c1, c2 = bpc.copy(), bpc.copy()
        e1 = o.apply_gate(c1, gate, [a, b], maxdim=maxdim, cutoff=1e-14, normalize_tensors=True)
        e2 = gr.apply_gate_gram(c2, gate, a, b, maxdim=maxdim, cutoff=1e-14, normalize_tensors=True)
        assert abs(e1 - e2) < 1e-12
        assert np.max(np.abs(np.diag(c1.messages[(a, b)]) - np.diag(c2.messages[(a, b)]))) < 1e-12      # S
        v1, v2 = sv.tns_to_statevector(c1.tns), sv.tns_to_statevector(c2.tns)
        assert sv.fidelity(v1, v2) > 1 - 1e-11                                                            # same state
        assert abs(np.vdot(v1, v1).real - np.vdot(v2, v2).real) < 1e-10 * np.vdot(v1, v1).real


def test_rank_deficient_and_wide_cases():
    """product state (bond dimension 1, rank-deficient Gram matrices, wide theta) and a corner site"""
    g = o.named_grid((3, 3))
    bpc = o.update(o.BeliefPropagationCache(o.product_state(np.complex128, lambda v: "↑", g)))
    layer = [("Rx", [v], 0.5) for v in g.vertices]
    bpc, _ = o.apply_gates(layer, bpc, update_cache=False, apply_kwargs=dict(normalize_tensors=False))
    gate = o.gate_matrix("Rzz", 0.25)
    c1, c2 = bpc.copy(), bpc.copy()
    for (a, b) in [((1, 1), (2, 1)), ((2, 1), (2, 2)), ((2, 2), (3, 2))]:
        e1 = o.apply_gate(c1, gate, [a, b], cutoff=1e-12, normalize_tensors=False)
        e2 = gr.apply_gate_gram(c2, gate, a, b, cutoff=1e-12, normalize_tensors=False)
        assert abs(e1 - e2) < 1e-13
        assert c1.tns.bond_dim(a, b) == c2.tns.bond_dim(a, b)
    assert sv.fidelity(sv.tns_to_statevector(c1.tns), sv.tns_to_statevector(c2.tns)) > 1 - 1e-12
