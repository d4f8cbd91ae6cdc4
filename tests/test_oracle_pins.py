"""Pin the CPU oracle against every known-answer / invariant test the reference holds for the hot path
(SURVEY.md 8c).  CPU only.  Each test names the reference test it restates."""
import math

import numpy as np
import pytest

import tnqs_oracle as o
import statevector as sv

Z = np.diag([1.0, -1.0]).astype(complex)
TIGHT = dict(maxiter=300, tolerance=1e-14)


def tfim_layer(g, dt=0.25, hx=1.0, hz=0.8, J=0.5):
    """layer of test/test_apply.jl:30-47"""
    layer = [("Rx", [v], 2 * hx * dt) for v in g.vertices]
    layer += [("Rz", [v], 2 * hz * dt) for v in g.vertices]
    for grp in o.edge_color(g):
        layer += [("Rzz", [a, b], 2 * J * dt) for (a, b) in grp]
    return layer


def test_custom_circuit_norm_and_bondcap():
    """test/test_apply.jl:11-20: Rx,Rx,CPHASE from "down-down", maxdim=2, cutoff=1e-10, no normalisation."""
    circuit = [("Rx", [(1, 1)], 0.5), ("Rx", [(2, 1)], 0.2), ("CPHASE", [(1, 1), (2, 1)], -0.3)]
    g = o.Graph([(1, 1), (2, 1)], [((1, 1), (2, 1))])
    psi0 = o.product_state(np.complex64, lambda v: "↓", g)
    bpc = o.update(o.BeliefPropagationCache(psi0))
    bpc, errs = o.apply_gates(circuit, bpc, apply_kwargs=dict(maxdim=2, cutoff=1e-10, normalize_tensors=False))
    assert bpc.tns.dtype == np.complex64
    assert bpc.tns.maxvirtualdim() <= 2
    vec = sv.tns_to_statevector(bpc.tns)
    assert abs(np.vdot(vec, vec).real - 1) < 1e-5
    ref = sv.run_circuit_statevector(g, {v: [0, 1] for v in g.vertices}, circuit)
    assert sv.fidelity(vec, ref) > 1 - 1e-5


@pytest.mark.parametrize("dtype,tol", [(np.complex64, 2e-5), (np.complex128, 1e-12)])
def test_tfim_layer_3x3_exact_without_truncation(dtype, tol):
    """test/test_apply.jl:23-53 + simple_update.jl:4 ("exact if no truncation is performed")."""
    g = o.named_grid((3, 3))
    psi0 = o.random_state(dtype, g, chi=1, seed=123)
    bpc = o.rescale(o.update(o.BeliefPropagationCache(psi0)))            # normalize(psi0; alg="bp")
    layer = tfim_layer(g)
    vecs = {v: np.asarray(bpc.tns.tensors[v]).reshape(2) for v in g.vertices}
    bpc, errs = o.apply_gates(layer, bpc, apply_kwargs=dict(cutoff=1e-10 if dtype == np.complex64 else 1e-20,
                                                            normalize_tensors=False))
    assert bpc.tns.dtype == dtype
    assert bpc.tns.maxvirtualdim() <= 2
    vec = sv.tns_to_statevector(bpc.tns)
    assert abs(np.vdot(vec, vec).real - 1) < tol
    ref = sv.run_circuit_statevector(g, vecs, layer)
    assert np.max(np.abs(vec - ref * np.vdot(ref, vec) / abs(np.vdot(ref, vec)))) < 10 * tol
    assert np.all(errs < 1e-6)


def test_two_layers_exact_complex128():
    g = o.named_grid((3, 3))
    psi0 = o.product_state(np.complex128, lambda v: "↑", g)
    bpc = o.update(o.BeliefPropagationCache(psi0))
    layer = tfim_layer(g)
    for _ in range(2):
        bpc, errs = o.apply_gates(layer, bpc, apply_kwargs=dict(cutoff=1e-24, normalize_tensors=False),
                                  bp_update_kwargs=TIGHT)
    vec = sv.tns_to_statevector(bpc.tns)
    ref = sv.run_circuit_statevector(g, {v: [1, 0] for v in g.vertices}, layer + layer)
    assert abs(np.vdot(vec, vec).real - 1) < 1e-11
    assert sv.fidelity(vec, ref) > 1 - 1e-11
    # the two bond messages written by apply_gate! are diag(S) (apply_gates.jl:126-135)
    for (a, b) in g.edges[:3]:
        m = bpc.message((a, b))
        assert m.shape[0] == m.shape[1]


@pytest.mark.parametrize("dtype", [np.float64, np.complex64, np.complex128])
def test_bp_exact_on_trees(dtype):
    """test/test_beliefpropagation.jl:9-56: comb tree (3,3): partition function == exact norm, RDM bp == exact."""
    g = o.comb_tree((3, 3))
    psi = o.random_state(dtype, g, chi=2, seed=7)
    bpc = o.update(o.BeliefPropagationCache(psi))                          # trees: maxiter=1, no tolerance
    assert len(bpc.messages) == 2 * len(g.edges)                           # :54
    vec = sv.tns_to_statevector(psi)
    z_exact = np.vdot(vec, vec).real
    z_bp = o.partitionfunction(bpc)
    eps = np.finfo(np.zeros(1, dtype=dtype).real.dtype).eps
    assert abs(z_bp - z_exact) / z_exact < 200 * eps
    for v in g.vertices[:4]:
        rho = o.rdm_1site(bpc, v)
        rho = rho / np.trace(rho)
        assert np.max(np.abs(rho - sv.rdm_statevector(vec, g, v))) < 200 * eps


def test_expect_bp_exact_on_line_and_not_on_loop():
    """test/test_expect.jl:19-44"""
    g = o.named_grid((6,))
    psi = o.random_state(np.complex128, g, chi=3, seed=11)
    bpc = o.update(o.BeliefPropagationCache(psi))
    vec = sv.tns_to_statevector(psi)
    for v in g.vertices:
        assert abs(o.expect_1site(bpc, Z, v) - sv.expect_statevector(vec, g, Z, v)) < 1e-12
    g2 = o.named_grid((3, 3))
    psi2 = o.random_state(np.complex128, g2, chi=2, seed=5)
    bpc2 = o.update(o.BeliefPropagationCache(psi2), **TIGHT)
    vec2 = sv.tns_to_statevector(psi2)
    d = abs(o.expect_1site(bpc2, Z, (2, 2)) - sv.expect_statevector(vec2, g2, Z, (2, 2)))
    assert d > 1e-6


def test_ghz_bond_entropy_is_log2():
    """test/test_constructors.jl:69-74"""
    g = o.named_grid((3, 3))
    tensors = {}
    for v in g.vertices:
        t = np.zeros((2,) + (2,) * g.degree(v), dtype=np.complex128)
        t[(0,) * t.ndim] = 1
        t[(1,) * t.ndim] = 1
        tensors[v] = t
    bpc = o.update(o.BeliefPropagationCache(o.TensorNetworkState(g, tensors)), **TIGHT)
    for e in g.edges[:4]:
        assert abs(o.bond_entropy(bpc, e) - math.log(2)) < 1e-9


def test_truncate_respects_maxdim_and_fidelity():
    """test/test_truncate.jl:12-35 (BP branch)"""
    g = o.named_hexagonal_lattice_graph(2, 2)
    psi = o.random_state(np.complex128, g, chi=3, seed=3)
    bpc = o.rescale(o.update(o.BeliefPropagationCache(psi), **TIGHT))
    t = o.truncate(bpc, maxdim=2, cutoff=1e-10, bp_update_kwargs=TIGHT)
    assert t.tns.maxvirtualdim() <= 2
    a, b = sv.tns_to_statevector(bpc.tns), sv.tns_to_statevector(t.tns)
    f = sv.fidelity(a, b)
    assert 0 <= f <= 1 + 1e-12


def test_truncation_rule_edge_cases():
    """NDTensors truncate! restatement"""
    p = np.array([0.5, 0.3, 0.15, 0.05])
    assert o.truncate_spectrum(p, None, None) == (4, 0.0)
    n, e = o.truncate_spectrum(p, 2, None)
    assert n == 2 and abs(e - 0.2) < 1e-15
    n, e = o.truncate_spectrum(p, None, 0.06)
    assert n == 3 and abs(e - 0.05) < 1e-15
    n, e = o.truncate_spectrum(p, 3, 0.25)
    assert n == 2 and abs(e - 0.2) < 1e-15
    assert o.truncate_spectrum(np.array([1.0]), 1, 0.5) == (1, 0.0)
    n, e = o.truncate_spectrum(np.array([1.0, 0.0, 0.0]), None, None)
    assert n == 1 and e == 0.0
    n, e = o.truncate_spectrum(np.array([1.0, 1e-3]), None, 1.0)      # mindim = 1
    assert n == 1


def test_unitary_gate_leaves_messages_at_fixed_point():
    """without truncation the post-gate update converges in one sweep."""
    g = o.named_grid((3, 3))
    bpc = o.update(o.BeliefPropagationCache(o.product_state(np.complex128, lambda v: "↑", g)))
    info = {}
    bpc, _ = o.apply_gates(tfim_layer(g), bpc, apply_kwargs=dict(cutoff=1e-24, normalize_tensors=False), info=info)
    assert info["n_updates"] == len(o.edge_color(g)) + 1
    assert all(s == 1 for s in info["sweeps"])


def test_apply_gate_errors():
    """apply_gates.jl:109-120"""
    g = o.named_grid((3, 3))
    bpc = o.BeliefPropagationCache(o.product_state(np.complex128, lambda v: "↑", g))
    with pytest.raises(RuntimeError):
        o.apply_gate(bpc, np.eye(4), [(1, 1), (3, 3)])
    with pytest.raises(RuntimeError):
        o.apply_gate(bpc, np.eye(8), [(1, 1), (2, 1), (3, 1)])


def test_multi_site_expect_bp_exact_on_trees():
    """expect(alg"bp") with several vertices (expect.jl:59-82): the region contraction is exact on trees, so it must equal
    the state-vector value (the known answer test_expect.jl:26-28 uses for single sites)."""
    Zm = np.diag([1.0, -1.0]); Xm = np.array([[0, 1], [1, 0.0]])
    for g, pairs in ((o.named_grid((5,)), [(((1,), (2,)), [(1,), (2,)]), (((1,), (4,)), [(1,), (2,), (3,), (4,)])]),
                     (o.comb_tree((3, 3)), [(((1, 1), (1, 2)), [(1, 1), (1, 2)]), (((1, 2), (3, 1)), [(1, 2), (1, 1), (2, 1), (3, 1)])])):
        psi = o.random_state(np.complex128, g, 3, seed=1)
        bpc = o.update(o.BeliefPropagationCache(psi))
        v = sv.tns_to_statevector(psi)
        for (a, b), region in pairs:
            val = o.expect_region(bpc, {a: Zm, b: Xm}, region)
            assert abs(val - sv.expect_statevector_multi(v, g, {a: Zm, b: Xm})) < 1e-12


def test_symmetric_gauge_invariants():
    """symmetric_gauge (symmetric_gauge.jl:1-62): the state is unchanged, both messages of every edge become the same diagonal
    matrix S, and that is a BP fixed point -- for COMPLEX messages this only holds with ITensors.eigen's index convention
    (functions of the message enter transposed), which is what the restatement pins."""
    for g in (o.named_grid((3, 3)), o.comb_tree((3, 3))):
        psi = o.random_state(np.complex128, g, 3, seed=4)
        bpc = o.update(o.BeliefPropagationCache(psi), maxiter=300, tolerance=1e-15)
        sg = o.symmetric_gauge(bpc)
        v0, v1 = sv.tns_to_statevector(psi), sv.tns_to_statevector(sg.tns)
        assert abs(sv.fidelity(v0, v1) - 1) < 1e-12
        up = o.update(sg, maxiter=1, tolerance=None)
        for (a, b) in g.edges:
            m = sg.message((a, b))
            assert np.allclose(m, np.diag(np.diag(m))) and np.allclose(m, sg.message((b, a)))
            for d in ((a, b), (b, a)):
                m0, m1 = sg.message(d), up.message(d)
                assert np.max(np.abs(m0 / np.trace(m0) - m1 / np.trace(m1))) < 1e-6


def test_htse_known_answer():
    """examples/hexagonal_heisenbergmodel_thermalstate.jl:36: the BP free-energy density of the imaginary-time evolved
    identity reproduces the 4th-order high-temperature series -ln 2 - 9/64 b^2 - 3/128 b^3 + 27/2048 b^4; what is left is
    the next series order (observed 2e-8 at beta = 0.1 ... 1.8e-4 at beta = 0.5, i.e. < 0.01 beta^5)."""
    from helpers import htse_free_energy
    res = htse_free_energy(o)
    assert [round(b, 6) for (b, _, _) in res] == [0.1, 0.2, 0.3, 0.4, 0.5]
    for b, f, f4 in res:
        assert abs(f - f4) < 0.01 * b ** 5


def test_pseudo_sqrt_casts_eigen_factors_to_the_message_precision_first():
    """src/utils.jl:100-107 (`safe_eigen`: D, U back to Float32) then :20-25 (cutoff test, sqrt, Q D Q^dagger in Float32): an eigenvalue
    that is below the cutoff in f64 but rounds ONTO it in f32 is kept by the reference (`abs(x) < cutoff` is false), one a little lower
    is dropped.  Everything downstream of the test runs in f32: the result has f32-level, not f64-level, residuals."""
    rng = np.random.default_rng(5)
    n = 6
    q, _ = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
    cut = float(np.float32(10 * np.finfo(np.float32).eps))
    for lam_small, kept in ((cut * (1 - 2e-8), True), (cut * (1 - 3e-7), False)):
        assert (np.float32(lam_small) < np.float32(cut)) == (not kept)
        w = np.array([1.0, 0.5, 0.2, 0.1, 0.01, lam_small])
        # the test matrix is built in f64 and handed over as complex64: the eigenvalues move by ~1e-8 -- far more than the 2e-8 relative
        # margin around the cutoff -- so the expected side is read off the f32-rounded eigenvalue of the ROUNDED matrix
        m = ((q * w) @ q.conj().T).astype(np.complex64)
        w32 = np.linalg.eigvalsh(m.astype(np.complex128)).astype(np.float32)
        expect_rank = int(np.count_nonzero(~((w32 == 0) | (np.abs(w32) < np.float32(cut)))))
        ms, mi = o.pseudo_sqrt_inv_sqrt(m, cut)
        assert ms.dtype == np.complex64 and mi.dtype == np.complex64
        proj = ms.astype(np.complex128) @ mi.astype(np.complex128)
        assert int(round(np.trace(proj).real)) == expect_rank
        w64, q64 = np.linalg.eigh(m.astype(np.complex128))
        keep = ~((w64.astype(np.float32) == 0) | (np.abs(w64.astype(np.float32)) < np.float32(cut)))
        m_kept = (q64[:, keep] * w64[keep]) @ q64[:, keep].conj().T
        res = np.linalg.norm(ms.astype(np.complex128) @ ms.astype(np.complex128) - m_kept)
        assert 1e-9 < res < 1e-5          # f32 arithmetic throughout (products of f32-rounded Q and sqrt(D)), not f64 cast at the end
