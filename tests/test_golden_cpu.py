"""CPU: the oracle reproduces the committed golden fixtures (guards against oracle / fixture drift), and the
K-level unit vectors (truncation rule, pseudo square root, message_diff)."""
import numpy as np
import pytest

import tnqs_oracle as o
from golden_util import load, layer_from_meta, TFIM_CASES, BP_CASES

Z = np.diag([1.0, -1.0]).astype(complex)


@pytest.mark.parametrize("name", ["tfim3x3_c128_maxdim2", "tfim3x3_c64_maxdim3", "heavyhex11_c128_maxdim4", "tfim4x4_c64_maxdim8_12layers"])
def test_oracle_reproduces_tfim_fixture(name):
    meta, z = load(name)
    g = o.Graph(meta["vertices"], meta["edges"])
    layer, seq = layer_from_meta(meta)
    dtype = np.dtype(meta["dtype"])
    bpkw = dict(edge_sequence=seq, maxiter=meta["sweeps"], tolerance=None)
    kw = dict(maxdim=meta["maxdim"], cutoff=meta["cutoff"], normalize_tensors=meta["normalize"])
    bpc = o.update(o.BeliefPropagationCache(o.product_state(dtype, lambda v: "↑", g)), **bpkw)
    tol = 1e-12 if dtype == np.complex128 else 1e-5
    for l in range(meta["nlayers"]):
        bpc, errs = o.apply_gates(layer, bpc, apply_kwargs=kw, bp_update_kwargs=bpkw)
        assert np.max(np.abs(errs - z["errs"][l])) < tol
        ez = np.array([o.expect_1site(bpc, Z, v) for v in g.vertices])
        assert np.max(np.abs(ez - z["expZ"][l])) < 100 * tol
        assert np.array_equal(np.array([bpc.tns.bond_dim(a, b) for (a, b) in g.edges]), z["bond_dims"][l])


@pytest.mark.parametrize("name", BP_CASES)
def test_oracle_reproduces_bp_fixture(name):
    meta, z = load(name)
    g = o.Graph(meta["vertices"], meta["edges"])
    psi = o.TensorNetworkState(g, {v: z[f"psi_{i}"] for i, v in enumerate(g.vertices)})
    tol = 1e-12 if meta["dtype"] == "complex128" else 1e-5
    for ns in (1, 2, 5):
        bpc = o.update(o.BeliefPropagationCache(psi), maxiter=ns, tolerance=None, edge_sequence=meta["seq"])
        got = np.concatenate([bpc.message(e).reshape(-1) for e in meta["seq"]])
        assert np.max(np.abs(got - z[f"msgs_{ns}"])) < tol
    if "regions" in meta:
        X = np.array([[0, 1], [1, 0.0]], dtype=complex)
        bpc = o.update(o.BeliefPropagationCache(psi), maxiter=60, tolerance=None, edge_sequence=meta["seq"])
        for r, ref in zip(meta["regions"], z["region_vals"]):
            assert o.steiner_vertices(g, r["vertices"]) == r["steiner"]
            assert abs(o.expect(bpc, {v: {"Z": Z, "X": X}[c] for v, c in zip(r["vertices"], r["ops"])}) - ref) < 1e-12


def test_unit_vectors():
    _, z = load("unit_vectors")
    for row in z["trunc_cases"]:
        i, md, co, n, e = int(row[0]), int(row[1]), float(row[2]), int(row[3]), float(row[4])
        got = o.truncate_spectrum(z[f"trunc_spec_{i}"], None if md < 0 else md, None if co < 0 else co)
        assert got[0] == n and abs(got[1] - e) < 1e-15
    ms, mi = o.pseudo_sqrt_inv_sqrt(z["sqrt_in"], 2.2e-15)
    assert np.max(np.abs(ms - z["sqrt_out"])) < 1e-13 and np.max(np.abs(mi - z["invsqrt_out"])) < 1e-10
    assert np.max(np.abs(ms @ ms - z["sqrt_in"])) < 1e-13                      # (M^1/2)^2 == M
    proj = ms @ mi
    assert np.max(np.abs(proj @ proj - proj)) < 1e-10                           # M^1/2 M^-1/2 is a projector (rank 3 of 5)
    assert abs(np.trace(proj).real - 3) < 1e-10
    assert abs(o.message_diff(z["diff_a"], z["diff_b"]) - float(z["diff_val"])) < 1e-15
