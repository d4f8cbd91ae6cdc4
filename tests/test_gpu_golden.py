"""GPU: the HIP path (through the C ABI) against the committed golden fixtures (tests/golden/*.npz).  These tests do
not need the oracle at run time.  Tolerances: complex128 1e-8 (<Z>, spectra, truncation errors), complex64 1e-5 per layer (the north star's bound on expectation values) / truncation errors 2e-3 relative."""
import numpy as np
import pytest

import tnqs_amd as tn
from golden_util import load, layer_from_meta, TFIM_CASES, BP_CASES
from helpers import c64_errs_close

pytestmark = pytest.mark.gpu


def contract_statevector(tns):
    g = tns.graph
    n = g.nv()
    lab = {}
    nxt = n
    for (a, b) in g.edges:
        lab[frozenset((a, b))] = nxt
        nxt += 1
    args = []
    for i, v in enumerate(g.vertices):
        args += [np.asarray(tns.tensors[v], dtype=complex), [i] + [lab[frozenset((v, w))] for w in g.neighbors(v)]]
    args.append(list(range(n)))
    return np.einsum(*args, optimize="greedy").reshape(-1)


@pytest.mark.parametrize("name", TFIM_CASES)
def test_apply_gates_matches_golden(name):
    meta, z = load(name)
    g = tn.NamedGraph(meta["vertices"], meta["edges"])
    layer, seq = layer_from_meta(meta)
    dtype = np.dtype(meta["dtype"])
    bpkw = dict(edge_sequence=seq, maxiter=meta["sweeps"], tolerance=None)
    kw = dict(maxdim=meta["maxdim"], cutoff=meta["cutoff"], normalize_tensors=meta["normalize"])
    bpc = tn.update(tn.BeliefPropagationCache(tn.tensornetworkstate(dtype, lambda v: "↑", g)), **bpkw)
    c128 = dtype == np.complex128
    for l in range(meta["nlayers"]):
        info = {}
        bpc, errs = tn.apply_gates(layer, bpc, apply_kwargs=kw, bp_update_kwargs=bpkw, info=info)
        assert info["n_updates"] == len(meta["groups"]) + 1
        assert np.array_equal(np.array([bpc.bond_dim(a, b) for (a, b) in g.edges]), z["bond_dims"][l]), l
        assert (np.max(np.abs(errs - z["errs"][l])) < 1e-9) if c128 else c64_errs_close(errs, z["errs"][l]), l
        scale = (l + 1) * 1e-8 if c128 else 1e-5          # ComplexF32: a FLAT 1e-5 on every layer (the north star's bound; the device measures 1-7e-6 over ten layers)
        ez = tn.expect_all(bpc, "Z")
        assert np.max(np.abs(ez - z["expZ"][l])) < scale, (l, np.max(np.abs(ez - z["expZ"][l])))
        sp = []
        for (a, b) in g.edges:
            for e in ((a, b), (b, a)):
                m = bpc.message(e).astype(np.complex128)
                w = np.linalg.eigvalsh((m + m.conj().T) / 2)
                sp.append(w / w.sum())
        assert np.max(np.abs(np.concatenate(sp) - z[f"spectra_{l}"])) < scale, l
    if meta["exact"]:
        vec = contract_statevector(bpc.network())
        ref = z["statevector"]
        fid = abs(np.vdot(vec, ref)) ** 2 / (np.vdot(vec, vec).real * np.vdot(ref, ref).real)
        assert fid > 1 - 1e-9
        assert abs(np.vdot(vec, vec).real - 1) < 1e-9            # test/test_apply.jl:20,53


@pytest.mark.parametrize("name", BP_CASES)
def test_bp_update_matches_golden(name):
    meta, z = load(name)
    g = tn.NamedGraph(meta["vertices"], meta["edges"])
    psi = tn.TensorNetworkState(g, {v: z[f"psi_{i}"] for i, v in enumerate(g.vertices)})
    tol = 1e-10 if meta["dtype"] == "complex128" else 1e-5
    bpc = tn.BeliefPropagationCache(psi)
    for ns in (1, 2, 5):
        out = tn.update(bpc, maxiter=ns, tolerance=None, edge_sequence=meta["seq"])
        got = np.concatenate([out.message(e).reshape(-1) for e in meta["seq"]])
        ref = z[f"msgs_{ns}"]
        assert np.max(np.abs(got - ref)) < tol * np.max(np.abs(ref)), ns
    out = tn.update(bpc, maxiter=60, tolerance=None, edge_sequence=meta["seq"])
    assert np.max(np.abs(tn.expect_all(out, "Z") - z["expZ"])) < 100 * tol
    for r, ref in zip(meta.get("regions", []), z["region_vals"] if "region_vals" in z.files else []):      # multi-site observables (ties, loops)
        region, _ = tn.steiner_region(g, r["vertices"])
        assert sorted(region, key=g.index.__getitem__) == r["steiner"]
        assert abs(tn.expect(out, (r["ops"], r["vertices"])) - ref) < 100 * tol * max(1.0, abs(ref)), r
    if "expZ_exact" in z.files:                                  # BP is exact on trees (test/test_expect.jl:26-28)
        assert np.max(np.abs(tn.expect_all(out, "Z") - z["expZ_exact"])) < 100 * tol
