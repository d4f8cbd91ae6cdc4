#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ from the CPU oracle (oracle/tnqs_oracle.py) and the independent
state-vector simulator (oracle/statevector.py).

The Julia reference cannot run in the build container (no julia binary / package depot) and ships no golden numbers,
so these vectors are RESTATEMENT-GENERATED; where no truncation occurs they are cross-checked here against the exact
state vector before being written.  Only gauge-invariant quantities are stored: truncation errors,
singular values (bond messages right after a gate are diag(S)), trace-normalised message spectra, <Z_v>, bond
dimensions, exact state vectors.  Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import tnqs_oracle as o          # noqa: E402
import statevector as sv         # noqa: E402

Z = np.diag([1.0, -1.0]).astype(complex)


def colour_sequence(groups):
    seq = []
    for grp in groups:
        seq += list(grp) + [(b, a) for (a, b) in grp]
    return seq


def spectra(bpc):
    out = []
    for (a, b) in bpc.g.edges:
        for e in ((a, b), (b, a)):
            m = bpc.message(e).astype(np.complex128)
            w = np.linalg.eigvalsh((m + m.conj().T) / 2)
            out.append(w / w.sum())
    return out


def tfim_case(name, g, groups, dtype, maxdim, cutoff, nlayers, rx, rz, rzz, normalize, sweeps=30, exact=False):
    layer = [("Rx", [v], rx) for v in g.vertices]
    if rz is not None:
        layer += [("Rz", [v], rz) for v in g.vertices]
    for grp in groups:
        layer += [("Rzz", [a, b], rzz) for (a, b) in grp]
    seq = colour_sequence(groups)
    bpkw = dict(edge_sequence=seq, maxiter=sweeps, tolerance=None)
    kw = dict(maxdim=maxdim, cutoff=cutoff, normalize_tensors=normalize)
    bpc = o.update(o.BeliefPropagationCache(o.product_state(dtype, lambda v: "↑", g)), **bpkw)
    rec = {"errs": [], "expZ": [], "bond_dims": [], "spectra": []}
    for _ in range(nlayers):
        bpc, errs = o.apply_gates(layer, bpc, apply_kwargs=kw, bp_update_kwargs=bpkw)
        rec["errs"].append(errs)
        rec["expZ"].append(np.array([o.expect_1site(bpc, Z, v) for v in g.vertices]))
        rec["bond_dims"].append(np.array([bpc.tns.bond_dim(a, b) for (a, b) in g.edges]))
        sp = spectra(bpc)
        rec["spectra"].append(np.concatenate(sp))
    out = {f"{k}": np.array(v) if k != "spectra" else np.array(v, dtype=object) for k, v in rec.items()}
    arrays = {"errs": np.array(rec["errs"]), "expZ": np.array(rec["expZ"]), "bond_dims": np.array(rec["bond_dims"])}
    for i, sp in enumerate(rec["spectra"]):
        arrays[f"spectra_{i}"] = sp
    if exact:
        vec = sv.tns_to_statevector(bpc.tns)
        ref = sv.run_circuit_statevector(g, {v: [1, 0] for v in g.vertices}, layer * nlayers)
        assert sv.fidelity(vec, ref) > 1 - 1e-10, "oracle disagrees with the exact state vector"
        arrays["statevector"] = ref.reshape(-1)
    meta = dict(name=name, vertices=[list(map(float, v)) for v in g.vertices], edges=[[list(map(float, a)), list(map(float, b))] for (a, b) in g.edges],
                groups=[[[list(map(float, a)), list(map(float, b))] for (a, b) in grp] for grp in groups], dtype=np.dtype(dtype).name,
                maxdim=maxdim, cutoff=cutoff, nlayers=nlayers, rx=rx, rz=rz, rzz=rzz, normalize=normalize, sweeps=sweeps, exact=exact)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), meta=json.dumps(meta), **arrays)
    print("wrote", name, {k: v.shape for k, v in arrays.items() if not k.startswith("spectra")})


def bp_case(name, g, chi, dtype, seed, regions=None):
    psi = o.random_state(dtype, g, chi, seed=seed)
    seq = o.forest_cover_edge_sequence(g)
    arrays = {"psi_" + str(i): psi.tensors[v] for i, v in enumerate(g.vertices)}
    for ns in (1, 2, 5):
        bpc = o.update(o.BeliefPropagationCache(psi), maxiter=ns, tolerance=None, edge_sequence=seq)
        arrays[f"msgs_{ns}"] = np.concatenate([bpc.message(e).reshape(-1) for e in seq])
    bpc = o.update(o.BeliefPropagationCache(psi), maxiter=60, tolerance=None, edge_sequence=seq)
    arrays["expZ"] = np.array([o.expect_1site(bpc, Z, v) for v in g.vertices])
    if g.is_tree():
        vec = sv.tns_to_statevector(psi)
        ex = np.array([sv.expect_statevector(vec, g, Z, v) for v in g.vertices])
        assert np.max(np.abs(ex - arrays["expZ"])) < 1e-10       # BP is exact on trees (test/test_expect.jl:26-28)
        arrays["expZ_exact"] = ex
    meta = dict(name=name, vertices=[list(map(float, v)) for v in g.vertices], edges=[[list(map(float, a)), list(map(float, b))] for (a, b) in g.edges],
                seq=[[list(map(float, a)), list(map(float, b))] for (a, b) in seq], chi=chi, dtype=np.dtype(dtype).name, seed=seed)
    if regions:
        # multi-site observables at the 60-sweep fixed point (expect.jl:59-82): the Steiner tree of the support (recalled rule, graphs.py /
        # o.steiner_vertices) and the dense contraction of the induced region; (ops, vertices) pairs of Pauli letters
        mats = {"Z": Z, "X": np.array([[0, 1], [1, 0.0]], dtype=complex)}
        vals, rmeta = [], []
        for ops, vs in regions:
            st = o.steiner_vertices(g, vs)
            vals.append(o.expect_region(bpc, {v: mats[c] for v, c in zip(vs, ops)}, st))
            rmeta.append(dict(ops=ops, vertices=[list(map(float, v)) for v in vs], steiner=[list(map(float, v)) for v in st]))
        arrays["region_vals"] = np.array(vals)
        meta["regions"] = rmeta
    np.savez_compressed(os.path.join(HERE, name + ".npz"), meta=json.dumps(meta), **arrays)
    print("wrote", name)


def unit_vectors():
    rng = np.random.default_rng(42)
    out = {}
    # truncation rule (NDTensors truncate!): rows = (spectrum id, maxdim or -1, cutoff or -1, n_keep, truncerr)
    cases = []
    specs = [np.array([0.5, 0.3, 0.15, 0.05]), np.array([1.0]), np.array([1.0, 0.0, 0.0]), np.array([0.9, 0.1, 1e-11, 1e-13]),
             np.array([0.25, 0.25, 0.25, 0.25]), np.sort(rng.random(12))[::-1]]
    for i, p in enumerate(specs):
        out[f"trunc_spec_{i}"] = p
        for md in (None, 1, 2, 3, 8):
            for co in (None, 0.0, 1e-10, 0.06, 0.3):
                n, e = o.truncate_spectrum(p, md, co)
                cases.append([i, -1 if md is None else md, -1.0 if co is None else co, n, e])
    out["trunc_cases"] = np.array(cases)
    # pseudo sqrt / inverse sqrt (src/utils.jl:18-27) incl. a rank-deficient message
    b = rng.standard_normal((5, 3)) + 1j * rng.standard_normal((5, 3))
    m = b @ b.conj().T
    m /= np.trace(m).real
    ms, mi = o.pseudo_sqrt_inv_sqrt(m, 2.2e-15)
    out["sqrt_in"], out["sqrt_out"], out["invsqrt_out"] = m, ms, mi
    # message_diff (beliefpropagationcache.jl:17-21)
    a, c = rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4)), rng.standard_normal((4, 4)) + 1j * rng.standard_normal((4, 4))
    out["diff_a"], out["diff_b"], out["diff_val"] = a, c, np.array(o.message_diff(a, c))
    np.savez_compressed(os.path.join(HERE, "unit_vectors.npz"), **out)
    print("wrote unit_vectors")


if __name__ == "__main__":
    g33 = o.named_grid((3, 3))
    grp33 = o.edge_color(g33)
    tfim_case("tfim3x3_c128_exact", g33, grp33, np.complex128, None, 1e-24, 2, 0.5, 0.4, 0.25, False, exact=True)
    tfim_case("tfim3x3_c128_maxdim2", g33, grp33, np.complex128, 2, 1e-10, 3, 0.5, 0.4, 0.25, True)
    tfim_case("tfim3x3_c128_maxdim4", g33, grp33, np.complex128, 4, 1e-10, 3, 0.5, 0.4, 0.25, True)
    tfim_case("tfim3x3_c64_maxdim3", g33, grp33, np.complex64, 3, 1e-10, 3, 0.5, 0.4, 0.25, True)
    hh = o.heavy_hexagonal_lattice(1, 1)
    tfim_case("heavyhex11_c128_maxdim4", hh, o.edge_color(hh), np.complex128, 4, 1e-12, 3, 0.4, None, np.pi / 2, True)
    cub = o.named_grid((2, 2, 3))
    tfim_case("cubic2x2x3_c128_maxdim2", cub, o.edge_color(cub), np.complex128, 2, 1e-10, 2, 0.3, 0.2, 0.4, True)
    tor = o.named_grid((3, 3, 3), periodic=True)
    tfim_case("cubic3x3x3p_c128_maxdim2", tor, o.edge_color(tor), np.complex128, 2, 1e-10, 1, 0.3, None, 0.4, True, sweeps=20)
    bp_case("bp_comb33_chi2_c128", o.comb_tree((3, 3)), 2, np.complex128, 7)
    bp_case("bp_grid3x3_chi3_c128", g33, 3, np.complex128, 5)
    bp_case("bp_grid3x3_chi3_c64", g33, 3, np.complex64, 5)
    # round 4: an evolution small enough to commit (the chi = 32 version runs device-vs-oracle in tests/test_gpu_fullsize.py): 4x4, ComplexF32, twelve
    # TFIM layers at dt = 0.1, maxdim 8 -- truncation live from the fourth layer on; and multi-site observables on regions with a tie / a loop
    g44 = o.named_grid((4, 4))
    tfim_case("tfim4x4_c64_maxdim8_12layers", g44, o.edge_color(g44), np.complex64, 8, 1e-10, 12, 0.5, None, 0.2, True, sweeps=4)
    bp_case("bp_grid4x4_chi4_c128_regions", g44, 4, np.complex128, 29,
            regions=[("ZZ", [(1, 1), (2, 2)]), ("ZX", [(2, 2), (3, 3)]), ("ZZZZ", [(2, 2), (2, 3), (3, 3), (3, 2)]), ("ZXZ", [(1, 1), (1, 2), (2, 2)]), ("ZX", [(1, 2), (3, 3)])])
    unit_vectors()
