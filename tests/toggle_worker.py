"""child process of tests/test_gpu_toggles.py: one fixed circuit, results as JSON on the last stdout line.  The library reads its
TNQS_NO_* switches once per process, so every variant needs its own process."""
import json
import sys

import numpy as np

import tnqs_amd as tn


def cubic16():
    """3x3x3 periodic cubic lattice at chi = 16 (the per-site shape of BASELINE configs[3]: 27 tensors of 268 MB): two BP sweeps in the
    library's default order and one layer -- the 16 x 16 plane kernels (two legs per pass, both messages of a forest per pass) against
    the single-leg route (TNQS_NO_PAIR=1), which the oracle tests cover"""
    g = tn.named_grid((3, 3, 3), periodic=True)
    chi = 16
    bpc = tn.BeliefPropagationCache(tn.tensornetworkstate(np.complex64, lambda v: "↑", g))
    rng = np.random.default_rng(7)
    for v in g.vertices:
        shp = (2,) + (chi,) * g.degree(v); n = int(np.prod(shp))
        bpc._set_tensor(v, rng.standard_normal(2 * n, dtype=np.float32).view(np.complex64).reshape(shp) / np.float32(np.sqrt(n)))
    tn.profile_enable(bpc, True)
    bpc = tn.update(bpc, maxiter=2, tolerance=None)
    prof = tn.profile_get(bpc)
    msgs = [bpc.message(e) for (a, b) in g.edges[:20] for e in ((a, b), (b, a))]
    layer = [("Rz", [v], -0.04) for v in g.vertices] + [("Rxx", [a, b], -0.08) for grp in tn.edge_color(g) for (a, b) in grp]
    b2, errs = tn.apply_gates(layer, bpc, apply_kwargs=dict(maxdim=chi, cutoff=1e-10, normalize_tensors=True), bp_update_kwargs=dict(maxiter=2, tolerance=None))
    out = dict(msgs=[[m.real.tolist(), m.imag.tolist()] for m in msgs], errs=errs.tolist(), z=[float(np.real(x)) for x in tn.expect_all(b2, "Z")],
               dims=[b2.bond_dim(a, b) for a, b in g.edges], pairgram=prof["bp_pairgram"]["launches"], pair=prof["bp_pair"]["launches"],
               pair_passes=prof["bp_pair"]["flops"] / (27 * 2 * 8.0 * 2 * chi ** 6 * chi))       # two-leg passes per site over the two sweeps (a pass = two mode products of 8 E chi flop)
    print(json.dumps(out))


def chi64phys():
    """PHYSICAL evolution up to the chi = 64 cap: 4x4 TFIM (J = 1, hx = 2.5, dt = 0.25: large steps, so that the bonds saturate within a few layers) from the product state, maxdim 64, cutoff 1e-13 -- the bond
    dimensions grow 2 -> 64 over the layers, theta is rank deficient on the way (the normal case early in an evolution), and the saturated layers
    run the chi = 64 kernels: register-direct products / epilogue, 64 x 64 and 128 x 128 MFMA Grams, Cholesky at n = 128, the Cholesky-QR theta SVD"""
    g = tn.named_grid((4, 4))
    groups = tn.edge_color(g, 4)
    layer = [("Rx", [v], 2 * 2.5 * 0.25) for v in g.vertices] + [("Rzz", [a, b], 2 * 1.0 * 0.25) for grp in groups for (a, b) in grp]
    bpc = tn.BeliefPropagationCache(tn.tensornetworkstate(np.complex64, lambda v: "↑", g))
    kw = dict(maxdim=64, cutoff=1e-13, normalize_tensors=True)
    errs_all, z_all, dims, tall = [], [], [], 0
    for it in range(9):
        info = {}
        bpc, errs = tn.apply_gates(layer, bpc, apply_kwargs=kw, info=info)
        errs_all += errs.tolist(); z_all += [float(np.real(x)) for x in tn.expect_all(bpc, "Z")]
        dims.append([bpc.bond_dim(a, b) for a, b in g.edges]); tall += info.get("n_tall_svd", 0)
    print(json.dumps(dict(errs=errs_all, z=z_all, dims=dims, tall=tall, norm=float(np.linalg.norm(bpc.tensor((2, 2)))))))


def chi32():
    """BASELINE configs[1] per-site shape on a 4x4 lattice (the four inner sites are bulk sites: degree 4, chi = 32, three gauged legs): one
    benchmark-style TFIM layer from a random chi = 32 state"""
    g = tn.named_grid((4, 4)); chi = 32
    bpc = tn.BeliefPropagationCache(tn.tensornetworkstate(np.complex64, lambda v: "↑", g))
    rng = np.random.default_rng(11)
    for v in g.vertices:
        shp = (2,) + (chi,) * g.degree(v); n = int(np.prod(shp))
        bpc._set_tensor(v, rng.standard_normal(2 * n, dtype=np.float32).view(np.complex64).reshape(shp) / np.float32(np.sqrt(n)))
    bpc = tn.update(bpc, maxiter=3, tolerance=None)
    layer = [("Rx", [v], 0.05) for v in g.vertices] + [("Rzz", [a, b], 0.3) for grp in tn.edge_color(g, 4) for (a, b) in grp]
    tn.profile_enable(bpc, True)
    info = {}
    b2, errs = tn.apply_gates(layer, bpc, apply_kwargs=dict(maxdim=chi, cutoff=1e-10, normalize_tensors=True), bp_update_kwargs=dict(maxiter=3, tolerance=None), info=info)
    prof = tn.profile_get(b2)
    sp = []
    for (a, b) in g.edges:
        m = b2.message((a, b)).astype(np.complex128); w = np.linalg.eigvalsh((m + m.conj().T) / 2); sp += (w / w.sum()).tolist()
    print(json.dumps(dict(errs=errs.tolist(), z=[float(np.real(x)) for x in tn.expect_all(b2, "Z")], dims=[b2.bond_dim(a, b) for a, b in g.edges], spectra=sp,
                          modeprod_launches=prof["gate_modeprod"]["launches"], gram_launches=prof["gate_gram"]["launches"], batches=info["n_batches"])))


def c128():
    """ComplexF64 (the reference's default element type): 4x4 lattice, random chi = 8 state, a layer of Rx + Rzz / SWAP and BP sweeps -- the f64
    matrix-core products, Grams and gate epilogue (kernels_f64.hip) against the generic vector kernels (TNQS_NO_MFMA=1)"""
    g = tn.named_grid((4, 4))
    psi = tn.random_tensornetworkstate(np.complex128, g, bond_dimension=8, seed=9)
    bpc = tn.update(tn.BeliefPropagationCache(psi), maxiter=20, tolerance=None)
    out = {}
    for name in ("Rzz", "SWAP"):
        layer = [("Rx", [v], 0.3) for v in g.vertices] + [((name, [a, b], 0.4) if name == "Rzz" else (name, [a, b])) for grp in tn.edge_color(g, 4) for (a, b) in grp]
        info = {}
        b2, errs = tn.apply_gates(layer, bpc, apply_kwargs=dict(maxdim=8, cutoff=1e-12, normalize_tensors=True), bp_update_kwargs=dict(maxiter=20, tolerance=None), info=info)
        msgs = [b2.message(e) for (a, b) in g.edges[:8] for e in ((a, b), (b, a))]
        out[name] = dict(errs=errs.tolist(), z=[float(np.real(x)) for x in tn.expect_all(b2, "Z")], dims=[b2.bond_dim(a, b) for a, b in g.edges],
                         msgs=[[m.real.tolist(), m.imag.tolist()] for m in msgs], lowrank=info["n_lowrank_svd"], fallbacks=info["n_lowrank_fallbacks"], n2=info["n_two_site"])
    print(json.dumps(out))


def tolsweeps():
    """BP updates that need SEVERAL sweeps to reach their tolerance inside apply_gates: the optimistic update (verdict read by the next batch, which starts over
    after the remaining sweeps when it is negative) against the blocking one"""
    g = tn.named_grid((4, 4))
    out = {}
    for dt, tol in ((np.complex64, 1e-7), (np.complex128, 1e-12)):
        psi = tn.random_tensornetworkstate(dt, g, bond_dimension=4, seed=21)
        bpc = tn.update(tn.BeliefPropagationCache(psi), maxiter=40, tolerance=tol)
        layer = [("Rx", [v], 0.7) for v in g.vertices] + [("Rzz", [a, b], 0.9) for grp in tn.edge_color(g, 4) for (a, b) in grp] + [("Ry", [v], 0.3) for v in g.vertices]
        info = {}
        b2, errs = tn.apply_gates(layer, bpc, apply_kwargs=dict(maxdim=4, cutoff=1e-10, normalize_tensors=True), bp_update_kwargs=dict(maxiter=40, tolerance=tol), info=info)
        out[np.dtype(dt).name] = dict(errs=errs.tolist(), z=[float(np.real(x)) for x in tn.expect_all(b2, "Z")], dims=[b2.bond_dim(a, b) for a, b in g.edges],
                                      sweeps=info["n_sweeps"], updates=info["n_updates"], not_converged=info["bp_not_converged"])
    print(json.dumps(out))


def hh16():
    """heavy-hex (2,2) at chi = 16 (BASELINE configs[2] per-site shape: degree-3 sites of 2 x 16^3, degree-2 sites of 2 x 16^2): three BP sweeps in the default
    order and one Rx + Rzz layer -- the one-kernel small-site message (scalar form, matrix-core form, with and without its fused epilogue) against the generic route"""
    g = tn.heavy_hexagonal_lattice(2, 2)
    chi = 16
    bpc = tn.BeliefPropagationCache(tn.tensornetworkstate(np.complex64, lambda v: "↑", g))
    rng = np.random.default_rng(11)
    for v in g.vertices:
        shp = (2,) + (chi,) * g.degree(v); n = int(np.prod(shp))
        bpc._set_tensor(v, rng.standard_normal(2 * n, dtype=np.float32).view(np.complex64).reshape(shp) / np.float32(np.sqrt(n)))
    tn.profile_enable(bpc, True)
    bpc = tn.update(bpc, maxiter=3, tolerance=None)
    prof = tn.profile_get(bpc)
    msgs = [bpc.message(e) for (a, b) in g.edges for e in ((a, b), (b, a))]
    layer = [("Rx", [v], 0.4) for v in g.vertices] + [("Rzz", [a, b], 0.3) for grp in tn.edge_color(g, 3) for (a, b) in grp]
    b2, errs = tn.apply_gates(layer, bpc, apply_kwargs=dict(maxdim=chi, cutoff=1e-10, normalize_tensors=True), bp_update_kwargs=dict(maxiter=3, tolerance=None))
    out = dict(msgs=[[m.real.tolist(), m.imag.tolist()] for m in msgs], errs=errs.tolist(), z=[float(np.real(x)) for x in tn.expect_all(b2, "Z")],
               dims=[b2.bond_dim(a, b) for a, b in g.edges], fused=prof["bp_fused"]["launches"], small=prof["small"]["launches"], modeprod=prof["bp_modeprod"]["launches"])
    print(json.dumps(out))


def specfail():
    """deferred verification that FAILS (engine.hpp Check).  A TFIM evolution from the product state: the bonds grow to the cap (8) within four layers and carry a
    decaying spectrum.  From then on every bond sits at its cap, so apply_gates enqueues the batches without their host round trip, assuming the new bond dimension is
    the cap again -- but with cutoff = 1e-4 the tail of the spectrum is cut and bonds come out smaller; and with a tight tolerance the BP updates need several
    sweeps.  Each failed check puts the snapshot back and reruns the step the careful way: results must be those of a run that never ran ahead (TNQS_NO_SPECULATION=1)."""
    out = {}
    for dt in (np.complex64, np.complex128):
        g = tn.named_grid((4, 4))
        bpc = tn.BeliefPropagationCache(tn.tensornetworkstate(dt, lambda v: "↑", g))
        layer = [("Rx", [v], 0.3) for v in g.vertices] + [("Rzz", [a, b], 0.4) for grp in tn.edge_color(g, 4) for (a, b) in grp]
        res = dict(errs=[], dims=[], z=[], spec=0, redone=0, sweeps=0)
        rounds = [(1e-12, dict(maxiter=25, tolerance=1e-6))] * 4 + [(1e-4, dict(maxiter=25, tolerance=1e-6))] * 2 + [(1e-12, dict(maxiter=25, tolerance=1e-9))] * 2 + [(1e-12, dict(maxiter=25, tolerance=1e-3))] * 4
        for cutoff, bpkw in rounds:
            info = {}
            bpc, errs = tn.apply_gates(layer, bpc, apply_kwargs=dict(maxdim=8, cutoff=cutoff, normalize_tensors=True), bp_update_kwargs=bpkw, info=info)
            res["errs"] += errs.tolist(); res["dims"].append([bpc.bond_dim(a, b) for a, b in g.edges])
            res["spec"] += info["n_spec_batches"]; res["redone"] += info["n_spec_redone"]; res["sweeps"] += info["n_sweeps"]
        res["z"] = [float(np.real(tn.expect(bpc, ("Z", [v])))) for v in g.vertices]
        out[np.dtype(dt).name] = res
    print(json.dumps(out))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "specfail":
        return specfail()
    if len(sys.argv) > 1 and sys.argv[1] == "hh16":
        return hh16()
    if len(sys.argv) > 1 and sys.argv[1] == "tolsweeps":
        return tolsweeps()
    if len(sys.argv) > 1 and sys.argv[1] == "c128":
        return c128()
    if len(sys.argv) > 1 and sys.argv[1] == "chi32":
        return chi32()
    if len(sys.argv) > 1 and sys.argv[1] == "cubic16":
        return cubic16()
    if len(sys.argv) > 1 and sys.argv[1] == "chi64phys":
        return chi64phys()
    g = tn.named_grid((4, 4))
    psi = tn.random_tensornetworkstate(np.complex64, g, bond_dimension=8, seed=3)
    bpc = tn.update(tn.BeliefPropagationCache(psi), maxiter=30, tolerance=None)
    out = {}
    for name in ("Rzz", "CNOT", "CPHASE", "SWAP", "Rxxyyzz"):
        layer = [((name, [a, b], 0.4) if name in ("Rzz", "CPHASE", "Rxxyyzz") else (name, [a, b])) for grp in tn.edge_color(g) for (a, b) in grp]
        info = {}
        b2, errs = tn.apply_gates(layer, bpc, apply_kwargs=dict(maxdim=8, cutoff=1e-10, normalize_tensors=True),
                                  bp_update_kwargs=dict(maxiter=20, tolerance=None), info=info)
        out[name] = dict(errs=errs.tolist(), z=[float(np.real(tn.expect(b2, ("Z", [v])))) for v in g.vertices],
                         dims=[b2.bond_dim(a, b) for a, b in g.edges], lowrank=info["n_lowrank_svd"], n2=info["n_two_site"])
    # no truncation requested: the reference keeps every singular value, the zero ones too -- the low-rank route must stand back
    info = {}
    b3, _ = tn.apply_gates([("Rzz", list(g.edges[5]), 0.4)], bpc, apply_kwargs=dict(maxdim=64, cutoff=None, normalize_tensors=True),
                           bp_update_kwargs=dict(maxiter=2, tolerance=None), info=info)
    out["uncapped"] = dict(dim=b3.bond_dim(*g.edges[5]), lowrank=info["n_lowrank_svd"])
    # degree-6 sites (3x3x3 periodic cubic, the per-site shape of BASELINE's cubic configuration at a small bond dimension): two layers
    gc = tn.named_grid((3, 3, 3), periodic=True)
    psic = tn.random_tensornetworkstate(np.complex64, gc, bond_dimension=3, seed=5)
    bc = tn.update(tn.BeliefPropagationCache(psic), maxiter=30, tolerance=None)
    layer = [("Rz", [v], -0.04) for v in gc.vertices] + [("Rxx", [a, b], 0.3) for grp in tn.edge_color(gc) for (a, b) in grp]
    errs_all = []
    for _ in range(2):
        bc, errs = tn.apply_gates(layer, bc, apply_kwargs=dict(maxdim=3, cutoff=1e-10, normalize_tensors=True), bp_update_kwargs=dict(maxiter=20, tolerance=None))
        errs_all += errs.tolist()
    out["cubic"] = dict(errs=errs_all, z=[float(np.real(tn.expect(bc, ("Z", [v])))) for v in gc.vertices], dims=[bc.bond_dim(a, b) for a, b in gc.edges])
    print(json.dumps(out))


if __name__ == "__main__":
    sys.exit(main())
