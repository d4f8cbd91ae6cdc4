"""GPU: alternative routes of the gate path give the same results as the default route.  Each variant runs in its own process
(tests/toggle_worker.py) because the library reads its switches once."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


_RUNS = {}


def run_worker(env, *args):
    """one worker process per (switches, mode); the result of a configuration is reused by the tests that compare against it (the default
    route is the reference of every parametrised case: without the memo it ran once per case)"""
    key = (tuple(sorted(env.items())), args)
    if key not in _RUNS:
        r = subprocess.run([sys.executable, os.path.join(HERE, "toggle_worker.py"), *args], env=dict(os.environ, PYTHONPATH=os.pathsep.join(sys.path), **env),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        _RUNS[key] = r.stdout.strip().splitlines()[-1]
    return json.loads(_RUNS[key])


def test_lowrank_theta_route_matches_the_full_svd():
    """theta of a gate with operator Schmidt rank kappa has rank <= kappa chi; when that is below its column count the SVD runs on the
    (r d) x (kappa chi) factor M = A conj(L) instead (GateItem in kernels.hpp).  Same bond dimensions, truncation
    errors and <Z> to f32 rounding (measured 3e-7 / 5e-7 on errors of 0.04 ... 0.4); taken for kappa = 2 gates in the bulk, never for
    kappa = 4 gates, and never when the requested bond cap exceeds kappa chi (the reference would keep the zero singular values too)."""
    on, off = run_worker({}), run_worker({"TNQS_NO_LOWRANK": "1"})
    for name in ("Rzz", "CNOT", "CPHASE", "SWAP", "Rxxyyzz"):
        a, b = on[name], off[name]
        assert b["lowrank"] == 0
        assert a["dims"] == b["dims"]
        assert np.max(np.abs(np.array(a["errs"]) - np.array(b["errs"]))) < 2e-6
        assert np.max(np.abs(np.array(a["z"]) - np.array(b["z"]))) < 5e-6
    for name in ("Rzz", "CNOT", "CPHASE"):
        assert 0 < on[name]["lowrank"] <= on[name]["n2"]
    for name in ("SWAP", "Rxxyyzz"):
        assert on[name]["lowrank"] == 0
    # uncapped: both routes keep all singular values of theta (31 here), more than kappa chi = 16 -- only the full SVD can deliver them
    assert on["uncapped"]["lowrank"] == 0 and on["uncapped"]["dim"] == off["uncapped"]["dim"] and on["uncapped"]["dim"] > 16


@pytest.mark.parametrize("switch", ["TNQS_NO_CHOL", "TNQS_NO_SMALLSVD", "TNQS_JACOBI_GLOBAL", "TNQS_NO_PAIR", "TNQS_NO_MFMA", "TNQS_NO_DEFER_1SITE", "TNQS_NO_PRODCACHE", "TNQS_NO_BRA_PRODUCTS",
                                    "TNQS_NO_SPECULATION", "TNQS_NO_PRECOND_SVD", "TNQS_NO_SMALL_SITE_BP", "TNQS_NO_BF16X3"])
def test_alternative_routes_match_the_default(switch):
    """every switch the library still has (csrc/engine_internal.hpp: one alternative route per kernel family -- round 6 deleted the A/B levers of decisions long
    made) selects another route of the same algorithm: all-eigen factorisation instead of Cholesky, Gram-eigen instead of the small-SVD route, global-memory
    Jacobi, single-leg mode products, no matrix-core kernels at all, one-site gates applied at once, no remembered BP products, no run-ahead, plain Jacobi,
    no small-site kernel, f32 matrix instructions.  Same bond dimensions; truncation errors and <Z> to f32
    rounding of the whole layer (bounds 2e-3 relative / 1e-5; the switch must at least run -- an intermediate version of the per-site Cholesky
    fallback crashed under TNQS_NO_CHOL without any test noticing)."""
    ref, alt = run_worker({}), run_worker({switch: "1"})
    for name in ("Rzz", "CNOT", "CPHASE", "SWAP", "Rxxyyzz", "cubic"):       # "cubic": degree-6 sites, two layers
        a, b = ref[name], alt[name]
        assert a["dims"] == b["dims"], name
        ea, eb = np.array(a["errs"]), np.array(b["errs"])
        # relative: an absolute 2e-5 here once hid a 6 % error in truncation errors of 1e-4 (the f32 underflow of the global-memory Jacobi
        # kernel, DESIGN.md 4.2, ran under TNQS_JACOBI_GLOBAL at every size and this test stayed green)
        assert np.all(np.abs(ea - eb) < 2e-3 * np.maximum(ea, eb) + 2e-7), (name, float(np.max(np.abs(ea - eb))))
        assert np.max(np.abs(np.array(a["z"]) - np.array(b["z"]))) < 1e-5, name


def test_staging_arena_overflow_keeps_descriptors_alive():
    """the pinned staging arena of descriptor uploads wraps around in the middle of a phase (TNQS_ARENA_KB=48: a few uploads fill it): only the
    HOST staging may be recycled at that point -- the device copies of descriptor arrays whose kernels are not launched yet must stay
    allocated (round-2 advisor finding: the overflow path released them, and the next allocation of the same size class could alias them).
    Bit-identical results to the default run, since nothing of the arithmetic changes."""
    ref, alt = run_worker({}), run_worker({"TNQS_ARENA_KB": "48"})
    for name in ("Rzz", "CNOT", "CPHASE", "SWAP", "Rxxyyzz", "cubic"):
        assert ref[name]["dims"] == alt[name]["dims"], name
        assert ref[name]["errs"] == alt[name]["errs"], name
        assert ref[name]["z"] == alt[name]["z"], name


@pytest.mark.parametrize("switch", ["TNQS_NO_DEFER_1SITE", "TNQS_NO_SPECULATION", "TNQS_NO_PRECOND_SVD", "TNQS_NO_BF16X3", "TNQS_NO_PAIR"])
def test_bulk_shape_routes_match(switch):
    """the chi = 32 bulk shape (BASELINE configs[1]) on the alternative routes that touch it: one-site gates applied at once, no run-ahead (every batch reads its
    results back, every update waits for its verdict), plain Jacobi for the theta SVD, f32 matrix instructions, single-leg products instead of the plane kernels.
    Same bond dimensions, truncation errors (relative), <Z> and message spectra to 1e-5."""
    ref, alt = run_worker({}, "chi32"), run_worker({switch: "1"}, "chi32")
    assert ref["dims"] == alt["dims"]
    ea, eb = np.array(ref["errs"]), np.array(alt["errs"])
    assert np.all(np.abs(ea - eb) < 2e-3 * np.maximum(ea, eb) + 2e-7), float(np.max(np.abs(ea - eb)))
    dz = float(np.max(np.abs(np.array(ref["z"]) - np.array(alt["z"])))); dsp = float(np.max(np.abs(np.array(ref["spectra"]) - np.array(alt["spectra"]))))
    print(switch, "max |dZ|", dz, " spectra", dsp, " max |derr|", float(np.max(np.abs(ea - eb))))
    assert dz < 1e-5 and dsp < 1e-5
    if switch == "TNQS_NO_PAIR":            # the plane route must actually have been taken by the default: it saves single-leg launches
        assert ref["modeprod_launches"] < alt["modeprod_launches"]


def test_torch_can_be_imported_after_the_library():
    """PyTorch-ROCm bundles its own HIP / HSA runtimes; the loader (_lib.py) shares them so that the import order does not matter.
    Without that, `import torch` after the library finds no GPU (checked with TNQS_NO_TORCH_RUNTIME=1)."""
    code = (
        "import sys; sys.path[:0] = %r\n"
        "import numpy as np, tnqs_amd as tn\n"
        "g = tn.named_grid((3, 3))\n"
        "bpc = tn.update(tn.BeliefPropagationCache(tn.random_tensornetworkstate(np.complex64, g, bond_dimension=2, seed=1)))\n"
        "z1 = tn.expect(bpc, ('Z', [g.vertices[0]]))\n"
        "import torch\n"
        "assert torch.cuda.is_available()\n"
        "assert torch.ones(4, device='cuda').sum().item() == 4.0\n"
        "assert abs(tn.expect(bpc, ('Z', [g.vertices[0]])) - z1) < 1e-12\n"
        "print('ok')\n") % (sys.path,)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_chi16_plane_kernels_match_the_single_leg_route():
    """BASELINE configs[3] per-site shape on the 3x3x3 torus: BP messages after two sweeps in the default order (elementwise: same site
    tensors, same order), then one layer (bond dimensions, truncation errors, <Z>) -- 16 x 16 plane kernels against single-leg products +
    plain Grams.  The plane route must actually have been taken (kernel-class launch counts)."""
    on, off = run_worker({}, "cubic16"), run_worker({"TNQS_NO_PAIR": "1"}, "cubic16")
    assert on["pairgram"] > 0 and on["pair"] > 0 and off["pairgram"] == 0 and off["pair"] == 0
    worst = 0.0
    for ma, mb in zip(on["msgs"], off["msgs"]):
        a = np.array(ma[0]) + 1j * np.array(ma[1]); b = np.array(mb[0]) + 1j * np.array(mb[1])
        worst = max(worst, float(np.max(np.abs(a - b)) / np.max(np.abs(b))))
    print("chi = 16 plane kernels vs single-leg route: messages", worst)
    assert worst < 2e-5
    assert on["dims"] == off["dims"]
    ea, eb = np.array(on["errs"]), np.array(off["errs"])
    assert np.all(np.abs(ea - eb) < 2e-3 * np.maximum(ea, eb) + 2e-7)
    assert np.max(np.abs(np.array(on["z"]) - np.array(off["z"]))) < 1e-5


def test_small_site_message_kernel_matches_the_generic_route():
    """heavy-hex at chi = 16: the whole message of a small site in one LDS-resident kernel (kernels.hip bp_small_site_kernel) -- on the f32 matrix cores when every
    leg is 16-dimensional, with the normalisation and message_diff of msg_finalize_kernel inside -- against the generic chain + Gram route.  Messages after three
    sweeps elementwise (same site tensors, same order), then one layer."""
    switch = "TNQS_NO_SMALL_SITE_BP"
    on, off = run_worker({}, "hh16"), run_worker({switch: "1"}, "hh16")
    assert on["modeprod"] < off["modeprod"], (on["modeprod"], off["modeprod"])
    assert on["small"] < off["small"], (on["small"], off["small"])          # (the epilogue launches are gone too)
    worst = 0.0
    for ma, mb in zip(on["msgs"], off["msgs"]):
        a = np.array(ma[0]) + 1j * np.array(ma[1]); b = np.array(mb[0]) + 1j * np.array(mb[1])
        worst = max(worst, float(np.max(np.abs(a - b)) / np.max(np.abs(b))))
    print(switch, "small-site messages", worst)
    assert worst < 2e-5
    assert on["dims"] == off["dims"]
    ea, eb = np.array(on["errs"]), np.array(off["errs"])
    assert np.all(np.abs(ea - eb) < 2e-3 * np.maximum(ea, eb) + 2e-7)
    assert np.max(np.abs(np.array(on["z"]) - np.array(off["z"]))) < 1e-5


def test_partial_products_kept_across_levels_change_nothing_but_the_pass_count():
    """3x3x3 torus, chi = 16: the BP partial products remembered from one level to the next (engine_bp.cpp ProdCache: the levels of one axis
    share the product over the other axes' legs, two axes share the factor over the third) against every level absorbing from the site
    tensor again (TNQS_NO_PRODCACHE=1).  The same products of the same buffers in a different order: messages to f32 rounding, same layer;
    and the remembered route must actually save two-leg passes."""
    on, off = run_worker({}, "cubic16"), run_worker({"TNQS_NO_PRODCACHE": "1"}, "cubic16")
    assert 0 < on["pair_passes"] < off["pair_passes"] - 0.5, (on["pair_passes"], off["pair_passes"])       # (two-leg passes per site: the launches of a level are batched either way)
    # a byte bound that holds three products of the 27 x 3 the sweep would keep: entries are evicted all the time (least recently used of all
    # sites), reuse mostly misses -- and nothing but the pass count may change
    tight = run_worker({"TNQS_BP_CACHE_MB": "800"}, "cubic16")
    assert on["pair_passes"] < tight["pair_passes"] <= off["pair_passes"] + 1e-9, (on["pair_passes"], tight["pair_passes"], off["pair_passes"])
    assert tight["dims"] == off["dims"] and np.max(np.abs(np.array(tight["z"]) - np.array(off["z"]))) < 1e-5
    worst = 0.0
    for ma, mb in zip(on["msgs"], off["msgs"]):
        a = np.array(ma[0]) + 1j * np.array(ma[1]); b = np.array(mb[0]) + 1j * np.array(mb[1])
        worst = max(worst, float(np.max(np.abs(a - b)) / np.max(np.abs(b))))
    print("remembered partial products: two-leg passes per site", on["pair_passes"], "against", off["pair_passes"], " messages", worst)
    assert worst < 2e-5
    assert on["dims"] == off["dims"]
    ea, eb = np.array(on["errs"]), np.array(off["errs"])
    assert np.all(np.abs(ea - eb) < 2e-3 * np.maximum(ea, eb) + 2e-7)
    assert np.max(np.abs(np.array(on["z"]) - np.array(off["z"]))) < 1e-5


def test_bra_side_products_change_nothing_but_the_pass_count():
    """3x3x3 torus, chi = 16: a degree-6 site absorbs half of the messages of a level on the bra side (engine_bp.cpp: for a Hermitian message that is the
    conjugate of the same two-leg product, so the product over an axis is built once per sweep and serves as ket and as bra factor) against every message on the
    ket side (TNQS_NO_BRA_PRODUCTS=1).  Messages are Hermitian to f32 rounding, so the two routes agree to that; the bra route must save two-leg passes."""
    on, off = run_worker({}, "cubic16"), run_worker({"TNQS_NO_BRA_PRODUCTS": "1"}, "cubic16")
    assert 0 < on["pair_passes"] < off["pair_passes"] - 0.5, (on["pair_passes"], off["pair_passes"])
    assert on["pairgram"] == off["pairgram"]
    worst = 0.0
    for ma, mb in zip(on["msgs"], off["msgs"]):
        a = np.array(ma[0]) + 1j * np.array(ma[1]); b = np.array(mb[0]) + 1j * np.array(mb[1])
        worst = max(worst, float(np.max(np.abs(a - b)) / np.max(np.abs(b))))
    print("bra-side products: two-leg passes per site over two sweeps", on["pair_passes"], "against", off["pair_passes"], " messages", worst)
    assert worst < 2e-5
    assert on["dims"] == off["dims"]
    ea, eb = np.array(on["errs"]), np.array(off["errs"])
    assert np.all(np.abs(ea - eb) < 2e-3 * np.maximum(ea, eb) + 2e-7)
    assert np.max(np.abs(np.array(on["z"]) - np.array(off["z"]))) < 1e-5


def test_f64_matrix_core_kernels_match_the_vector_kernels():
    """ComplexF64 state: mode products, Grams and the gate epilogue on the f64 matrix cores (kernels_f64.hip) against the generic vector kernels
    (TNQS_NO_MFMA=1).  The same f64 arithmetic in a different summation order: bond dimensions, truncation errors, <Z> and message spectra to 1e-10."""
    on, off = run_worker({}, "c128"), run_worker({"TNQS_NO_MFMA": "1"}, "c128")
    for name in ("Rzz", "SWAP"):
        a, b = on[name], off[name]
        assert a["dims"] == b["dims"], name
        ea, eb = np.array(a["errs"]), np.array(b["errs"])
        assert np.all(np.abs(ea - eb) < 1e-9 * np.maximum(ea, eb) + 1e-13), (name, float(np.max(np.abs(ea - eb))))
        assert np.max(np.abs(np.array(a["z"]) - np.array(b["z"]))) < 1e-10, name
        worst = 0.0                                        # message SPECTRA: the messages themselves carry the gauge of the theta SVD's singular vectors
        for ma, mb in zip(a["msgs"], b["msgs"]):
            x = np.array(ma[0]) + 1j * np.array(ma[1]); y = np.array(mb[0]) + 1j * np.array(mb[1])
            wx, wy = np.linalg.eigvalsh((x + x.conj().T) / 2), np.linalg.eigvalsh((y + y.conj().T) / 2)
            worst = max(worst, float(np.max(np.abs(wx / wx.sum() - wy / wy.sum()))))
        print(name, "f64 matrix cores against vector kernels: message spectra", worst, " max |dZ|", float(np.max(np.abs(np.array(a["z"]) - np.array(b["z"])))))
        assert worst < 1e-10, name


def test_chi64_kernels_match_the_generic_route_on_a_physical_evolution():
    """nine TFIM layers from the product state at maxdim 64 (bonds grow 2 -> 64, theta rank deficient on the way): the chi = 64 kernels
    (kernels_chi64.hip and the Cholesky-QR theta SVD) against the generic route (TNQS_NO_CHI64=1: round-1 kernels, global-memory Jacobi).
    Same bond dimensions layer by layer; truncation errors and <Z> to f32 rounding of the whole evolution."""
    on, off = run_worker({}, "chi64phys"), run_worker({"TNQS_NO_CHI64": "1"}, "chi64phys")
    print("bond dimensions per layer:", [max(d) for d in on["dims"]], " tall SVDs:", on["tall"])
    # the cutoff of this evolution (1e-13 on S^2, chosen so that the bonds saturate) sits at f32 rounding: a singular value on the threshold may fall
    # either side of it when the rounding of the contractions changes (three- vs four-multiplication products) -- bond dimensions may differ
    # by one on a few bonds, nothing more
    da, db = np.array(on["dims"]), np.array(off["dims"])
    assert da.shape == db.shape and np.max(np.abs(da - db)) <= 1 and np.count_nonzero(da != db) <= max(1, da.size // 50), (da - db).tolist()
    assert max(on["dims"][-1]) == 64
    assert on["tall"] > 0 and off["tall"] == 0
    ea, eb = np.array(on["errs"]), np.array(off["errs"])
    print("chi = 64 physical evolution: max |derr|", float(np.max(np.abs(ea - eb))), " max err", float(ea.max()), " max |dZ|", float(np.max(np.abs(np.array(on["z"]) - np.array(off["z"])))))
    assert np.all(np.abs(ea - eb) < 5e-3 * np.maximum(ea, eb) + 5e-7)
    assert np.max(np.abs(np.array(on["z"]) - np.array(off["z"]))) < 1e-4
    assert abs(on["norm"] - 1) < 1e-4 and np.all(ea >= 0) and np.all(ea <= 1)


def test_partial_product_cache_under_a_small_budget():
    """TNQS_BP_CACHE_MB=700 holds two or three of the 268 MB products of the 3x3x3 torus: entries are evicted all the time (in bulk, least recently
    used first).  Validity is by buffer identity, so an evicted product is simply recomputed: same layer as with the default budget, more two-leg passes."""
    on, small = run_worker({}, "cubic16"), run_worker({"TNQS_BP_CACHE_MB": "700"}, "cubic16")
    assert small["pair_passes"] > on["pair_passes"] + 0.2, (small["pair_passes"], on["pair_passes"])       # (two-leg passes per site: the launches of a level are batched either way)
    assert on["dims"] == small["dims"]
    ea, eb = np.array(on["errs"]), np.array(small["errs"])
    assert np.all(np.abs(ea - eb) < 2e-3 * np.maximum(ea, eb) + 2e-7)
    assert np.max(np.abs(np.array(on["z"]) - np.array(small["z"]))) < 1e-5


def test_c128_lowrank_theta_route_matches_the_full_svd():
    """ComplexF64 (round 4): theta SVD on the low-rank factor with B orthogonalised by CholeskyQR2 against the SVD of the full theta
    (TNQS_NO_LOWRANK=1): same bond dimensions, truncation errors and <Z> to 1e-9; taken for the kappa = 2 gate (Rzz) in the bulk, never for SWAP
    (kappa = 4: K = 4 chi is not below theta's column count), and no gate fell back on this well-conditioned state."""
    on, off = run_worker({}, "c128"), run_worker({"TNQS_NO_LOWRANK": "1"}, "c128")
    assert on["Rzz"]["lowrank"] > 0 and on["Rzz"]["fallbacks"] == 0 and on["SWAP"]["lowrank"] == 0 and off["Rzz"]["lowrank"] == 0
    for name in ("Rzz", "SWAP"):
        a, b = on[name], off[name]
        assert a["dims"] == b["dims"]
        assert np.max(np.abs(np.array(a["errs"]) - np.array(b["errs"]))) < 1e-9
        assert np.max(np.abs(np.array(a["z"]) - np.array(b["z"]))) < 1e-9       # (messages are not compared elementwise: the two routes fix the phases of the singular vectors differently)


def test_optimistic_bp_update_starts_over_when_its_sweep_did_not_converge():
    """Inside apply_gates a BP update returns after enqueuing its first sweep; the batch that follows prepares itself meanwhile and reads the verdict before its
    first launch.  Here every update needs several sweeps (tight tolerance, strong gates): the verdict is negative each time, the update is
    continued and the batch starts over -- same sweep counts and bit-identical results as with blocking updates (TNQS_NO_SPECULATION=1)."""
    on, off = run_worker({}, "tolsweeps"), run_worker({"TNQS_NO_SPECULATION": "1"}, "tolsweeps")
    for k in ("complex64", "complex128"):
        a, b = on[k], off[k]
        assert a["sweeps"] == b["sweeps"] and a["updates"] == b["updates"] == 5 and a["sweeps"] > 2 * a["updates"], (a["sweeps"], b["sweeps"])
        assert a["not_converged"] == b["not_converged"]
        assert a["dims"] == b["dims"] and a["errs"] == b["errs"] and a["z"] == b["z"]


def test_failed_deferred_verification_reruns_the_step_with_identical_results():
    """round 6: apply_gates runs ahead of the device (engine.hpp Check) -- gate batches whose bonds sit at their cap are enqueued without their host round trip,
    BP updates leave their verdict pending.  Here the assumptions FAIL on purpose (a cutoff that bites at saturated bonds; updates that need several sweeps): every
    failed check puts the snapshot of the state back, drains the stream and runs the step again the careful way.  Bit-identical bond dimensions, truncation
    errors, sweep counts and <Z> to a run that never ran ahead; and the run-ahead route must actually have been taken and have failed."""
    on, off = run_worker({}, "specfail"), run_worker({"TNQS_NO_SPECULATION": "1"}, "specfail")
    for k in ("complex64", "complex128"):
        a, b = on[k], off[k]
        assert b["spec"] == 0 and b["redone"] == 0
        assert a["redone"] > 0 and (a["spec"] > 0 or k == "complex128"), (k, a["spec"], a["redone"])      # (ComplexF64 batches never run ahead: the second factorisation pass needs its read-back; their BP updates do)
        assert a["dims"] == b["dims"] and a["sweeps"] == b["sweeps"], k
        assert a["errs"] == b["errs"] and a["z"] == b["z"], k
        assert max(a["dims"][3]) == 8 and min(min(d) for d in a["dims"][4:6]) < 8      # saturated after four layers; then the cutoff did bite
        print(k, "batches enqueued on assumptions:", a["spec"], " steps redone:", a["redone"], " bond dimensions after the cutoff layers:", sorted(set(a["dims"][5])))


def test_a_failing_batch_leaves_the_state_as_it_was():
    """the C ABI called IN PLACE on a handle: a message with a negative eigenvalue next to one gate of a batch makes the batch fail (DomainError in the
    reference, src/utils.jl:21) -- every gate's status is checked before anything of the handle is replaced: every site tensor, message and bond dimension
    is what it was before the call."""
    code = (
        "import sys, ctypes as C; sys.path[:0] = %r\n"
        "import numpy as np, tnqs_amd as tn\n"
        "from tnqs_amd import core, _lib as L\n"
        "g = tn.named_grid((4, 4))\n"
        "bpc = tn.update(tn.BeliefPropagationCache(tn.random_tensornetworkstate(np.complex64, g, bond_dimension=4, seed=3)), maxiter=10, tolerance=None)\n"
        "grp = tn.edge_color(g, 4)[0]\n"
        "layer = [('Rzz', [a, b], 0.3) for (a, b) in grp]\n"
        "a, b = grp[-1]\n"
        "w = [x for x in g.neighbors(a) if x != b][0]\n"
        "bad = np.diag([1.0, 0.5, 0.2, -0.3]).astype(np.complex64)\n"
        "bpc.setmessage((w, a), bad)\n"
        "before = {v: bpc.tensor(v) for v in g.vertices}; mb = {e: bpc.message(e) for e in g.edges}\n"
        "ng, nv_a, nv_p, vs_a, vs_p, mat_a = core._marshal_circuit(layer, g)\n"
        "ao = core._apply_opts(dict(maxdim=4, cutoff=1e-10, normalize_tensors=True), False)\n"
        "bo, keep = core._bp_opts(g, dict(maxiter=1, tolerance=None))\n"
        "errs = np.zeros(ng); st = L.ApplyStats()\n"
        "rc = L.lib.tnqs_apply_gates(bpc._h, ng, nv_p, vs_p, mat_a.ctypes.data_as(C.POINTER(C.c_double)), C.byref(ao), C.byref(bo), errs.ctypes.data_as(C.POINTER(C.c_double)), C.byref(st))\n"
        "assert rc != 0, rc\n"
        "for v in g.vertices: assert np.array_equal(bpc.tensor(v), before[v]), v\n"
        "for e in g.edges: assert np.array_equal(bpc.message(e), mb[e]) and bpc.bond_dim(*e) == 4, e\n"
        "print('ok')\n") % (sys.path,)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-1500:] + r.stderr[-3000:]
