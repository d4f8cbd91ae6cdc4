"""GPU: the vertex-sharded (2-rank) path of the library against the unsharded one.  Two processes share GPU 0 and
exchange through torch.distributed's gloo backend (host-staged); in production the same callback runs on RCCL."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("dt,nranks", [("c128", 2), ("c64", 2), ("chi32", 2), ("illc128", 2), ("c128", 4), ("illc128", 4),
                                       ("z6chi16", 2), ("z4chi64", 2), ("pendshard", 2)])
def test_sharded_apply_gates_matches_single_rank(tmp_path, dt, nranks):
    """2 and 4 ranks (with 4, ranks own three or four vertices each and sit out whole colour batches -- every collective must still be
    issued by all of them).  A one-off run with 8 ranks of all four cases gave the same deviations."""
    out = str(tmp_path / "res.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr", "127.0.0.1",
           "--master-port", str(29533 + nranks), os.path.join(ROOT, "tests", "sharded_worker.py"), out, dt]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    print(r.stdout[-1500:])
    z = np.load(out)
    # "illc128": ComplexF64, cutoff = 1e-14 -- agreement to 1e-11 needs the second factorisation pass on both sides (1e-9 with a single pass)
    tol = 1e-9 if dt == "c128" else (1e-11 if dt == "illc128" else 1e-5)      # c64, chi32, z6chi16, z4chi64: 1e-5 (the north star's bound)
    assert np.array_equal(z["dims_sh"], z["dims_un"])
    assert np.max(np.abs(z["errs_sh"] - z["errs_un"])) < (1e-10 if dt in ("c128", "illc128") else 1e-5)
    print(dt, "max |<Z>_sharded - <Z>_single| =", np.max(np.abs(z["ez_sh"] - z["ez_un"])), " spectra:", np.max(np.abs(z["sp_sh"] - z["sp_un"])))
    assert np.max(np.abs(z["ez_sh"] - z["ez_un"])) < tol
    assert np.max(np.abs(z["sp_sh"] - z["sp_un"])) < tol
    assert int(z["n_exchanges"]) > 0
    if len(z["zz_sh"]):       # multi-site expectation values (one exchange per region vertex) and the symmetric gauge on the sharded handle
        ok = ~np.isnan(z["zz_un"])
        assert np.array_equal(np.isnan(z["zz_sh"]), np.isnan(z["zz_un"])) and ok.any()
        assert np.max(np.abs(z["zz_sh"][ok] - z["zz_un"][ok])) < tol
        # the bond spectrum S of the gauge is compared up to its norm: messages are normalised by the SUM of their elements
        # (abstract...:182-187), which depends on the bond basis -- and the two runs fix the SVD gauge of a bond on different ranks
        nrm = lambda x: x / np.linalg.norm(x)
        assert np.max(np.abs(z["ezg_sh"] - z["ezg_un"])) < tol and np.max(np.abs(nrm(z["sg_sh"]) - nrm(z["sg_un"]))) < tol


def test_rccl_transport_loads_and_runs_on_one_gpu():
    """the library's own RCCL transport (tnqs_set_sharding_rccl: librccl.so loaded at run time, in-place ncclAllGather on the handle's
    stream): unique id, one-rank communicator, all-gather round trip, teardown.  RCCL refuses two ranks on one GPU, so this is what a
    single-GPU box can check of it; the two-rank test below needs two GPUs."""
    import tnqs_amd as tn
    tn.dist.rccl_selftest(0, 1 << 20)
    # a one-rank sharded handle goes through tnqs_set_sharding_rccl as well (communicator of size 1, nothing to exchange)
    g = tn.named_grid((2, 2))
    b = tn.BeliefPropagationCache(tn.random_tensornetworkstate(np.complex64, g, bond_dimension=2, seed=1))
    sh = tn.shard(b, 0, 1, transport="rccl", exch_bytes=1 << 20)
    b2 = tn.update(b, maxiter=3, tolerance=None)
    assert np.isfinite(tn.expect(b2, ("Z", [g.vertices[0]]))) and sh.n_exchanges == 0


def test_one_rank_runs_every_exchange_of_a_layer_through_rccl(monkeypatch):
    """round-5 verdict item 7: RCCL on the data path of a MULTI-LEVEL layer, as far as one GPU can take it.  The exchange points are the same code for both
    transports -- the engine packs every block into the handle's exchange buffer and calls exchange(), which is an ncclAllGather enqueued on the handle's stream
    or the host callback (csrc/sharding.cpp) -- so what the callback tests above cannot show is the stream-ordered transport itself under the real sequence of
    pack / gather / unpack launches.  TNQS_FORCE_EXCHANGE=1 makes a ONE-rank RCCL handle take the sharded path (State::sharded): every BP level and every gate
    batch of two TFIM layers goes through reduce -> exchange buffer -> ncclAllGather (communicator of size one) -> finalize from the gathered block, with no
    stream synchronisation in between.  Results against the plain handle; the all-gather count against the layer's structure."""
    import tnqs_amd as tn
    g = tn.named_grid((4, 4)); groups = tn.edge_color(g, 4)
    layer = [("Rx", [v], 0.3) for v in g.vertices]
    for grp in groups:
        layer += [("Rzz", [a, b], 0.4) for (a, b) in grp]
    psi = tn.random_tensornetworkstate(np.complex64, g, bond_dimension=4, seed=3)
    kw = dict(maxdim=8, cutoff=1e-10, normalize_tensors=True)
    bpkw = dict(maxiter=30, tolerance=1e-7)
    plain = tn.update(tn.BeliefPropagationCache(psi), **bpkw)
    monkeypatch.setenv("TNQS_FORCE_EXCHANGE", "1")
    forced = tn.BeliefPropagationCache(psi)
    sh = tn.shard(forced, 0, 1, transport="rccl", exch_bytes=8 << 20)
    monkeypatch.delenv("TNQS_FORCE_EXCHANGE")
    forced = tn.update(forced, **bpkw)
    n0 = sh.n_exchanges
    assert n0 > 0                                                    # the update alone went through the transport (one all-gather per level and sweep)
    for _ in range(2):
        i1, i2 = {}, {}
        plain, e1 = tn.apply_gates(layer, plain, apply_kwargs=kw, bp_update_kwargs=bpkw, info=i1)
        forced, e2 = tn.apply_gates(layer, forced, apply_kwargs=kw, bp_update_kwargs=bpkw, info=i2)
        assert i1["n_sweeps"] == i2["n_sweeps"] and i2.get("n_spec_batches", 0) == 0
        assert np.max(np.abs(e1 - e2)) < 1e-5 * max(1e-12, float(np.max(e1)))
        assert [plain.bond_dim(a, b) for (a, b) in g.edges] == [forced.bond_dim(a, b) for (a, b) in g.edges]
    assert np.max(np.abs(tn.expect_all(plain, "Z") - tn.expect_all(forced, "Z"))) < 1e-5
    # per layer: two levels per sweep, and two exchanges per colour batch only when a gate straddles two ranks -- with one rank none does: ONE (the records)
    assert sh.n_exchanges - n0 >= 2 * (2 * 5 + 4)


@pytest.mark.parametrize("dt", ["c64", "chi32", "illc128"])
def test_sharded_over_rccl_matches_single_rank(tmp_path, dt):
    """two ranks on two GPUs, torch.distributed backend nccl for the rendezvous, the data path on the library's RCCL transport"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: RCCL does not allow two ranks on one device (the gloo-backed tests above cover the protocol, "
                    "test_rccl_transport_loads_and_runs_on_one_gpu the transport)")
    out = str(tmp_path / "res.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", TNQS_WORKER_BACKEND="nccl")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", os.path.join(ROOT, "tests", "sharded_worker.py"), out, dt]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    z = np.load(out)
    assert str(z["transport"]) == "RcclSharding" and int(z["n_exchanges"]) > 0
    tol = 1e-11 if dt == "illc128" else 1e-5
    assert np.array_equal(z["dims_sh"], z["dims_un"])
    assert np.max(np.abs(z["ez_sh"] - z["ez_un"])) < tol and np.max(np.abs(z["sp_sh"] - z["sp_un"])) < tol
