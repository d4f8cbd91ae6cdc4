"""CPU (gloo, world_size 2): the vertex-sharded protocol -- what is computed where and what is exchanged -- gives the
same truncation errors, bond dimensions and <Z> as the serial algorithm; plus the host-side partition helper."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

import tnqs_amd as tn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_vertices():
    for nv, w in [(400, 8), (25, 4), (7, 2), (3, 3), (5, 1)]:
        own = tn.partition_vertices(nv, w)
        assert len(own) == nv and sorted(set(own)) == list(range(w))
        counts = np.bincount(own, minlength=w)
        assert counts.max() - counts.min() <= 1
        assert own == sorted(own)
    with pytest.raises(ValueError):
        tn.partition_vertices(0, 2)
    assert tn.dist.exchange_bytes_needed(32, 2, 760, 400, 8) < 200 << 20


def test_partition_by_work_balances_the_bulk_sites():
    """strong scaling is decided by the heaviest rank: a boundary site of the open 20 x 20 lattice costs 1/32 (degree 3) or 1/1024 (corner) of a bulk
    site at chi = 32, so equal vertex counts leave the end ranks with 27 and the middle ranks with 45 bulk sites at 8 ranks.  The weighted
    partition is contiguous, uses every rank, and is optimal: its heaviest block is within one bulk site of total / ranks (brute force on a small case)."""
    g = tn.named_grid((20, 20))
    w = tn.dist.site_weights(g, 32)
    for world in (2, 4, 8):
        own = tn.partition_vertices(g.nv(), world, w)
        assert len(own) == g.nv() and own == sorted(own) and sorted(set(own)) == list(range(world))
        summ = tn.dist.partition_summary(g, own, 32)
        total = sum(w)
        assert max(summ["load_in_bulk_sites"]) <= total / world + 1.0, summ        # optimal to within one site
        # the end ranks own the top and bottom rows (boundary sites pay the latency floor of the cost model): fewer bulk sites there, never more
        assert summ["bulk_sites"][0] <= min(summ["bulk_sites"][1:-1] or summ["bulk_sites"]) and summ["bulk_sites"][-1] <= min(summ["bulk_sites"][1:-1] or summ["bulk_sites"])
        own0 = tn.partition_vertices(g.nv(), world, tn.dist.site_weights(g, 32, floor=0.0))      # elements only: bulk counts within one of each other
        s0 = tn.dist.partition_summary(g, own0, 32)
        assert max(s0["bulk_sites"]) - min(s0["bulk_sites"]) <= 1, s0
    assert tn.dist.partition_summary(g, tn.partition_vertices(g.nv(), 8), 32)["bulk_sites"] == [27, 45, 45, 45, 45, 45, 45, 27]
    # optimality against brute force: 9 weights, 3 blocks
    import itertools
    rng = np.random.default_rng(5)
    for _ in range(20):
        ww = rng.integers(1, 50, size=9).astype(float)
        own = tn.partition_vertices(9, 3, ww)
        mine = max(ww[np.array(own) == r].sum() for r in range(3))
        best = min(max(ww[:a].sum(), ww[a:b].sum(), ww[b:].sum()) for a, b in itertools.combinations(range(1, 9), 2))
        assert mine == best
    # the same against the O(world nv^2) linear-partition dynamic programme the bisection replaced (round-5 advisor finding), random sizes / zero weights / ties
    def dp_bottleneck(nv, world, w):
        pre = np.concatenate([[0.0], np.cumsum(w)]); k = min(world, nv)
        best = np.full((k + 1, nv + 1), np.inf); best[0][0] = 0.0
        for j in range(1, k + 1):
            for i in range(j, nv - (k - j) + 1):
                best[j][i] = min(max(best[j - 1][t], pre[i] - pre[t]) for t in range(j - 1, i))
        return best[k][nv]
    for trial in range(120):
        nv, world = int(rng.integers(1, 30)), int(rng.integers(1, 10))
        ww = (rng.random(nv) * (rng.random(nv) > 0.2)) if trial % 3 else rng.integers(0, 4, nv).astype(float)
        own = tn.partition_vertices(nv, world, list(ww)); k = min(world, nv)
        assert len(own) == nv and own == sorted(own) and sorted(set(own)) == list(range(k)), (nv, world, own)      # contiguous, every rank non-empty
        mine = max(ww[np.array(own) == r].sum() for r in range(k)); b = dp_bottleneck(nv, world, ww)
        assert abs(mine - b) <= 1e-12 * max(1.0, b), (nv, world, list(ww), own)
    # ... and it is fast where the programme was not: 100 x 100 sites on 8 ranks (minutes before)
    import time
    g100 = tn.named_grid((100, 100)); t0 = time.perf_counter()
    own = tn.partition_vertices(g100.nv(), 8, tn.dist.site_weights(g100, 32))
    assert time.perf_counter() - t0 < 5.0 and np.bincount(own).min() > 1200
    # more ranks than vertices: one vertex each, the surplus ranks own nothing
    assert tn.partition_vertices(3, 5, [1.0, 1.0, 1.0]) == [0, 1, 2]
    with pytest.raises(ValueError):
        tn.partition_vertices(4, 2, [1.0, -1.0, 1.0, 1.0])


def test_two_rank_protocol_matches_serial_oracle(tmp_path):
    out = str(tmp_path / "sim.pkl")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "tests", "sharding_sim_worker.py"), out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    z = pickle.load(open(out, "rb"))
    assert z["dims"] == z["odims"]
    assert np.max(np.abs(z["errs"] - z["oerrs"])) < 1e-12
    assert np.max(np.abs(z["ez"] - z["oez"])) < 1e-10
    assert sorted(set(z["owner"])) == [0, 1]


def _default_levels(g):
    """dependency levels of the library's default sweep order (host-only debug entry, as tests/test_bp_schedule.py reads it)"""
    import ctypes as C
    lib = C.CDLL(tn.LIB_PATH)
    fn = lib.tnqs_dbg_default_sequence_graph
    fn.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
    fn.restype = C.c_int
    idx = {v: i for i, v in enumerate(g.vertices)}
    es = np.array([idx[a] for (a, b) in g.edges], dtype=np.int32); ed = np.array([idx[b] for (a, b) in g.edges], dtype=np.int32)
    cap = 2 * g.ne(); src = (C.c_int * cap)(); dst = (C.c_int * cap)(); lev = (C.c_int * cap)(); n = C.c_int(0)
    assert fn(g.nv(), g.ne(), es.ctypes.data_as(C.POINTER(C.c_int32)), ed.ctypes.data_as(C.POINTER(C.c_int32)), src, dst, lev, cap, C.byref(n)) == 0
    out = [[] for _ in range(max(lev[i] for i in range(cap)) + 1)]
    for i in range(cap):
        out[lev[i]].append((g.vertices[src[i]], g.vertices[dst[i]]))
    return out


@pytest.mark.parametrize("name,make,chi,ncol", [("c2 20x20 chi 32", lambda: tn.named_grid((20, 20)), 32, 4),
                                                ("c4 10^3 periodic chi 16", lambda: tn.named_grid((10, 10, 10), periodic=True), 16, None),
                                                ("c5 32x32 chi 64", lambda: tn.named_grid((32, 32)), 64, 4)])
def test_exchange_buffer_holds_every_exchange_of_the_8_gpu_configurations(name, make, chi, ncol):
    """round-5 verdict item 7: the first 8-GPU run must not fail on buffer sizing.  The exchange points of one layer of BASELINE configs[1], [3], [4] at EIGHT
    ranks, restated from the engine's slot rules (dist.exchange_plan), against the buffer `shard()` allocates by default -- and the bytes per layer, which for
    20 x 20 is the figure the one-GPU proxy measured through the real library (profiles/r5_shard_proxy_x3.txt: 125.39 MB gathered per rank and layer, 18 exchanges)."""
    g = make()
    groups = tn.edge_color(g, ncol) if ncol else tn.edge_color(g)
    world = 8
    owner = tn.partition_vertices(g.nv(), world, tn.dist.site_weights(g, chi))
    plan = tn.dist.exchange_plan(g, owner, chi, groups, _default_levels(g))
    default_buf = world * tn.dist.exchange_bytes_needed(chi, 2, g.ne(), g.nv(), 8) // max(1, world // 2)
    print(name, {k: plan[k] for k in ("max_exchange_bytes", "bytes_gathered_per_layer", "exchanges_per_layer")}, "buffer", default_buf)
    assert plan["max_exchange_bytes"] <= default_buf, (name, plan, default_buf)
    assert plan["max_exchange_bytes"] <= 0.6 * default_buf                                  # (and with room to spare: bond dimensions below the cap only shrink the blocks)
    if name.startswith("c2"):
        # (the restatement is compared with the library's own counters, exactly, in tests/test_bench_launch.py::test_eight_ranks_over_gloo; the proxy balances its ranks
        #  with its own load model -- 63 / 45 / 45 / 46 / 47 / 46 / 45 / 63 vertices -- so its blocks differ from this owner map's by a few per cent: 119.4 against 125.4 MB)
        assert plan["exchanges_per_layer"] == 18 and abs(plan["bytes_gathered_per_layer"] / 1e6 - 125.39) < 0.07 * 125.39, plan


def test_exchange_plan_cuts_a_level_where_the_engine_does():
    """a BP level is cut into sub-batches by workspace bytes (engine_bp.cpp: 2 x the source site's tensor per message against TNQS_BP_WS_MB), each an exchange of its own:
    a tighter bound means more exchanges, never fewer bytes (the largest block of every piece is what travels), and no bound at all means one exchange per level."""
    g = tn.named_grid((6, 6))
    groups = tn.edge_color(g, 4)
    owner = tn.partition_vertices(g.nv(), 4, tn.dist.site_weights(g, 8))
    levels = _default_levels(g)
    whole = tn.dist.exchange_plan(g, owner, 8, groups, levels, bp_ws_bytes=1 << 60)
    assert whole["by_kind"]["bp_level"]["count"] == len(levels)
    per_msg = 2 * 2 * 8 ** 4 * 8                                   # a bulk site's message: 2 x (d chi^4) elements x 8 bytes
    tight = tn.dist.exchange_plan(g, owner, 8, groups, levels, bp_ws_bytes=3 * per_msg)
    assert tight["by_kind"]["bp_level"]["count"] > whole["by_kind"]["bp_level"]["count"]
    assert tight["by_kind"]["bp_level"]["sum"] >= whole["by_kind"]["bp_level"]["sum"]
    assert tight["by_kind"]["gate_record"] == whole["by_kind"]["gate_record"] and tight["by_kind"]["gate_gram"] == whole["by_kind"]["gate_gram"]
    one = tn.dist.exchange_plan(g, owner, 8, groups, levels, bp_ws_bytes=1)      # every message alone (the first of a piece is always taken)
    assert one["by_kind"]["bp_level"]["count"] == sum(len(lv) for lv in levels)
