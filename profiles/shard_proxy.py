#!/usr/bin/env python
"""Strong-scaling proxy on ONE GPU, on the SHARDED code path (round-4 verdict: the 7 x 7 open lattice timed before was neither the heaviest rank's load nor the
sharded path).  N ranks of the real vertex-sharded apply_gates of the benchmark layer (work-balanced partition, callback transport) live in ONE process -- one
handle, one host thread and one exchange buffer per rank -- and share the device by TAKING TURNS: a rank holds a token while it computes and gives it up inside
every exchange, where it waits for its peers at a barrier; the gathered block is then copied device-to-device from the peers' buffers (bulk-synchronous
emulation; one process, because the queues of eight processes oversubscribe the hardware scheduler: 100-170 ms per rank instead of ~20).  Each rank records the
wall time of every superstep (host preparation + kernels + the stream synchronisation in front of the exchange).  From those:
    t_rank[r]  = sum over supersteps of rank r's own time                      -- what rank r would spend computing on a GPU of its own
    t_bsp      = sum over supersteps of the slowest rank's time                -- a run in which every exchange is a barrier (what the callback transport does)
and the fit t = a * load + b over the ranks' (load in bulk sites, t_rank) pairs of all N.  Communication itself (xGMI) is NOT in these numbers; bytes per layer
and the time of the emulation's device-to-device gather copies are reported next to them.
    python profiles/shard_proxy.py --ranks 1,2,4,8 [--L 20 --chi 32 --steps 3 --warmup 2]
"""
import argparse, ctypes as C, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np


class LocalSharding:
    """callback transport between handles of one process: rank r's block of every exchange is read straight out of rank r's exchange buffer"""

    def __init__(self, rank, world, owner, cap_per_rank, shared):
        import torch
        from tnqs_amd import _lib as L
        self.rank, self.world, self.owner, self.shared, self.torch = rank, world, list(owner), shared, torch
        self.per_rank_cap = cap_per_rank
        self.buf = torch.empty(cap_per_rank * world, dtype=torch.uint8, device="cuda:0")
        self.n_exchanges = 0; self.bytes_exchanged = 0
        self.steps = []; self.copy_s = 0.0; self.t0 = None
        shared["bufs"][rank] = self.buf

        def _cb(ctx, base, bytes_per_rank, nranks):
            try:
                n = int(bytes_per_rank); sh = self.shared
                self.release()                                    # the library synchronised its stream before calling: this rank's superstep ends here
                sh["bar"].wait()                                  # every rank has packed its block
                self.acquire_untimed()
                t = time.perf_counter()
                for r in range(nranks):
                    if r != self.rank:
                        self.buf[r * n:(r + 1) * n].copy_(sh["bufs"][r][r * n:(r + 1) * n])
                self.torch.cuda.synchronize()
                self.copy_s += time.perf_counter() - t
                sh["tok"].release()
                sh["bar"].wait()                                  # nobody repacks its buffer before every peer has read it
                self.acquire()
                self.n_exchanges += 1; self.bytes_exchanged += n * nranks
                return 0
            except Exception as e:                                # never let an exception cross the C boundary
                print(f"[shard_proxy] exchange failed on rank {self.rank}: {e!r}", file=sys.stderr)
                try:
                    self.shared["bar"].abort()
                except Exception:
                    pass
                return 1
        self.cb = L.ALLGATHER_FN(_cb)

    def acquire(self):
        self.shared["tok"].acquire(); self.t0 = time.perf_counter()

    def acquire_untimed(self):
        self.shared["tok"].acquire()

    def release(self):
        self.steps.append(time.perf_counter() - self.t0); self.shared["tok"].release()


def run(world, args):
    import torch
    import tnqs_amd as tn
    from tnqs_amd import dist as tdist, _lib as L
    import bench
    Lx, chi = args.L, args.chi
    g = tn.named_grid((Lx, Lx)); groups = tn.edge_color(g, 4); layer = bench.tfim_layer(tn, g, groups)
    kw = dict(maxdim=chi, cutoff=1e-10, normalize_tensors=True)
    owner = tdist.partition_vertices(g.nv(), world, tdist.site_weights(g, chi)) if world > 1 else [0] * g.nv()
    summ = tdist.partition_summary(g, owner, chi)
    shared = {"tok": threading.Lock(), "bar": threading.Barrier(world), "bufs": [None] * world}
    cap = tdist.exchange_bytes_needed(chi, 2, g.ne(), g.nv(), 8)
    bpcs, shs = [], []
    for r in range(world):
        b = tn.BeliefPropagationCache(tn.tensornetworkstate(np.complex64, lambda v: "↑", g), device=0)
        sh = None
        if world > 1:
            sh = LocalSharding(r, world, owner, cap, shared)
            ow, owp = L.i32(owner)
            L.check(L.lib.tnqs_set_sharding(b._h, r, world, owp, sh.cb, None, C.c_void_p(sh.buf.data_ptr()), C.c_int64(sh.buf.numel())))
            b._shard = sh
        for v, t in bench.random_state_tensors(g, chi, 2, 1234, np.complex64, wanted=(None if world == 1 else b.owns)):
            if isinstance(t, tuple):
                b._declare_dims(v, t)
            else:
                b._set_tensor(v, t)
        bpcs.append(b); shs.append(sh)
    nl = args.warmup + args.steps
    per_layer = [[None] * nl for _ in range(world)]; copy_s = [0.0] * world; infos = [None] * world; errors = []
    outer = threading.Barrier(world)

    def body(r):
        try:
            b = bpcs[r]
            for it in range(nl):
                outer.wait()
                if world > 1:
                    sh = shs[r]; sh.steps = []; sh.copy_s = 0.0
                    sh.acquire()
                else:
                    t0 = time.perf_counter()
                info = {}
                b, _errs = tn.apply_gates(layer, b, apply_kwargs=kw, info=info)
                torch.cuda.synchronize()
                if world > 1:
                    sh.release(); per_layer[r][it] = list(sh.steps); copy_s[r] = sh.copy_s
                else:
                    per_layer[r][it] = [time.perf_counter() - t0]
                infos[r] = info
        except Exception as e:                                    # noqa: BLE001
            errors.append((r, repr(e)))
            try:
                outer.abort(); shared["bar"].abort()
            except Exception:
                pass

    ths = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    [t.start() for t in ths]; [t.join() for t in ths]
    if errors:
        raise SystemExit(f"shard_proxy: N = {world}: {errors}")
    arr = np.array([[per_layer[r][it] for it in range(args.warmup, nl)] for r in range(world)])      # [rank][layer][superstep]
    t_rank = arr.sum(axis=2).mean(axis=1) * 1e3
    t_bsp = arr.max(axis=0).sum(axis=1).mean() * 1e3
    nst = arr.shape[2]
    out = {"n_ranks": world, "L": Lx, "chi": chi, "supersteps_per_layer": int(nst), "exchanges_per_layer": int(nst - 1), "partition": summ,
           "ms_per_layer_by_rank": [round(float(x), 3) for x in t_rank], "ms_per_layer_heaviest_rank": round(float(t_rank.max()), 3),
           "ms_per_layer_bsp": round(float(t_bsp), 3), "bp_sweeps": infos[0].get("n_sweeps"),
           "svd_sweeps_slowest_gate": max(i.get("n_svd_sweeps_max", 0) for i in infos),
           "MB_gathered_per_layer_per_rank": (round(shs[0].bytes_exchanged / nl / 1e6, 2) if world > 1 else 0.0),
           "emulation_gather_copies_ms_per_layer": (round(1e3 * max(copy_s), 3) if world > 1 else 0.0)}
    del bpcs, shs
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--L", type=int, default=20); ap.add_argument("--chi", type=int, default=32)
    ap.add_argument("--steps", type=int, default=3); ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    res = []
    for n in [int(x) for x in args.ranks.split(",")]:
        res.append(run(n, args)); print("PROXY " + json.dumps(res[-1]), flush=True)
    xs, ys = [], []
    for r in res:
        xs += r["partition"]["load_in_bulk_sites"]; ys += r["ms_per_layer_by_rank"]
    a, b = (np.polyfit(np.array(xs), np.array(ys), 1) if len(set(xs)) > 1 else (float("nan"), float("nan")))
    t1 = [r for r in res if r["n_ranks"] == 1]
    print(json.dumps({"fit_ms": {"a_per_bulk_site": round(float(a), 4), "b": round(float(b), 3)},
                      "speedup_bsp_before_communication": {str(r["n_ranks"]): (round(t1[0]["ms_per_layer_heaviest_rank"] / r["ms_per_layer_bsp"], 2) if t1 else None) for r in res},
                      "speedup_heaviest_rank_before_communication": {str(r["n_ranks"]): (round(t1[0]["ms_per_layer_heaviest_rank"] / r["ms_per_layer_heaviest_rank"], 2) if t1 else None) for r in res},
                      "runs": res}))
