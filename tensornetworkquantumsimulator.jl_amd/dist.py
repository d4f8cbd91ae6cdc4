"""Multi-GPU sharding of the hot path (no reference analogue): one process per GPU, vertices are
block-partitioned over the ranks, each rank holds and updates only its own site tensors, messages are replicated.
The C library packs its exchange payloads into a device buffer owned here (a torch tensor) and calls back for an
all-gather, which runs through torch.distributed -- backend "nccl" (= RCCL over xGMI on ROCm) in production; with
the "gloo" backend the buffer is staged through the host, which lets two ranks share one GPU in tests."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _lib as L


def partition_vertices(nv: int, world: int, weights: Optional[Sequence[float]] = None) -> List[int]:
    """contiguous blocks of vertex positions: owner[v] = rank.  Any partition is valid (messages are replicated, every per-vertex
    unit of work belongs to exactly one rank); what a partition decides is the time of the slowest rank.
    weights = None: blocks of equal vertex COUNT.
    weights given (one per vertex; `site_weights` below: the elements of the site tensor at the evolution's bond dimension -- what every tensor
    pass of a BP level or gate batch costs -- with a floor for the latency-bound share every site pays): the contiguous partition that minimises the heaviest block -- on an open 20 x 20
    lattice at chi = 32 a boundary site costs 1/32 (degree 3) or 1/1024 (corner) of a bulk site, and equal counts would give the end
    ranks 27 and the middle ranks 45 bulk sites at 8 ranks instead of 40 / 41 each.  Ties: the lexicographically smallest cut positions
    among the optimal ones, so every rank computes the same partition from the same inputs."""
    if world < 1 or nv < 1:
        raise ValueError("partition_vertices: nv and world must be positive")
    if weights is None:
        base, rem = divmod(nv, world)
        owner = []
        for r in range(world):
            owner += [r] * (base + (1 if r < rem else 0))
        return owner
    w = [float(x) for x in weights]
    if len(w) != nv or any(not (x >= 0.0) for x in w):
        raise ValueError("partition_vertices: one non-negative weight per vertex")
    k = min(world, nv)
    pre = [0.0]
    for x in w:
        pre.append(pre[-1] + x)
    total = pre[-1]
    import bisect

    def blocks_needed(B):
        """need[i] = fewest blocks of weight <= B that cover the vertices i.. (greedy: every block as long as it can be), nv + 1 when a vertex alone exceeds B"""
        need = [0] * (nv + 1)
        for i in range(nv - 1, -1, -1):
            j = bisect.bisect_right(pre, pre[i] + B, i + 1, nv + 1) - 1       # furthest j with pre[j] - pre[i] <= B
            need[i] = (1 + need[j]) if j > i else nv + 1
        return need
    # the optimal bottleneck (round-5 advisor finding: the linear-partition dynamic programme this replaces was O(world nv^2) in pure Python -- minutes on a
    # 100 x 100 lattice, on every rank, before the first kernel): bisection on B with the greedy feasibility test, O(nv log nv) per probe; the bottleneck of
    # the greedy partition at the converged bound is an attained value, i.e. the optimum itself
    lo, hi = max(w), max(total, max(w))
    for _ in range(64):
        mid = 0.5 * (lo + hi)
        if blocks_needed(mid)[0] <= k:
            hi = mid
        else:
            lo = mid
    need = blocks_needed(hi)
    B, i = 0.0, 0
    while i < nv:                                                               # the greedy partition at `hi`: its heaviest block is the bound every cut below respects
        j = bisect.bisect_right(pre, pre[i] + hi, i + 1, nv + 1) - 1
        B = max(B, pre[j] - pre[i]); i = j
    need = blocks_needed(B)
    # among the partitions that attain it, the most even one: cut r goes as close to the r/k quantile of the total weight as the bound allows (the vertices before
    # it must fit r blocks -- guaranteed by how the previous cuts were placed --, its own block must stay <= B, and the rest must still fit k - r non-empty blocks).
    # A pure function of (nv, world, weights): every rank computes the same partition
    bounds = [0]
    for r in range(1, k):
        a = bounds[-1]
        hi_t = bisect.bisect_right(pre, pre[a] + B, a + 1, nv + 1) - 1          # the block [a, t) stays within the bound up to here
        hi_t = min(hi_t, nv - (k - r))                                          # ... and leaves a vertex for every later block
        lo_t = a + 1
        while lo_t < hi_t and need[lo_t] > k - r:                               # the rest must fit k - r blocks (need is non-increasing in t)
            lo_t += 1
        ideal = bisect.bisect_left(pre, total * r / k, 0, nv + 1)
        if ideal > 0 and abs(pre[ideal - 1] - total * r / k) <= abs(pre[min(ideal, nv)] - total * r / k):
            ideal -= 1
        bounds.append(max(lo_t, min(hi_t, ideal)))
    bounds.append(nv)
    owner = []
    for r in range(k):
        owner += [r] * (bounds[r + 1] - bounds[r])
    return owner


def site_weights(graph, chi: int, d: int = 2, floor: float = 0.17) -> List[float]:
    """cost model of a vertex, in units of the elements of the largest site tensor (d * chi^max degree): its tensor passes cost its own elements,
    d * chi^degree; on top of that every site pays the latency-bound part of the path (its share of the per-gate factorisation chain, the small-tensor
    BP kernels, descriptor traffic), which does not shrink with the tensor -- measured on the sharded 20 x 20 chi = 32 layer (profiles/shard_proxy.py,
    round 5): a rank with 40 bulk + 25 boundary sites took 26.3 ms against 22.5 ms for 40 bulk + 4 boundary sites, i.e. ~0.18 ms per boundary site next
    to 0.39 ms per bulk site, although a degree-3 site holds 1/32 of a bulk site's elements; with the floor at 0.3 the end ranks (36 bulk + 24 boundary sites) then
    took 22.9 ms against 23.7-25.6 ms for the interior ranks (42 + 4), which puts a boundary site at 0.06 ms = 0.17 of a bulk site.  `floor` is that latency share
    (0 = elements only)."""
    zmax = max(graph.degree(v) for v in graph.vertices)
    bulk = float(d) * float(chi) ** zmax
    return [max(float(d) * float(chi) ** graph.degree(v) / bulk, float(floor)) for v in graph.vertices]


def partition_summary(graph, owner: Sequence[int], chi: int, d: int = 2) -> dict:
    """per-rank load of a partition: vertex counts, counts of the highest-degree ("bulk") vertices and weights relative to one bulk site"""
    world = max(owner) + 1
    zmax = max(graph.degree(v) for v in graph.vertices)
    wts = site_weights(graph, chi, d)
    cnt, bulk, load = [0] * world, [0] * world, [0.0] * world
    for i, v in enumerate(graph.vertices):
        r = owner[i]
        cnt[r] += 1; bulk[r] += 1 if graph.degree(v) == zmax else 0; load[r] += wts[i]
    return {"vertices": cnt, "bulk_sites": bulk, "load_in_bulk_sites": [round(x, 2) for x in load]}


def exchange_bytes_needed(max_chi: int, d: int, n_edges: int, n_vertices: int, esz: int) -> int:
    """upper bound of one exchange block set: Gram matrices of every site of a colour batch, or all raw messages of
    one BP level, or the per-gate result records (S and X2)."""
    n = d * max_chi
    grams = n_vertices * (n * n * 16 + 256)
    msgs = 2 * n_edges * (max_chi * max_chi * esz + 256)
    recs = n_edges * (32 + max_chi * d * 8 + n * d * max_chi * esz + 256)
    return int(max(grams, msgs, recs)) + (1 << 20)


def exchange_plan(graph, owner: Sequence[int], chi: int, colour_groups, bp_levels, d: int = 2, esz: int = 8, bp_updates_per_layer: Optional[int] = None,
                  sweeps_per_update: int = 1, bp_ws_bytes: int = 24576 << 20) -> dict:
    """what the library's exchange points of ONE colour-batched layer need, restated on the host from the engine's slot rules (csrc/engine_bp.cpp: one all-gather per
    BP level, every message a 256-byte-rounded slot in its OWNER's block; csrc/engine_gates.cpp: per colour batch the f64 Gram matrices of the sites whose gate
    STRADDLES two ranks, then one record (chi', status, truncation error, S -- and X2 for a straddling gate) per gate in the block of its first vertex's owner).
    An exchange moves `stride x nranks` bytes into every rank, stride = the largest block of any rank: the exchange buffer must hold the largest such product.
    bp_levels: the dependency levels of one sweep as lists of (src, dst) vertex pairs in sequence order (tests: tnqs_dbg_default_sequence_graph).
    Returns {"max_exchange_bytes", "bytes_gathered_per_layer", "exchanges_per_layer", "by_kind"}."""
    idx = graph.index
    world = max(owner) + 1
    r256 = lambda b: (int(b) + 255) & ~255
    n = d * chi
    kinds = {"bp_level": [], "gate_gram": [], "gate_record": []}
    for lev in bp_levels:
        # a level is cut into sub-batches by workspace bytes (engine_bp.cpp: 2 x the source site's tensor per message, whoever owns it, against TNQS_BP_WS_MB = 24 GiB
        # by default -- a 20 x 20 level of 760 messages x 32 MiB is cut in two), and every sub-batch is an exchange of its own
        start = 0
        while start < len(lev):
            blocks, used, end = [0] * world, 0, start
            while end < len(lev):
                need = 2 * d * chi ** graph.degree(lev[end][0]) * esz
                if end > start and used + need > bp_ws_bytes:
                    break
                used += need; blocks[owner[idx[lev[end][0]]]] += r256(chi * chi * esz); end += 1
            kinds["bp_level"].append(max(blocks) * world)
            start = end
    x2 = n * d * chi * esz
    for grp in colour_groups:
        gb, rb = [0] * world, [0] * world
        for (a, b) in grp:
            ra, rbk = owner[idx[a]], owner[idx[b]]
            if ra != rbk:
                gb[ra] += r256(n * n * 16); gb[rbk] += r256(n * n * 16)
            rb[ra] += r256(32 + chi * 8 + (x2 if ra != rbk else 0))
        if max(gb):
            kinds["gate_gram"].append(max(gb) * world)
        kinds["gate_record"].append(max(rb) * world)
    nup = len(colour_groups) + 1 if bp_updates_per_layer is None else bp_updates_per_layer
    per_layer = nup * sweeps_per_update * sum(kinds["bp_level"]) + sum(kinds["gate_gram"]) + sum(kinds["gate_record"])
    nex = nup * sweeps_per_update * len(kinds["bp_level"]) + len(kinds["gate_gram"]) + len(kinds["gate_record"])
    return {"max_exchange_bytes": max(max(v) if v else 0 for v in kinds.values()), "bytes_gathered_per_layer": per_layer, "exchanges_per_layer": nex,
            "by_kind": {k: {"count": len(v), "max": max(v) if v else 0, "sum": sum(v)} for k, v in kinds.items()}}


class Sharding:
    """keeps the exchange tensor and the ctypes callback alive for as long as any handle copy uses them"""

    def __init__(self, rank: int, world: int, owner: List[int], exch_bytes: int, group=None, device: Optional[int] = None):
        import torch
        import torch.distributed as dist
        self.rank, self.world, self.owner, self.group = rank, world, list(owner), group
        self.torch, self.dist = torch, dist
        dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.per_rank_cap = (exch_bytes + world - 1) // world
        self.buf = torch.empty(self.per_rank_cap * world, dtype=torch.uint8, device=dev)
        self.backend = dist.get_backend(group)
        self.n_exchanges = 0
        self.bytes_exchanged = 0
        # hooks around the host part of a gloo exchange (profiles/shard_proxy.py: ranks that share ONE GPU take turns on it -- a rank gives the device up
        # while it waits for its peers in the collective and takes it back before the gathered block is copied to the device)
        self.on_release = None
        self.on_acquire = None

        def _cb(ctx, base, bytes_per_rank, nranks):
            try:
                assert base == self.buf.data_ptr() and nranks == self.world
                n = int(bytes_per_rank)
                flat = self.buf[: n * nranks]
                mine = flat[self.rank * n:(self.rank + 1) * n]
                if self.backend == "nccl":
                    # RCCL all-gather straight between the GPUs' exchange buffers (xGMI); the send block is copied out first so
                    # that input and output never alias
                    self.dist.all_gather_into_tensor(flat, mine.clone(), group=self.group)
                    self.torch.cuda.synchronize()
                else:                                   # gloo: stage through the host (tests; ranks may share a GPU)
                    host = mine.cpu()
                    outs = [self.torch.empty_like(host) for _ in range(nranks)]
                    if self.on_release is not None:
                        self.on_release()
                    self.dist.all_gather(outs, host, group=self.group)
                    if self.on_acquire is not None:
                        self.on_acquire()
                    flat.copy_(self.torch.cat(outs).to(flat.device))
                    self.torch.cuda.synchronize()
                self.n_exchanges += 1
                self.bytes_exchanged += n * nranks
                return 0
            except Exception as e:                      # never let an exception cross the C boundary
                import sys
                print(f"[tnqs dist] all-gather callback failed: {e!r}", file=sys.stderr)
                return 1

        self.cb = L.ALLGATHER_FN(_cb)


def _prefer_torch_rccl():
    """PyTorch-ROCm bundles its own librccl.so (same SONAME as /opt/rocm's).  If the library loaded /opt/rocm's copy and `import torch` came
    later, the process would hold two RCCLs; pointing the library at torch's copy (TNQS_RCCL_LIB, read on its first RCCL call) makes both
    use one -- the same arrangement _lib.py makes for the HIP runtime.  torch itself is not imported here."""
    import os
    if os.environ.get("TNQS_RCCL_LIB"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is not None and spec.submodule_search_locations:
            path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "librccl.so")
            if os.path.exists(path):
                os.environ["TNQS_RCCL_LIB"] = path
    except Exception:
        pass


def rccl_preflight(selftest_device: Optional[int] = None) -> Optional[str]:
    """LOCAL check of the in-library RCCL transport, no collective: librccl.so loads and exports what is needed (tnqs_rccl_preflight) and,
    when a device is named, a one-rank communicator round trip runs on it (tnqs_rccl_selftest).  Returns None when fine, else the reason."""
    _prefer_torch_rccl()
    try:
        L.check(L.lib.tnqs_rccl_preflight())
        if selftest_device is not None:
            L.check(L.lib.tnqs_rccl_selftest(int(selftest_device), C.c_int64(1 << 16)))
    except Exception as e:                                      # noqa: BLE001 -- any failure means "no in-library RCCL in this process"
        return f"{type(e).__name__}: {e}"
    return None


def ranks_agree(ok: bool, group=None, device: Optional[int] = None) -> bool:
    """all ranks learn whether EVERY rank said ok (all_reduce MIN over the process group the host already has)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return bool(ok)
    on_gpu = dist.get_backend(group) == "nccl"
    if on_gpu and device is not None:
        torch.cuda.set_device(device)                           # collectives of an nccl group run on the current device
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=(f"cuda:{torch.cuda.current_device()}" if on_gpu else "cpu"))
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return int(flag.item()) == 1


class RcclSharding:
    """transport = RCCL inside the library (tnqs_set_sharding_rccl): nothing of the data path runs in Python.  Only the 128-byte
    ncclUniqueId travels through the host once: rank 0 creates it, `broadcast` hands it to the other ranks (default:
    torch.distributed.broadcast_object_list on whatever backend the process group has -- gloo is fine, it is not the data path).
    No rank may fail ALONE before the collective communicator set-up (its peers would wait for it forever): every rank runs the local
    preflight first and the ranks agree on the outcome; rank 0 broadcasts an error marker instead of the id if it cannot create one."""

    def __init__(self, bpc, rank: int, world: int, owner: List[int], exch_bytes: int, group=None, broadcast=None):
        import weakref
        _prefer_torch_rccl()
        self.rank, self.world, self.owner, self.group = rank, world, list(owner), group
        self._ref = weakref.ref(bpc)              # any live handle of the family will do for the counters (core.copy re-attaches)
        self._last = (0, 0)
        why = rccl_preflight()
        if world > 1 and broadcast is None:
            if not ranks_agree(why is None, group=group, device=bpc.device):
                raise L.TnqsError("RCCL transport unavailable on at least one rank" + (f" (this rank: {why})" if why else ""))
        elif why is not None:
            raise L.TnqsError("RCCL transport unavailable: " + why)
        uid = C.create_string_buffer(128)
        payload = [b""]
        if rank == 0:
            try:
                L.check(L.lib.tnqs_rccl_unique_id(uid))
                payload = [bytes(uid.raw)]
            except Exception as e:                              # noqa: BLE001 -- the other ranks must hear about it, not wait for an id
                payload = [("ERR " + repr(e)).encode()]
        if world > 1:
            if broadcast is not None:
                payload = [broadcast(payload[0])]
            else:
                import torch.distributed as dist
                dist.broadcast_object_list(payload, src=0, group=group)
        if len(payload[0]) != 128:
            raise L.TnqsError("RCCL: rank 0 could not create the communicator id: " + payload[0].decode("utf-8", "replace"))
        self.uid = C.create_string_buffer(payload[0], 128)
        ow, owp = L.i32(owner)
        L.check(L.lib.tnqs_set_sharding_rccl(bpc._h, rank, world, owp, self.uid, C.c_int64(int(exch_bytes))))

    def attach(self, bpc):
        import weakref
        self._ref = weakref.ref(bpc)

    def _stats(self):
        bpc = self._ref()
        if bpc is not None and getattr(bpc, "_h", None) is not None:
            n, b = C.c_int64(), C.c_int64()
            L.check(L.lib.tnqs_sharding_stats(bpc._h, C.byref(n), C.byref(b)))
            self._last = (n.value, b.value)
        return self._last

    @property
    def n_exchanges(self) -> int:
        return self._stats()[0]

    @property
    def bytes_exchanged(self) -> int:
        return self._stats()[1]


def rccl_selftest(device: int = 0, nbytes: int = 1 << 20):
    """one-rank round trip through the library's RCCL transport (a single GPU cannot host two RCCL ranks)"""
    _prefer_torch_rccl()
    L.check(L.lib.tnqs_rccl_selftest(int(device), C.c_int64(int(nbytes))))


def shard(bpc, rank: int, world: int, owner: Optional[List[int]] = None, exch_bytes: Optional[int] = None, group=None,
          max_chi: int = 64, transport: Optional[str] = None, broadcast=None, balance_chi: Optional[int] = None):
    """attach vertex sharding to a freshly created cache (call on every rank, before uploading site tensors).
    transport = "rccl": the library's own RCCL all-gather on its stream (production; one rank per GPU);
                "callback": torch.distributed through a host callback (gloo tests in which ranks share a GPU);
                None: "rccl" when the process group's backend is nccl, else "callback".
    owner = None: contiguous blocks balanced by WORK -- site-tensor elements at bond dimension `balance_chi` (default: max_chi, the
    bond dimension the evolution saturates at); pass an explicit owner list for anything else."""
    g = bpc.graph
    if owner is None:
        owner = partition_vertices(g.nv(), world, site_weights(g, int(balance_chi or max_chi)))
    if exch_bytes is None:
        exch_bytes = world * exchange_bytes_needed(max_chi, 2, g.ne(), g.nv(), 8 if np.dtype(bpc.dtype) in (np.dtype(np.complex64), np.dtype(np.float32)) else 16) // max(1, world // 2)
    if transport is None:
        transport = "callback"
        try:
            import torch.distributed as dist
            if dist.is_initialized() and dist.get_backend(group) == "nccl":
                transport = "rccl"
        except Exception:
            pass
    if transport == "rccl":
        sh = RcclSharding(bpc, rank, world, owner, exch_bytes, group=group, broadcast=broadcast)
        bpc._shard = sh
        return sh
    if transport != "callback":
        raise ValueError(f"shard: unknown transport {transport!r}")
    sh = Sharding(rank, world, owner, exch_bytes, group=group, device=bpc.device)
    ow, owp = L.i32(owner)
    L.check(L.lib.tnqs_set_sharding(bpc._h, rank, world, owp, sh.cb, None, C.c_void_p(sh.buf.data_ptr()),
                                    C.c_int64(sh.buf.numel())))
    bpc._shard = sh
    return sh
