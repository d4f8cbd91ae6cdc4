"""CPU baseline of the hot path (TEST / BENCH INFRASTRUCTURE ONLY -- never imported by the product).

The parity oracle (tnqs_oracle.py) is written for line-by-line correspondence with the reference and runs one gate and one message at
a time; timing it says nothing about what a CPU can do.  This module is the same algorithm organised the way a tuned CPU code would run
it on a many-core host, so that `bench.py`'s `cpu_baseline` can adjudicate a GPU / CPU ratio:

  * the arithmetic is the oracle's (simple_update, pseudo_sqrt_inv_sqrt, truncate_spectrum, apply_gate are CALLED, not re-derived), i.e.
    the reference's own contraction sequence: mode products as GEMMs (no transposition copies: batched `matmul` on views), thin QR and
    SVD through LAPACK (`geqrf/ungqr`, `gesdd`), Hermitian eigen in f64 -- src/Apply/simple_update.jl:21-77, abstract...:162-190;
  * the schedule is the reference's (apply_gates.jl:46-98: a BP update before every colour group, a final one; Gauss-Seidel sweeps over
    the cache's edge sequence), executed the way the device engine executes it: the messages of a sweep are grouped into dependency
    levels (a message only waits for the messages it reads that precede it in the sequence -- same values as the sequential loop), the
    gates of a colour group are vertex-disjoint; the members of a level / group run on a thread pool (numpy releases the GIL inside
    BLAS / LAPACK and its copy loops), and when there are fewer members than cores the contractions of a member are chunked onto the
    idle workers of the same pool; BLAS itself runs single-threaded per call.

`measure()` also measures what the host's BLAS delivers on (a) a large square ComplexF32 GEMM with all threads and (b) the mode-product
shape of the workload ((2 chi^3) x chi times chi x chi) on the same thread pool, and reports the layer's algorithmic GFLOP/s
(384 chi^5 flop per gate + 64 chi^5 per message) as a fraction of both."""
from __future__ import annotations

import os
import time
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Optional, Tuple

import numpy as np

import tnqs_oracle as o


class _Serial:
    """stand-in for the oracle's internal chunk pool: inside this module the parallelism is ACROSS sites, not inside one contraction"""
    _max_workers = 1

    @staticmethod
    def map(fn, it):
        return map(fn, it)


def _levels(g: o.Graph, seq: List[Tuple]) -> List[List[int]]:
    """dependency levels of one Gauss-Seidel sweep over `seq` (positions): message (u -> v) reads (k -> u), k != v; a read of a message
    that comes EARLIER in the sequence sees this sweep's value and orders the two, a read of a later one sees the previous sweep's"""
    pos = {e: t for t, e in enumerate(seq)}
    level = [0] * len(seq)
    for t, (u, v) in enumerate(seq):
        lv = 0
        for k in g.nbrs[u]:
            if k == v:
                continue
            p = pos.get((k, u))
            if p is not None and p < t:
                lv = max(lv, level[p] + 1)
        level[t] = lv
    out: List[List[int]] = [[] for _ in range(max(level) + 1 if level else 0)]
    for t, lv in enumerate(level):
        out[lv].append(t)
    return out


def _message(bpc, fresh: Dict, pos: Dict, t: int, e, normalize=True) -> np.ndarray:
    """updated_message (abstract...:162-190) reading, for every incoming message, this sweep's value when it precedes position t in the
    sequence and the previous sweep's otherwise -- exactly what the sequential loop of update_iteration! (:204-218) sees"""
    u, v = e
    g = bpc.g
    psi = bpc.tns.tensors[u]
    tt = psi
    for k in g.nbrs[u]:
        if k == v:
            continue
        p = pos.get((k, u))
        m = fresh[(k, u)] if (p is not None and p < t and (k, u) in fresh) else bpc.message((k, u))
        tt = o._absorb(tt, g.leg(u, k), m)
    m = o._gram(tt, psi, g.leg(u, v))
    if normalize:
        s = m.sum()
        if s != 0:
            m = m / s
    return m.astype(psi.dtype, copy=False)


class _Inner:
    """the oracle's chunked contractions (tnqs_oracle._absorb / _gram split their loop over the untouched indices into 4 x _max_workers
    chunks) running on the SAME thread pool as the tasks that call them: a level / colour group with fewer members than cores gives every
    member cores / members workers' worth of chunks.  BLAS itself stays single-threaded: concurrent callers of a multi-threaded OpenBLAS
    queue up behind its global lock (measured: 2 BLAS threads x 64 tasks ran 1.6x SLOWER than 1 x 64).  No deadlock: the members of a map
    (<= cores / 2 of them block waiting for their chunks) leave at least as many workers free."""

    def __init__(self, pool: ThreadPoolExecutor, width: int):
        self._pool, self._max_workers = pool, max(1, width)

    def map(self, fn, it):
        return self._pool.map(fn, it)


def _inner(pool: ThreadPoolExecutor, ntasks: int):
    width = pool._max_workers // max(1, ntasks) // 2
    return _Inner(pool, width) if (width >= 1 and ntasks * 2 <= pool._max_workers) else _Serial()


def update(bpc, pool: ThreadPoolExecutor, maxiter: Optional[int] = None, tolerance: Optional[float] = None, info: Optional[dict] = None):
    """update (abstract...:223-259) with the messages of a dependency level computed concurrently"""
    dk = bpc.default_bp_update_kwargs()
    maxiter = dk["maxiter"] if maxiter is None else maxiter
    seq = bpc.edge_sequence
    pos = {e: t for t, e in enumerate(seq)}
    levels = _levels(bpc.g, seq)
    bpc = bpc.copy()
    niter = maxiter
    for it in range(1, maxiter + 1):
        fresh: Dict = {}
        diffs = [0.0] * len(seq)
        for lev in levels:
            o._POOL = _inner(pool, len(lev))
            res = list(pool.map(lambda t: _message(bpc, fresh, pos, t, seq[t]), lev))
            for t, m in zip(lev, res):
                if tolerance is not None:
                    diffs[t] = o.message_diff(m, bpc.message(seq[t]))
                fresh[seq[t]] = m
        bpc.messages.update(fresh)
        if tolerance is not None and sum(diffs) / len(seq) <= tolerance:
            niter = it
            break
    if info is not None:
        info["niter"] = niter
    return bpc


def apply_layer(bpc, one_site: List, colour_groups: List[List], pool: ThreadPoolExecutor, apply_kwargs: dict, bp_kwargs: dict):
    """apply_gates (apply_gates.jl:46-98) for a Trotter layer [one-site gates on every vertex] + [two-site gates by edge colour]: the
    walk over the gate list triggers a BP update in front of every colour group (its first gate touches an affected vertex, :68-78) and a
    final one (:93-95); the gates between two updates are vertex-disjoint and run concurrently.  Returns (cache, errors, sweeps)."""
    bpc = bpc.copy()
    sweeps = []

    def one(gate):
        mat, verts = o.resolve_gate(gate)
        return o.apply_gate(bpc, mat, verts, **apply_kwargs)
    o._POOL = _inner(pool, len(one_site))
    list(pool.map(one, one_site))
    errs = []
    for grp in colour_groups:
        inf = {}
        bpc = update(bpc, pool, info=inf, **bp_kwargs)
        sweeps.append(inf["niter"])
        # apply_gate mutates bpc (tensors of its two vertices, the two messages of its bond): disjoint for the gates of one colour

        def two(gate, b=bpc):
            mat, verts = o.resolve_gate(gate)
            return o.apply_gate(b, mat, verts, **apply_kwargs)
        o._POOL = _inner(pool, len(grp))
        errs += list(pool.map(two, grp))
    inf = {}
    bpc = update(bpc, pool, info=inf, **bp_kwargs)
    sweeps.append(inf["niter"])
    return bpc, np.array(errs), sweeps


def _gemm_rates(chi: int, nthreads: int, pool: ThreadPoolExecutor) -> dict:
    """what the host BLAS delivers: (a) one large square ComplexF32 GEMM on all threads, (b) the workload's mode-product shape
    ((2 chi^3) x chi @ chi x chi, out of cache) on the thread pool, one BLAS thread each"""
    from threadpoolctl import threadpool_limits
    rng = np.random.default_rng(0)
    n = 3072
    a = (rng.standard_normal((n, n), dtype=np.float32) + 1j * rng.standard_normal((n, n), dtype=np.float32)).astype(np.complex64)
    b = a.T.copy()
    with threadpool_limits(limits=nthreads):
        a @ b
        t0 = time.perf_counter(); a @ b; a @ b; dt = (time.perf_counter() - t0) / 2
    square = 8.0 * n ** 3 / dt / 1e9
    rows = 2 * chi ** 3
    x = [(rng.standard_normal((rows, chi), dtype=np.float32) + 1j).astype(np.complex64) for _ in range(nthreads)]
    m = (rng.standard_normal((chi, chi), dtype=np.float32) + 1j).astype(np.complex64)
    with threadpool_limits(limits=1):
        list(pool.map(lambda t: t @ m, x))
        reps = 4
        t0 = time.perf_counter()
        for _ in range(reps):
            list(pool.map(lambda t: t @ m, x))
        dt = (time.perf_counter() - t0) / reps
    skinny = 8.0 * rows * chi * chi * nthreads / dt / 1e9
    return {"square_cgemm_gflops": round(square, 1), "mode_product_shape_gflops": round(skinny, 1)}


def _keep_big_blocks_on_the_heap():
    """glibc hands every allocation above 128 KiB to mmap and gives it back on free: each 16 MB temporary of a contraction is mapped, page-
    faulted in (4096 faults) and unmapped again -- with 64 threads doing that at once the kernel's mm lock, not the arithmetic, sets the
    pace.  Raising the mmap / trim thresholds keeps those blocks in the malloc arenas, where they are reused warm.  (What a compiled CPU
    code gets for free by owning its workspaces.)"""
    try:
        import ctypes
        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-3, 1 << 30)        # M_MMAP_THRESHOLD
        libc.mallopt(-1, 1 << 30)        # M_TRIM_THRESHOLD (int argument)
        libc.mallopt(-2, 1 << 28)        # M_TOP_PAD
        return True
    except Exception:
        return False


class parallel_oracle:
    """context manager for callers other than measure() (the parity tests at the 268 MB-per-site shapes): a thread pool for update() /
    apply_layer(), single-threaded BLAS per call, every site-tensor contraction on the copy-free batched-matmul path; restores the
    oracle's module state on exit.  `with parallel_oracle() as pool: bpc = update(bpc, pool, ...)`"""

    def __init__(self, nthreads: Optional[int] = None):
        self.nthreads = nthreads or max(1, min(64, (os.cpu_count() or 2) // 2))

    def __enter__(self):
        from threadpoolctl import threadpool_limits
        if os.environ.get("TNQS_CPU_NO_MALLOPT") != "1":
            _keep_big_blocks_on_the_heap()
        self._saved = (o._POOL, o._BIG)
        o._POOL, o._BIG = _Serial(), 1 << 12
        self._pool = ThreadPoolExecutor(max_workers=self.nthreads)
        self._limits = threadpool_limits(limits=1)
        self._limits.__enter__()
        return self._pool

    def __exit__(self, *exc):
        self._limits.__exit__(*exc)
        self._pool.shutdown()
        o._POOL, o._BIG = self._saved
        return False


def measure(chi: int = 32, L: int = 8, nthreads: Optional[int] = None, seed: int = 1234, nlayers: int = 1, periodic: bool = True) -> dict:
    """one TFIM layer (README.md:42-48 angles) on an L x L lattice at bond dimension chi, ComplexF32, from BP-converged messages,
    reference-default bp_update_kwargs.  periodic = True: the torus (every site has the bulk degree 4, L^2 sites, 2 L^2 edges, four colours
    for even L); periodic = False: the open lattice of the benchmark itself (BASELINE.json configs[1] at L = 20: 400 sites, 760 edges)."""
    from threadpoolctl import threadpool_limits
    # physical cores on an SMT-2 host, capped at 64: numpy's bundled OpenBLAS is built for at most 64 concurrent callers (NUM_THREADS = 64;
    # beyond that it warns, and with the nested chunk maps of this module it crashed on the 128-core box)
    nthreads = nthreads or max(1, min(64, (os.cpu_count() or 2) // 2))
    if os.environ.get("TNQS_CPU_NO_MALLOPT") != "1":
        _keep_big_blocks_on_the_heap()
    g = o.named_grid((L, L), periodic=periodic)
    groups = o.edge_color(g)
    one_site = [("Rx", [v], 2 * 2.5 * 0.01) for v in g.vertices]
    colour_groups = [[("Rzz", [a, b], 2 * 1.0 * 0.01) for (a, b) in grp] for grp in groups]
    rng = np.random.default_rng(seed)
    tensors = {}
    for v in g.vertices:
        shp = (2,) + (chi,) * g.degree(v)
        n = int(np.prod(shp))
        tensors[v] = rng.standard_normal(2 * n, dtype=np.float32).view(np.complex64).reshape(shp) / np.float32(np.sqrt(n))
    # edge sequence: per colour, all messages a -> b, then all b -> a.  Any sequence is a valid Gauss-Seidel order (abstract...:204-218); the
    # reference's default (a DFS post-order over spanning forests) chains almost every message behind another one, this one gives 8 levels of
    # |E| / 4 independent messages per sweep -- the order a CPU code that wants to use its cores would pick
    seq = []
    for grp in groups:
        seq += [(a, b) for (a, b) in grp] + [(b, a) for (a, b) in grp]
    bpc = o.BeliefPropagationCache(o.TensorNetworkState(g, tensors), edge_sequence=seq)
    kw = dict(maxdim=chi, cutoff=1e-10, normalize_tensors=True)
    bpkw = dict(bpc.default_bp_update_kwargs())
    saved = (o._POOL, o._BIG)
    o._POOL, o._BIG = _Serial(), 1 << 12          # every site-tensor contraction takes the copy-free batched-matmul path, serial inside
    prof: Dict[str, float] = {}
    unwrap = []
    if os.environ.get("TNQS_CPU_PROFILE") == "1":     # thread-seconds per primitive (sum over the pool's threads)
        import threading
        lock = threading.Lock()

        def wrap(mod, name):
            fn = getattr(mod, name)

            def timed(*a, **k):
                t = time.perf_counter()
                try:
                    return fn(*a, **k)
                finally:
                    dt_ = time.perf_counter() - t
                    with lock:
                        prof[name] = prof.get(name, 0.0) + dt_
            setattr(mod, name, timed)
            unwrap.append((mod, name, fn))
        for mod, name in ((o, "_absorb"), (o, "_gram"), (np.linalg, "qr"), (np.linalg, "svd"), (np.linalg, "eigh"), (o, "apply_gate")):
            wrap(mod, name)
    try:
        with ThreadPoolExecutor(max_workers=nthreads) as pool:
            rates = _gemm_rates(chi, nthreads, pool)
            with threadpool_limits(limits=1):
                bpc = update(bpc, pool, **bpkw)                        # warm-up outside the timing: converged messages
                t0 = time.perf_counter()
                sweeps_all = []
                for _ in range(nlayers):
                    bpc, errs, sweeps = apply_layer(bpc, one_site, colour_groups, pool, kw, bpkw)
                    sweeps_all.append(sweeps)
                dt = (time.perf_counter() - t0) / nlayers
    finally:
        o._POOL, o._BIG = saved
        for mod, name, fn in unwrap:
            setattr(mod, name, fn)
    n2 = len(g.edges)
    nsweeps = float(np.mean([sum(s) for s in sweeps_all]))
    # SURVEY.md 8d per site degree z (n = z - 1 absorbed legs): a message costs (n + 1) d chi^(z+1) cMAC, a gate 2 (2 n d + 3 d^2) chi^(z+1) per
    # PAIR of bulk sites, i.e. (2 n d + 3 d^2) chi^(z+1) per site; on the torus every site has z = 4 (384 chi^5 flop per gate, 64 chi^5 per message)
    flops = 0.0
    for (a, b) in g.edges:
        for v in (a, b):
            z = g.degree(v); flops += 8.0 * (2 * (z - 1) * 2 + 12) * chi ** (z + 1) + nsweeps * 8.0 * z * 2 * chi ** (z + 1)
    gf = flops / dt / 1e9
    return {"gates_per_s": n2 / dt, "seconds_per_layer": dt, "n_two_site": n2, "sites": len(g.vertices), "threads": nthreads,
            "bp_sweeps": sweeps_all[-1], "algorithmic_gflops": round(gf, 1), **rates,
            "frac_of_square_cgemm": round(gf / rates["square_cgemm_gflops"], 3),
            "frac_of_mode_product_shape": round(gf / rates["mode_product_shape_gflops"], 3),
            "max_truncation_error": float(np.max(errs)) if len(errs) else 0.0,
            **({"profile_thread_seconds": {k: round(v, 2) for k, v in prof.items()}} if prof else {})}


def measure_host(chi: int = 32, L: int = 20, periodic: bool = False, nproc: Optional[int] = None, seed: int = 1234) -> dict:
    """the whole host: numpy's OpenBLAS admits at most 64 concurrent callers per process, so a 128-core box is driven by `nproc` PROCESSES of
    64 threads, each pinned to its own block of cores (one socket each on the 2 x 64-core GPU box) and each running one layer of its own copy
    of the workload at the same time.  Reported: the AGGREGATE rate (sum of the processes' gates over the common wall time) and the
    per-process figures -- a throughput baseline; the latency of a single layer is that of one process."""
    import json
    import subprocess
    import sys
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    from cpu_port import cpu_budget
    ncores, _ = cpu_budget()                                                     # physical cores, capped by the cgroup's CPU quota
    nproc = nproc or max(1, min(4, ncores // 64))
    per = max(1, min(64, ncores // nproc))
    procs = []
    t0 = time.perf_counter()
    for r in range(nproc):
        mask = cpus[r * per:(r + 1) * per]                                  # first hardware thread of consecutive cores (Linux numbers siblings last)
        code = ("import os, sys, json; os.sched_setaffinity(0, %r); sys.path.insert(0, %r); import cpu_layer; "
                "print(json.dumps(cpu_layer.measure(%d, %d, nthreads=%d, seed=%d, periodic=%r)))"
                % (set(mask), os.path.dirname(os.path.abspath(__file__)), chi, L, per, seed + r, periodic))
        procs.append(subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for pr in procs:
        so, se = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError("cpu_layer worker failed: " + se[-2000:])
        outs.append(json.loads(so.strip().splitlines()[-1]))
    wall = time.perf_counter() - t0
    slowest = max(x["seconds_per_layer"] for x in outs)
    total_gates = sum(x["n_two_site"] for x in outs)
    agg = dict(outs[0])
    agg.update(gates_per_s=total_gates / slowest, seconds_per_layer=slowest, threads=per * nproc, processes=nproc, threads_per_process=per,
               per_process_gates_per_s=[round(x["gates_per_s"], 2) for x in outs], algorithmic_gflops=round(sum(x["algorithmic_gflops"] for x in outs), 1),
               wall_seconds_incl_setup=round(wall, 1), periodic=periodic, L=L)
    agg["frac_of_square_cgemm"] = round(agg["algorithmic_gflops"] / (nproc * outs[0]["square_cgemm_gflops"]), 3)
    agg["frac_of_mode_product_shape"] = round(agg["algorithmic_gflops"] / (nproc * outs[0]["mode_product_shape_gflops"]), 3)
    return agg


if __name__ == "__main__":
    import json
    import sys
    chi = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    if len(sys.argv) > 3 and sys.argv[3] == "host":
        print(json.dumps(measure_host(chi, L)))
    else:
        print(json.dumps(measure(chi, L, periodic=not (len(sys.argv) > 3 and sys.argv[3] == "open"))))
