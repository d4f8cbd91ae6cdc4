"""Host side of the COMPILED CPU restatement (oracle/cpu_port.cpp).  TEST / BENCH INFRASTRUCTURE ONLY -- never imported by the product.

`build()` compiles cpu_port.cpp with g++ -O3 -fopenmp into oracle/_cport/libtnqs_cpu.so (git-ignored; __graft_entry__.build() calls it; the built file travels
to the GPU box with the snapshot).  The library takes its BLAS / LAPACK from the OpenBLAS that ships with scipy (dlopen at run time, one BLAS thread per call).
`CpuNet` mirrors an oracle cache (tnqs_oracle.BeliefPropagationCache) on the compiled side; `update`, `apply_layer` run the reference's schedule
(apply_gates.jl:46-98: a BP update in front of every colour group and a final one) with the messages of a dependency level and the gates of a colour group on the
OpenMP team.  `measure` / `measure_host` time one TFIM layer of the benchmark workload: bench.py's `cpu_baseline` (kind "port", compiled)."""
from __future__ import annotations

import ctypes as C
import glob
import os
import subprocess
import sys
import time
from typing import Dict, List, Optional, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_cport", "libtnqs_cpu.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "cpu_port.cpp")
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        subprocess.run(["g++", "-O3", "-march=x86-64-v3", "-fopenmp", "-fPIC", "-shared", "-std=c++17", src, "-ldl", "-o", SO], check=True)
    return SO


def _blas_path() -> str:
    import scipy
    cands = glob.glob(os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs", "libscipy_openblas*.so"))
    cands = [c for c in cands if "64_" not in os.path.basename(c)] or cands          # the LP64 build (32-bit integers: LAPACKE's default interface)
    if not cands:
        raise RuntimeError("cpu_port: scipy's bundled OpenBLAS not found")
    return cands[0]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            build()
        L = C.CDLL(SO)
        L.tnqs_cpu_last_error.restype = C.c_char_p
        L.tnqs_cpu_create.restype = C.c_void_p
        L.tnqs_cpu_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.tnqs_cpu_destroy.argtypes = [C.c_void_p]
        L.tnqs_cpu_set_tensor.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.tnqs_cpu_tensor_dims.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.tnqs_cpu_get_tensor.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.tnqs_cpu_set_message.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.tnqs_cpu_get_message.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.tnqs_cpu_update.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
        L.tnqs_cpu_apply_two_site.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p]
        L.tnqs_cpu_apply_one_site.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.tnqs_cpu_timers.argtypes = [C.c_void_p, C.c_int]
        if L.tnqs_cpu_init(_blas_path().encode()) != 0:
            raise RuntimeError("cpu_port: " + L.tnqs_cpu_last_error().decode())
        L.tnqs_cpu_set_threads(max(1, min(16, (os.cpu_count() or 2) // 2)))       # (measure() sets its own; an OpenMP default of 256 threads on tiny test lattices only spins)
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        raise RuntimeError("cpu_port: " + lib().tnqs_cpu_last_error().decode())


def _levels(nbrs: List[List[int]], seq: List[Tuple[int, int]]) -> List[List[int]]:
    """dependency levels of one Gauss-Seidel sweep (positions): a read of a message that comes EARLIER in the sequence orders the two"""
    pos = {e: t for t, e in enumerate(seq)}
    level = [0] * len(seq)
    for t, (u, v) in enumerate(seq):
        lv = 0
        for k in nbrs[u]:
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


class CpuNet:
    """a ComplexF32 tensor network + BP messages on the compiled side; vertices are positions in `vertices`"""

    def __init__(self, vertices: List, nbrs: Dict, d: int = 2):
        self.vertices = list(vertices)
        self.index = {v: i for i, v in enumerate(self.vertices)}
        self.nbrs = [[self.index[w] for w in nbrs[v]] for v in self.vertices]
        deg = np.array([len(n) for n in self.nbrs], dtype=np.int32)
        flat = np.array([w for n in self.nbrs for w in n] or [0], dtype=np.int32)
        self.d = d
        self._h = lib().tnqs_cpu_create(len(self.vertices), d, deg.ctypes.data, flat.ctypes.data)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.tnqs_cpu_destroy(self._h)
            self._h = None

    @classmethod
    def from_oracle(cls, bpc):
        g = bpc.g
        net = cls(g.vertices, g.nbrs, d=bpc.tns.tensors[g.vertices[0]].shape[0])
        for v in g.vertices:
            net.set_tensor(v, bpc.tns.tensors[v])
        for (a, b), m in bpc.messages.items():
            net.set_message(a, b, m)
        return net

    def set_tensor(self, v, t):
        t = np.ascontiguousarray(t, dtype=np.complex64)
        dims = np.array(t.shape, dtype=np.int32)
        _chk(lib().tnqs_cpu_set_tensor(self._h, self.index[v], t.ctypes.data, dims.ctypes.data))

    def tensor(self, v) -> np.ndarray:
        dims = np.zeros(16, dtype=np.int32)
        nd = lib().tnqs_cpu_tensor_dims(self._h, self.index[v], dims.ctypes.data)
        out = np.empty(tuple(int(x) for x in dims[:nd]), dtype=np.complex64)
        lib().tnqs_cpu_get_tensor(self._h, self.index[v], out.ctypes.data)
        return out

    def bond_dim(self, u, v) -> int:
        dims = np.zeros(16, dtype=np.int32)
        lib().tnqs_cpu_tensor_dims(self._h, self.index[u], dims.ctypes.data)
        return int(dims[1 + self.nbrs[self.index[u]].index(self.index[v])])

    def set_message(self, a, b, m):
        m = np.ascontiguousarray(m, dtype=np.complex64)
        lib().tnqs_cpu_set_message(self._h, self.index[a], self.index[b], m.ctypes.data, m.shape[0])

    def message(self, a, b) -> np.ndarray:
        chi = self.bond_dim(a, b)
        out = np.empty((chi, chi), dtype=np.complex64)
        lib().tnqs_cpu_get_message(self._h, self.index[a], self.index[b], out.ctypes.data, chi)
        return out

    def update(self, seq, maxiter: int = 25, tolerance: Optional[float] = 1e-5, normalize: bool = True) -> Tuple[int, float]:
        s = [(self.index[a], self.index[b]) for (a, b) in seq]
        lev = _levels(self.nbrs, s)
        su = np.array([a for a, _ in s], dtype=np.int32); sv = np.array([b for _, b in s], dtype=np.int32)
        off = np.cumsum([0] + [len(x) for x in lev]).astype(np.int32); order = np.array([t for x in lev for t in x], dtype=np.int32)
        niter = C.c_int(0); diff = C.c_double(0.0)
        _chk(lib().tnqs_cpu_update(self._h, len(s), su.ctypes.data, sv.ctypes.data, len(lev), off.ctypes.data, order.ctypes.data, int(maxiter),
                                   float(-1.0 if tolerance is None else tolerance), 1 if normalize else 0, C.addressof(niter), C.addressof(diff)))
        return niter.value, diff.value

    def apply_two_site(self, pairs, mats, maxdim: int, cutoff: float, normalize: bool = True) -> np.ndarray:
        v1 = np.array([self.index[a] for a, _ in pairs], dtype=np.int32); v2 = np.array([self.index[b] for _, b in pairs], dtype=np.int32)
        g = np.ascontiguousarray(np.stack(mats), dtype=np.complex128)
        errs = np.zeros(len(pairs), dtype=np.float64)
        _chk(lib().tnqs_cpu_apply_two_site(self._h, len(pairs), v1.ctypes.data, v2.ctypes.data, g.ctypes.data, int(maxdim or 0),
                                           float(-1.0 if cutoff is None else cutoff), 1 if normalize else 0, errs.ctypes.data))
        return errs

    def apply_one_site(self, verts, mats, normalize: bool = True):
        vs = np.array([self.index[v] for v in verts], dtype=np.int32)
        g = np.ascontiguousarray(np.stack(mats), dtype=np.complex128)
        _chk(lib().tnqs_cpu_apply_one_site(self._h, len(verts), vs.ctypes.data, g.ctypes.data, 1 if normalize else 0))


def apply_layer(net: CpuNet, one_site, colour_groups, seq, apply_kwargs: dict, maxiter: int = 25, tolerance: Optional[float] = 1e-5):
    """apply_gates (apply_gates.jl:46-98) for a Trotter layer [one-site gates on every vertex] + [two-site gates by edge colour]: an update in front of every
    colour group (its first gate touches an affected vertex, :68-78) and a final one (:93-95).  Returns (errors, sweeps per update)."""
    import tnqs_oracle as o
    mats, verts = zip(*[o.resolve_gate(g) for g in one_site]) if one_site else ((), ())
    if one_site:
        net.apply_one_site([v[0] for v in verts], list(mats), normalize=apply_kwargs.get("normalize_tensors", True))
    errs, sweeps = [], []
    for grp in colour_groups:
        sweeps.append(net.update(seq, maxiter, tolerance)[0])
        mats, verts = zip(*[o.resolve_gate(g) for g in grp])
        errs += list(net.apply_two_site([tuple(v) for v in verts], list(mats), apply_kwargs.get("maxdim"), apply_kwargs.get("cutoff"), apply_kwargs.get("normalize_tensors", True)))
    sweeps.append(net.update(seq, maxiter, tolerance)[0])
    return np.array(errs), sweeps


def measure(chi: int = 32, L: int = 8, nthreads: Optional[int] = None, seed: int = 1234, nlayers: int = 1, periodic: bool = False) -> dict:
    """one TFIM layer (README.md:42-48 angles) of the benchmark workload on the compiled port: L x L lattice at bond dimension chi, ComplexF32, iid site tensors,
    BP-converged messages, reference-default bp_update_kwargs (maxiter 25, tolerance 1e-5); the sweep order is the colour order a CPU code that wants to use its
    cores would pick (per colour all messages a -> b, then all b -> a: 8 levels of |E| / 4 independent messages) -- any order is a valid Gauss-Seidel order."""
    import tnqs_oracle as o
    nthreads = nthreads or max(1, min(64, (os.cpu_count() or 2) // 2))
    lib().tnqs_cpu_set_threads(nthreads)
    g = o.named_grid((L, L), periodic=periodic)
    groups = o.edge_color(g)
    one_site = [("Rx", [v], 2 * 2.5 * 0.01) for v in g.vertices]
    colour_groups = [[("Rzz", [a, b], 2 * 1.0 * 0.01) for (a, b) in grp] for grp in groups]
    rng = np.random.default_rng(seed)
    net = CpuNet(g.vertices, g.nbrs)
    for v in g.vertices:
        shp = (2,) + (chi,) * g.degree(v)
        n = int(np.prod(shp))
        net.set_tensor(v, rng.standard_normal(2 * n, dtype=np.float32).view(np.complex64).reshape(shp) / np.float32(np.sqrt(n)))
    seq = []
    for grp in groups:
        seq += [(a, b) for (a, b) in grp] + [(b, a) for (a, b) in grp]
    kw = dict(maxdim=chi, cutoff=1e-10, normalize_tensors=True)
    net.update(seq, 25, 1e-5)                                       # warm-up outside the timing: converged messages
    tm = np.zeros(7); lib().tnqs_cpu_timers(tm.ctypes.data, 1)
    t0 = time.perf_counter()
    sweeps_all = []
    for _ in range(nlayers):
        errs, sweeps = apply_layer(net, one_site, colour_groups, seq, kw)
        sweeps_all.append(sweeps)
    dt = (time.perf_counter() - t0) / nlayers
    lib().tnqs_cpu_timers(tm.ctypes.data, 0)
    n2 = len(g.edges)
    nsweeps = float(np.mean([sum(s) for s in sweeps_all]))
    flops = 0.0                                                     # SURVEY.md 8d per site degree z (as oracle/cpu_layer.py counts them)
    for (a, b) in g.edges:
        for v in (a, b):
            z = g.degree(v); flops += 8.0 * (2 * (z - 1) * 2 + 12) * chi ** (z + 1) + nsweeps * 8.0 * z * 2 * chi ** (z + 1)
    return {"gates_per_s": n2 / dt, "seconds_per_layer": dt, "n_two_site": n2, "sites": len(g.vertices), "threads": nthreads, "bp_sweeps": sweeps_all[-1],
            "algorithmic_gflops": round(flops / dt / 1e9, 1), "max_truncation_error": float(np.max(errs)) if len(errs) else 0.0,
            "thread_seconds_per_layer": dict(zip(("transpose", "gemm", "permute", "qr", "small"), (round(float(x) / nlayers, 2) for x in tm[:5]))),
            "wall_seconds_per_layer": {"update": round(float(tm[5]) / nlayers, 2), "two_site": round(float(tm[6]) / nlayers, 2)}}


def cpu_budget() -> Tuple[int, Optional[float]]:
    """(threads worth starting, CFS quota in CPUs or None): physical cores in the affinity mask (SMT-2: half the hardware threads), capped by the cgroup's CPU quota.
    The GPU boxes show all 256 hardware threads of the 2 x 64-core host but run under cpu.max = 16 CPUs: a 128-thread team there is throttled for most of every
    period (measured: 194 s per 20 x 20 layer with 128 threads, 24 s with 16)."""
    try:
        cpus = len(os.sched_getaffinity(0))
    except AttributeError:
        cpus = os.cpu_count() or 1
    cores = max(1, cpus // 2) if cpus >= 4 else cpus
    quota = None
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: None if t.split()[0] == "max" else float(t.split()[0]) / float(t.split()[1])),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: None if int(t) <= 0 else int(t) / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))):
        try:
            with open(path) as f:
                quota = parse(f.read().strip())
            break
        except (OSError, ValueError, IndexError):
            continue
    if quota:
        cores = max(1, min(cores, int(quota)))
    return cores, quota


def measure_host(chi: int = 32, L: int = 20, periodic: bool = False, nproc: Optional[int] = None, seed: int = 1234, per: Optional[int] = None) -> dict:
    """the whole CPU share this container has (cpu_budget): one OpenMP team per 64 cores -- scipy's OpenBLAS admits a bounded number of concurrent callers per process (its
    buffer table is sized for 64 threads), so more cores than that are driven by `nproc` pinned PROCESSES, each running one layer of its own copy of the workload at
    the same time (as cpu_layer.measure_host does for the numpy port).  Reported: the aggregate rate over the common wall time of the slowest process."""
    import json
    build()
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    ncores, quota = cpu_budget()
    nproc = nproc or max(1, min(4, ncores // 64))
    per = per or max(1, min(64, ncores // nproc))
    procs = []
    t0 = time.perf_counter()
    for r in range(nproc):
        mask = cpus[r * per:(r + 1) * per]
        code = ("import os, sys, json; os.sched_setaffinity(0, %r); os.environ['OMP_PROC_BIND'] = 'false'; sys.path.insert(0, %r); import cpu_port; "
                "print(json.dumps(cpu_port.measure(%d, %d, nthreads=%d, seed=%d, periodic=%r)))" % (set(mask), HERE, chi, L, per, seed + r, periodic))
        procs.append(subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for pr in procs:
        so, se = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError("cpu_port worker failed: " + se[-2000:])
        outs.append(json.loads(so.strip().splitlines()[-1]))
    slowest = max(x["seconds_per_layer"] for x in outs)
    agg = dict(outs[0])
    agg.update(gates_per_s=sum(x["n_two_site"] for x in outs) / slowest, seconds_per_layer=slowest, threads=per * nproc, processes=nproc, threads_per_process=per,
               per_process_gates_per_s=[round(x["gates_per_s"], 2) for x in outs], algorithmic_gflops=round(sum(x["algorithmic_gflops"] for x in outs), 1),
               wall_seconds_incl_setup=round(time.perf_counter() - t0, 1), periodic=periodic, L=L, cpu_quota=quota, hardware_threads=len(cpus))
    return agg


if __name__ == "__main__":
    import json
    chi = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    print(json.dumps(measure_host(chi, L) if (len(sys.argv) > 3 and sys.argv[3] == "host") else measure(chi, L)))
