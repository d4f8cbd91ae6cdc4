"""CPU oracle: numpy restatement of the reference's BP-gauged gate-application path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package
(`tensornetworkquantumsimulator.jl_amd/`) may import this module; it is used by
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` only,
and there only as the checker / the timed CPU baseline.

What it restates (all paths relative to /root/reference, TensorNetworkQuantumSimulator.jl v0.4.4):
  src/Apply/apply_gates.jl:46-143           apply_gates scheduling rule + apply_gate!
  src/Apply/simple_update.jl:21-77          simple_update (sqrt-env gauge, QR, gate, SVD, ungauge)
  src/utils.jl:18-35,94-108                 pseudo_sqrt_inv_sqrt / safe_eigen (f64 eigen)
  src/MessagePassing/abstractbeliefpropagationcache.jl:99-116,150-259   messages, updated_message, update
  src/MessagePassing/beliefpropagationcache.jl:17-21,51-140            message_diff, defaults, rescale
  src/TensorNetworks/tensornetworkstate.jl:50-75,93-103,141-161        bp_factors, default_message, constructors
  src/truncate.jl:5-38                      BP truncation
  src/expect.jl:59-82                       single-site BP expectation value
  src/Apply/gate_definitions.jl:21-64,248-281   gate matrices (qiskit convention)
  src/graph_ops.jl:6-18                     heavy-hex lattice

The tensor algebra the reference delegates to ITensors.jl 0.9 / NDTensors (NOT under
/root/reference; Project.toml:27, no Manifest so exact version unpinned) is restated
from its published behaviour: `qr` (thin Householder QR), `factorize_svd(...; ortho="none")`
(SVD, L = U sqrt(S), R = sqrt(S) V^dagger), and NDTensors `truncate!` (relative cutoff on S^2,
see `truncate_spectrum` below).

PINNING STATUS.  The reference ships no golden numbers and Julia is absent from the build
container, so no reference-generated vectors exist.  The oracle is pinned against every
known-answer / invariant test the reference holds for this path (tests/test_oracle_pins.py):
  test/test_apply.jl:17-20,50-53   norm_sqr == 1 after un-normalised unitary circuits; bond cap
  test/test_beliefpropagation.jl:28-29,48-54   BP exact on trees; 1-site RDM bp == exact
  test/test_constructors.jl:69-74  GHZ bond entropy == log 2
  test/test_expect.jl:19-28        <Z> bp == exact on a line
  src/Apply/simple_update.jl:4     "exact if no truncation is performed" -> state-vector check
  examples/hexagonal_heisenbergmodel_thermalstate.jl:36   4th-order high-temperature series of the Heisenberg free energy
                                   (imaginary-time gates on d = 4 sites, freenergy + rescale!), reproduced to the next series order
  test/test_truncate.jl:29-33, src/symmetric_gauge.jl   truncate / symmetric-gauge invariants
and against an independent dense state-vector simulator (oracle/statevector.py).
Numeric parity with the Julia package itself is therefore "pinned by invariants only".

Conventions:
  * site tensor axes: (s, leg to nbr_0, leg to nbr_1, ...) with neighbours in ascending vertex position
  * message m[(u, v)] is chi x chi with axes (ket b, bra b'); default = identity
  * norm network: sum psi[b] conj(psi[b']) m[b, b']
"""
from __future__ import annotations

import itertools
import math
from typing import Dict, Hashable, Iterable, List, Optional, Sequence, Tuple

import numpy as np

Vertex = Hashable
DEdge = Tuple[Vertex, Vertex]


# --------------------------------------------------------------------------------------
# graphs
# --------------------------------------------------------------------------------------
class Graph:
    """Minimal named graph: ordered vertices, undirected edges, canonical neighbour order."""

    def __init__(self, vertices: Sequence[Vertex], edges: Iterable[Tuple[Vertex, Vertex]]):
        self.vertices: List[Vertex] = list(vertices)
        self.pos: Dict[Vertex, int] = {v: i for i, v in enumerate(self.vertices)}
        seen = set()
        self.edges: List[Tuple[Vertex, Vertex]] = []
        for (u, v) in edges:
            if u == v:
                raise ValueError("self loop")
            a, b = (u, v) if self.pos[u] < self.pos[v] else (v, u)
            if (a, b) not in seen:
                seen.add((a, b))
                self.edges.append((a, b))
        self.nbrs: Dict[Vertex, List[Vertex]] = {v: [] for v in self.vertices}
        for (a, b) in self.edges:
            self.nbrs[a].append(b)
            self.nbrs[b].append(a)
        for v in self.vertices:
            self.nbrs[v].sort(key=lambda w: self.pos[w])
        self._eset = seen

    def has_edge(self, u, v) -> bool:
        return (u, v) in self._eset or (v, u) in self._eset

    def degree(self, v) -> int:
        return len(self.nbrs[v])

    def leg(self, v, w) -> int:
        """axis (1-based after the site axis) of v's tensor that points at neighbour w"""
        return 1 + self.nbrs[v].index(w)

    def directed_edges(self) -> List[DEdge]:
        out = []
        for (a, b) in self.edges:
            out.append((a, b))
            out.append((b, a))
        return out

    def is_tree(self) -> bool:
        # connected forest == tree; BP defaults only need "no cycles"
        parent = {v: v for v in self.vertices}

        def find(x):
            while parent[x] != x:
                parent[x] = parent[parent[x]]
                x = parent[x]
            return x

        for (a, b) in self.edges:
            ra, rb = find(a), find(b)
            if ra == rb:
                return False
            parent[ra] = rb
        return True


def named_grid(dims: Sequence[int], periodic: bool = False) -> Graph:
    """NamedGraphs.named_grid: vertices are 1-based tuples, first coordinate fastest."""
    dims = tuple(int(x) for x in dims)
    rng = [range(1, n + 1) for n in dims]
    verts = [tuple(reversed(t)) for t in itertools.product(*reversed(rng))]
    if len(dims) == 1:
        verts = [(v[0],) for v in verts]
    edges = []
    for v in verts:
        for ax, n in enumerate(dims):
            if v[ax] < n:
                w = list(v); w[ax] += 1
                edges.append((v, tuple(w)))
            elif periodic and n > 2:
                w = list(v); w[ax] = 1
                edges.append((v, tuple(w)))
    return Graph(verts, edges)


def named_hexagonal_lattice_graph(nx: int, ny: int) -> Graph:
    """Honeycomb lattice with nx x ny hexagons [upstream NamedGraphs/Graphs, recalled: same construction
    as the classic brick-wall generator].  Only the topology matters to the oracle."""
    m, n = nx, ny
    M, N = 2 * m + 2, n
    removed = {(0, M - 1), (N, (M - 1) * (N % 2))}
    nodes = [(i, j) for i in range(N + 1) for j in range(M) if (i, j) not in removed]
    ns = set(nodes)
    edges = []
    for i in range(N + 1):
        for j in range(M - 1):
            if (i, j) in ns and (i, j + 1) in ns:
                edges.append(((i, j), (i, j + 1)))
    for i in range(N):
        for j in range(M):
            if i % 2 == j % 2 and (i, j) in ns and (i + 1, j) in ns:
                edges.append(((i, j), (i + 1, j)))
    ren = {v: (v[0] + 1, v[1] + 1) for v in nodes}
    return Graph([ren[v] for v in nodes], [(ren[a], ren[b]) for (a, b) in edges])


def heavy_hexagonal_lattice(nx: int, ny: int) -> Graph:
    """src/graph_ops.jl:6-18: decorate every edge of the hexagonal lattice with a vertex."""
    g = named_hexagonal_lattice_graph(nx, ny)
    ren = {v: (2 * v[0] - 1, 2 * v[1] - 1) for v in g.vertices}
    verts = [ren[v] for v in g.vertices]
    edges = []
    for (a, b) in g.edges:
        a2, b2 = ren[a], ren[b]
        mid = ((a2[0] + b2[0]) / 2, (a2[1] + b2[1]) / 2)
        verts.append(mid)
        edges.append((a2, mid))
        edges.append((mid, b2))
    return Graph(verts, edges)


def comb_tree(dims: Tuple[int, int]) -> Graph:
    """NamedGraphs named_comb_tree((nx, ny)): a backbone of nx sites with a tooth of ny sites each."""
    nx, ny = dims
    verts = [(i, j) for j in range(1, ny + 1) for i in range(1, nx + 1)]
    edges = [((i, 1), (i + 1, 1)) for i in range(1, nx)]
    edges += [((i, j), (i, j + 1)) for i in range(1, nx + 1) for j in range(1, ny)]
    return Graph(verts, edges)


def edge_color(g: Graph) -> List[List[Tuple[Vertex, Vertex]]]:
    """Proper edge colouring (test-side stand-in for SimpleGraphAlgorithms.edge_color, truncate.jl:20).
    Any proper colouring is valid input to the reference algorithm; greedy by edge order."""
    used: Dict[Vertex, set] = {v: set() for v in g.vertices}
    groups: List[List[Tuple[Vertex, Vertex]]] = []
    for (a, b) in g.edges:
        c = 0
        while c in used[a] or c in used[b]:
            c += 1
        used[a].add(c); used[b].add(c)
        while len(groups) <= c:
            groups.append([])
        groups[c].append((a, b))
    return groups


def forest_cover_edge_sequence(g: Graph) -> List[DEdge]:
    """[upstream NamedGraphs.GraphsExtensions, recalled] default BP edge order
    (beliefpropagationcache.jl:28): for every forest of a greedy spanning-forest cover,
    post-order DFS edges towards the root, then the reversed edges in reversed order."""
    remaining = set(g.edges)
    seq: List[DEdge] = []
    while remaining:
        adj: Dict[Vertex, List[Vertex]] = {v: [] for v in g.vertices}
        for (a, b) in g.edges:
            if (a, b) in remaining:
                adj[a].append(b); adj[b].append(a)
        visited = set()
        used = set()
        for root in g.vertices:
            if root in visited or not adj[root]:
                continue
            # BFS spanning tree of this component
            tree: Dict[Vertex, List[Vertex]] = {}
            visited.add(root)
            queue = [root]
            while queue:
                x = queue.pop(0)
                tree.setdefault(x, [])
                for y in adj[x]:
                    if y not in visited:
                        visited.add(y)
                        tree[x].append(y)
                        queue.append(y)
                        e = (x, y) if (x, y) in remaining else (y, x)
                        used.add(e)
            post: List[DEdge] = []

            def dfs(x):
                for y in tree.get(x, []):
                    dfs(y)
                    post.append((y, x))

            import sys
            sys.setrecursionlimit(max(10000, sys.getrecursionlimit()))
            dfs(root)
            seq.extend(post)
            seq.extend([(b, a) for (a, b) in reversed(post)])
        remaining -= used
    return seq


# --------------------------------------------------------------------------------------
# gates  (gate_definitions.jl:21-64, 248-281; qiskit convention, docs/src/gates.md)
# --------------------------------------------------------------------------------------
_X = np.array([[0, 1], [1, 0]], dtype=complex)
_Y = np.array([[0, -1j], [1j, 0]], dtype=complex)
_Z = np.array([[1, 0], [0, -1]], dtype=complex)
_I = np.eye(2, dtype=complex)


def _expm_herm(h: np.ndarray, t: float) -> np.ndarray:
    w, v = np.linalg.eigh(h)
    return (v * np.exp(-1j * t * w)) @ v.conj().T


def gate_matrix(name: str, *params) -> np.ndarray:
    """d^k x d^k matrix; first listed vertex = most significant qubit."""
    n = name
    if n == "X": return _X.copy()
    if n == "Y": return _Y.copy()
    if n == "Z": return _Z.copy()
    if n == "I": return _I.copy()
    if n == "H": return np.array([[1, 1], [1, -1]], dtype=complex) / math.sqrt(2)
    if n == "Rx":
        (t,) = params; c, s = math.cos(t / 2), math.sin(t / 2)
        return np.array([[c, -1j * s], [-1j * s, c]])
    if n == "Ry":
        (t,) = params; c, s = math.cos(t / 2), math.sin(t / 2)
        return np.array([[c, -s], [s, c]], dtype=complex)
    if n == "Rz":
        (t,) = params
        return np.diag([np.exp(-0.5j * t), np.exp(0.5j * t)])
    if n == "P":
        (p,) = params
        return np.diag([1.0, np.exp(1j * p)])
    if n in ("CNOT", "CX"):
        return np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]], dtype=complex)
    if n == "CY":
        return np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, -1j], [0, 0, 1j, 0]], dtype=complex)
    if n == "CZ":
        return np.diag([1, 1, 1, -1]).astype(complex)
    if n == "SWAP":
        return np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=complex)
    if n == "iSWAP":
        return np.array([[1, 0, 0, 0], [0, 0, 1j, 0], [0, 1j, 0, 0], [0, 0, 0, 1]], dtype=complex)
    if n in ("Rxx", "Ryy", "Rzz"):
        (t,) = params
        p = {"Rxx": _X, "Ryy": _Y, "Rzz": _Z}[n]
        return _expm_herm(np.kron(p, p), t / 2)        # exp(-i theta/2 PP)  (:46-51)
    if n == "CPHASE":
        (p,) = params
        return np.diag([1, 1, 1, np.exp(1j * p)])
    if n in ("CRx", "CRy", "CRz"):
        (t,) = params
        u = gate_matrix("R" + n[2], t)
        m = np.eye(4, dtype=complex); m[2:, 2:] = u
        return m
    if n == "Rxxyy":
        (t,) = params
        return _expm_herm(0.5 * (np.kron(_X, _X) + np.kron(_Y, _Y)), t)       # :263-266
    if n == "Rxxyyzz":
        (t,) = params
        return _expm_herm(0.5 * (np.kron(_X, _X) + np.kron(_Y, _Y) + np.kron(_Z, _Z)), t)  # :276-279
    if n == "xx_plus_yy":
        t, b = params
        c, s = math.cos(t / 2), math.sin(t / 2)
        return np.array([[1, 0, 0, 0],
                         [0, c, -1j * s * np.exp(-1j * b), 0],
                         [0, -1j * s * np.exp(1j * b), c, 0],
                         [0, 0, 0, 1]])                                             # :253-259
    raise ValueError(f'Unknown gate "{name}".')


# --------------------------------------------------------------------------------------
# tensor network state + BP cache
# --------------------------------------------------------------------------------------
class TensorNetworkState:
    """tensornetworkstate.jl:12-15 -- graph + one tensor per vertex, axes (s, nbr legs...)."""

    def __init__(self, g: Graph, tensors: Dict[Vertex, np.ndarray]):
        self.g = g
        self.tensors = dict(tensors)
        for v in g.vertices:
            t = self.tensors[v]
            assert t.ndim == 1 + g.degree(v), (v, t.shape)
        for (a, b) in g.edges:
            assert self.tensors[a].shape[g.leg(a, b)] == self.tensors[b].shape[g.leg(b, a)]

    @property
    def dtype(self):
        return self.tensors[self.g.vertices[0]].dtype

    def copy(self):
        return TensorNetworkState(self.g, dict(self.tensors))

    def bond_dim(self, u, v) -> int:
        return self.tensors[u].shape[self.g.leg(u, v)]

    def maxvirtualdim(self) -> int:     # abstracttensornetwork.jl:27-29
        return max([self.bond_dim(a, b) for (a, b) in self.g.edges], default=1)


def product_state(dtype, f, g: Graph, d: int = 2) -> TensorNetworkState:
    """tensornetworkstate.jl:141-161; "↑"/"Up"/"0" = (1,0), "↓"/"Dn"/"1" = (0,1)."""
    tensors = {}
    for v in g.vertices:
        s = f(v)
        if isinstance(s, str):
            vec = np.zeros(d, dtype=dtype)
            if s in ("↑", "Up", "0", "Z+"):
                vec[0] = 1
            elif s in ("↓", "Dn", "1", "Z-"):
                vec[1] = 1
            elif s in ("+", "X+"):
                vec[:2] = 1 / math.sqrt(2)
            elif s in ("-", "X-"):
                vec[0] = 1 / math.sqrt(2); vec[1] = -1 / math.sqrt(2)
            else:
                raise ValueError(s)
        else:
            vec = np.asarray(s, dtype=dtype)
        tensors[v] = vec.reshape((len(vec),) + (1,) * g.degree(v))
    return TensorNetworkState(g, tensors)


def random_state(dtype, g: Graph, chi: int, d: int = 2, seed: int = 1234) -> TensorNetworkState:
    """random_tensornetworkstate (tensornetworkstate.jl:93-103): iid normal entries, all bonds chi."""
    rng = np.random.default_rng(seed)
    tensors = {}
    for v in g.vertices:
        shp = (d,) + (chi,) * g.degree(v)
        t = rng.standard_normal(shp)
        if np.issubdtype(np.dtype(dtype), np.complexfloating):
            t = (t + 1j * rng.standard_normal(shp)) / math.sqrt(2)
        tensors[v] = t.astype(dtype)
    return TensorNetworkState(g, tensors)


def default_tolerance(dtype) -> Optional[float]:
    """beliefpropagationcache.jl:104-108"""
    dt = np.dtype(dtype)
    if dt in (np.dtype(np.float32), np.dtype(np.complex64)):
        return 1.0e-5
    if dt in (np.dtype(np.float64), np.dtype(np.complex128)):
        return 1.0e-8
    return None


class BeliefPropagationCache:
    """beliefpropagationcache.jl:9-15 -- {network, messages, edge_sequence}."""

    def __init__(self, tns: TensorNetworkState, messages: Optional[Dict[DEdge, np.ndarray]] = None,
                 edge_sequence: Optional[List[DEdge]] = None):
        self.tns = tns
        self.messages: Dict[DEdge, np.ndarray] = dict(messages or {})
        self.edge_sequence = list(edge_sequence) if edge_sequence is not None else forest_cover_edge_sequence(tns.g)

    @property
    def g(self) -> Graph:
        return self.tns.g

    def copy(self):                                   # :35-37 (shallow)
        return BeliefPropagationCache(self.tns.copy(), dict(self.messages), list(self.edge_sequence))

    def message(self, e: DEdge) -> np.ndarray:        # abstract...:99-102, tensornetworkstate.jl:72-75
        m = self.messages.get(e)
        if m is None:
            chi = self.tns.bond_dim(*e)
            m = np.eye(chi, dtype=self.tns.dtype)
        return m

    def default_bp_update_kwargs(self):               # beliefpropagationcache.jl:110-117
        if self.g.is_tree():
            return dict(maxiter=1, tolerance=None)
        return dict(maxiter=25, tolerance=default_tolerance(self.tns.dtype))


def incoming_edges(g: Graph, verts: Sequence[Vertex], ignore: Sequence[DEdge] = ()) -> List[DEdge]:
    """boundary_edges(...; dir=:in) (abstract...:150-156): edges k->v with v in verts, k outside."""
    vs = set(verts)
    out = []
    for v in verts:
        for k in g.nbrs[v]:
            if k not in vs and (k, v) not in ignore:
                out.append((k, v))
    return out


_BIG = 1 << 22          # elements from which the chunked, threaded contractions below replace the plain tensordot calls
_POOL = None


def _pool():
    """thread pool for the large-tensor paths (numpy releases the GIL inside matmul / tensordot): at the per-site shapes of the
    BASELINE configurations (2 * 16^6 or 2 * 64^4 elements per tensor) the plain tensordot + moveaxis formulation spends its time
    in single-threaded transposition copies -- minutes per BP sweep.  Same arithmetic, same index conventions, only the loop
    over the untouched indices is split into chunks."""
    global _POOL
    if _POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=max(1, min(32, (os.cpu_count() or 2) // 2)))
    return _POOL


def _chunks(n: int, parts: int):
    parts = max(1, min(parts, n))
    step = (n + parts - 1) // parts
    return [(a, min(n, a + step)) for a in range(0, n, step)]


def _qr_thin(a: np.ndarray):
    """thin Householder QR.  Tall matrices (>= 8192 rows) go through the communication-avoiding arrangement of the same factorisation (TSQR):
    row blocks of 4096 are factorised on their own (LAPACK geqrf/ungqr on a cache-resident block), the stacked R factors once more, and
    Q = blockdiag(Q_i) [Q'_1; ...; Q'_p].  A 65536 x 64 LAPACK geqrf streams the 33 MB matrix through memory once per reflector (level-2
    panel): 1.1 s on one core, 4.3 s with 64 of them running side by side (profiles/cpu_prim_bench.py).  Q R is the same product either
    way; R's diagonal phases may differ from LAPACK's, which simple_update is invariant to (Q D^* D R)."""
    mrows, n = a.shape
    blk = _QR_BLOCK
    if mrows < 4 * blk or mrows < 4 * n or mrows % blk != 0:
        return np.linalg.qr(a, mode="reduced")
    nb = mrows // blk
    a3 = a.reshape(nb, blk, n)
    qs, rs = np.linalg.qr(a3, mode="reduced")                       # gufunc: one LAPACK call per block
    q2, r = _qr_thin(rs.reshape(nb * n, n))                         # the stacked R factors, again in blocks while they are tall
    q = np.matmul(qs, q2.reshape(nb, n, n)).reshape(mrows, n)
    return q, r


_QR_BLOCK = 4096        # rows per block: 1.3 s per 65536 x 64 factorisation with 64 concurrent callers (2048: 2.4 s, 8192: 2.8 s, one LAPACK call: 4.3 s)


def _absorb(t: np.ndarray, axis: int, m: np.ndarray) -> np.ndarray:
    """t[.. l ..] m[l, l'] -> axis replaced by l' (kept in place)."""
    if t.size < _BIG:
        r = np.tensordot(t, m, axes=([axis], [0]))
        return np.moveaxis(r, -1, axis)
    t = np.ascontiguousarray(t)
    shape = t.shape
    k, kn = shape[axis], m.shape[1]
    pre = int(np.prod(shape[:axis], dtype=np.int64)); post = int(np.prod(shape[axis + 1:], dtype=np.int64))
    dt = np.result_type(t.dtype, m.dtype)
    pool = _pool(); nw = pool._max_workers
    if post == 1:                                        # last axis: one tall GEMM, split by rows
        t2 = t.reshape(pre, k); out = np.empty((pre, kn), dtype=dt); mm = np.ascontiguousarray(m.astype(dt))
        def job(ab):
            out[ab[0]:ab[1]] = t2[ab[0]:ab[1]] @ mm
        list(pool.map(job, _chunks(pre, 4 * nw)))
        return out.reshape(shape[:axis] + (kn,))
    t3 = t.reshape(pre, k, post); out = np.empty((pre, kn, post), dtype=dt); mt = np.ascontiguousarray(m.T.astype(dt))
    if pre >= 4 * nw and post <= 64:                     # many small slabs: one tall GEMM per block of slabs through a cache-sized transpose
        # (batched matmul would issue one k x k by k x post BLAS call per slab -- 2048 calls of 260 kflop for the third bond leg at chi = 32;
        #  with 64 threads doing that at once OpenBLAS spends its time in the per-call buffer lock: measured 38 GFLOP/s for the whole box)
        mm = np.ascontiguousarray(m.astype(dt))
        def job(ab):
            step = max(1, 65536 // (k * post))           # <= 512 KiB of complex64 per block
            for a in range(ab[0], ab[1], step):
                b = min(ab[1], a + step)
                x = np.ascontiguousarray(t3[a:b].transpose(0, 2, 1)).reshape(-1, k)       # ((b - a) post) x k
                out[a:b] = (x @ mm).reshape(b - a, post, kn).transpose(0, 2, 1)
        list(pool.map(job, _chunks(pre, 4 * nw)))
    elif pre >= 4 * nw:                                  # out[p] = m^T t3[p], split over p
        def job(ab):
            out[ab[0]:ab[1]] = np.matmul(mt, t3[ab[0]:ab[1]])
        list(pool.map(job, _chunks(pre, 4 * nw)))
    else:                                                # few, wide slabs: 2-D products on column ranges of each slab
        # Column blocks of 8192 are gathered into a contiguous scratch first: the rows of a slab lie `post` elements apart (a power of
        # two, 256 KiB at chi = 32), so a strided B operand puts all k rows of a column into the same cache sets -- measured on 8 threads
        # 130 -> 13 ms for the first bond leg of a chi = 32 site tensor, and worse with 64 threads sharing an L3.
        def job(pab):
            q, a, b = pab
            nbk = 8192
            buf = np.empty((k, min(nbk, b - a)), dtype=dt)
            for c in range(a, b, nbk):
                d = min(b, c + nbk)
                np.copyto(buf[:, :d - c], t3[q, :, c:d])
                np.matmul(mt, buf[:, :d - c], out=out[q, :, c:d])
        list(pool.map(job, [(q, a, b) for q in range(pre) for (a, b) in _chunks(post, max(1, 4 * nw // pre))]))
    return out.reshape(shape[:axis] + (kn,) + shape[axis + 1:])


def _gram(t: np.ndarray, y: np.ndarray, ax: int) -> np.ndarray:
    """m[b, b'] = sum over every other index of t[.. b ..] conj(y[.. b' ..]) (both in the same index order)"""
    if t.size < _BIG:
        other = [i for i in range(t.ndim) if i != ax]
        return np.tensordot(t, y.conj(), axes=(other, other))
    t = np.ascontiguousarray(t); y = np.ascontiguousarray(y)
    shape = t.shape; k = shape[ax]
    pre = int(np.prod(shape[:ax], dtype=np.int64)); post = int(np.prod(shape[ax + 1:], dtype=np.int64))
    pool = _pool(); nw = pool._max_workers
    if post == 1:
        t2, y2 = t.reshape(pre, k), y.reshape(pre, k)
        parts = list(pool.map(lambda ab: t2[ab[0]:ab[1]].T @ y2[ab[0]:ab[1]].conj(), _chunks(pre, 4 * nw)))
        return sum(parts[1:], parts[0])
    t3, y3 = t.reshape(pre, k, post), y.reshape(pre, k, post)
    if pre >= 4 * nw:
        parts = list(pool.map(lambda ab: np.tensordot(t3[ab[0]:ab[1]], y3[ab[0]:ab[1]].conj(), axes=([0, 2], [0, 2])), _chunks(pre, 4 * nw)))
    else:
        parts = list(pool.map(lambda pab: t3[pab[0], :, pab[1]:pab[2]] @ y3[pab[0], :, pab[1]:pab[2]].conj().T,
                              [(q, a, b) for q in range(pre) for (a, b) in _chunks(post, max(1, 4 * nw // pre))]))
    return sum(parts[1:], parts[0])


def updated_message(bpc: BeliefPropagationCache, e: DEdge, normalize: bool = True) -> np.ndarray:
    """abstract...:162-190.  m_{u->v}[b,b'] = sum psi_u[s,b,l..] conj(psi_u[s,b',l'..]) prod m_{k->u}[l_k,l_k']
    then m / sum(m) if the sum is non-zero."""
    u, v = e
    g = bpc.g
    psi = bpc.tns.tensors[u]
    t = psi
    for k in g.nbrs[u]:
        if k == v:
            continue
        t = _absorb(t, g.leg(u, k), bpc.message((k, u)))
    ax = g.leg(u, v)
    m = _gram(t, psi, ax)                                   # [b, b']
    if normalize:
        s = m.sum()
        if s != 0:
            m = m / s
    return m.astype(psi.dtype, copy=False)


def message_diff(a: np.ndarray, b: np.ndarray) -> float:
    """beliefpropagationcache.jl:17-21"""
    na, nb = np.linalg.norm(a), np.linalg.norm(b)
    f = abs(np.vdot(a, b) / (na * nb)) ** 2
    return float(1 - f)


def update(bpc: BeliefPropagationCache, maxiter: Optional[int] = None, tolerance: Optional[float] = None,
           edge_sequence: Optional[List[DEdge]] = None, normalize: bool = True, info: Optional[dict] = None):
    """abstract...:223-259 (Gauss-Seidel over edge_sequence, newest messages used immediately).  Defaults as set_default_kwargs
    (beliefpropagationcache.jl:63-72): maxiter = default_bp_maxiter (25, 1 on trees; :39), tolerance = nothing (:62) -- i.e. a bare
    `update(bpc)` or `update(bpc; maxiter = 10)` never checks convergence; the dtype tolerance only comes with default_bp_update_kwargs
    (:110-117), which apply_gates / truncate use when bp_update_kwargs is omitted."""
    dk = bpc.default_bp_update_kwargs()
    if maxiter is None:
        maxiter = dk["maxiter"]
    seq = edge_sequence if edge_sequence is not None else bpc.edge_sequence
    bpc = bpc.copy()
    niter, avg = maxiter, None
    converged = False
    for i in range(1, maxiter + 1):
        diff = 0.0
        for e in seq:
            prev = bpc.message(e)
            new = updated_message(bpc, e, normalize=normalize)
            bpc.messages[e] = new
            if tolerance is not None:
                diff += message_diff(new, prev)
        if tolerance is not None:
            avg = diff / len(seq)
            if avg <= tolerance:
                converged, niter = True, i
                break
    if info is not None:
        info.update(niter=niter, diff=avg, converged=converged)
    return bpc


# --------------------------------------------------------------------------------------
# simple update
# --------------------------------------------------------------------------------------
def pseudo_sqrt_inv_sqrt(m: np.ndarray, cutoff: float) -> Tuple[np.ndarray, np.ndarray]:
    """utils.jl:18-27 with safe_eigen (:94-108): Hermitian eigen in f64; D and U are cast back to the message precision FIRST
    (`adapt(dtype)(D), adapt(dtype)(U)`, :102,:106), then the cutoff test, the square roots and the products Q D Q^dagger all run in the
    message precision (:20-25).  For f32 messages an eigenvalue within an f32 ulp of the cutoff therefore lands where the reference puts it."""
    dt = m.dtype
    rt = np.float32 if dt in (np.dtype(np.complex64), np.dtype(np.float32)) else np.float64
    w64, q64 = np.linalg.eigh(m.astype(np.complex128), UPLO="U")
    w, q = w64.astype(rt), q64.astype(dt if np.iscomplexobj(m) else (np.complex64 if rt is np.float32 else np.complex128))
    cut = rt(cutoff)
    zero = (w == 0) | (np.abs(w) < cut)
    if np.any(w[~zero] < 0):
        raise ValueError("DomainError: sqrt of negative message eigenvalue (reference assumes PSD messages)")
    ws = np.where(zero, rt(0), np.sqrt(np.where(zero, rt(1), w))).astype(rt)
    wi = np.where(zero, rt(0), rt(1) / np.where(zero, rt(1), ws)).astype(rt)
    msqrt = (q * ws) @ q.conj().T
    minv = (q * wi) @ q.conj().T
    if not np.iscomplexobj(m):
        msqrt, minv = msqrt.real, minv.real
    return msqrt.astype(dt), minv.astype(dt)


def truncate_spectrum(p: np.ndarray, maxdim: Optional[int], cutoff: Optional[float], mindim: int = 1):
    """[upstream NDTensors.truncate!, recalled; SURVEY.md 3.6].  p = S^2 sorted descending.
    returns (n_keep, truncerr).  cutoff is relative to sum(p); cutoff=None behaves as 0 (drops exact zeros)."""
    p = np.array(p, copy=True)
    n = len(p)
    if p[0] <= 0:
        return 1, 0.0
    if n == 1:
        return 1, 0.0
    p[p < 0] = 0
    ptype = p.dtype.type
    truncerr = ptype(0)
    md = n if maxdim is None else int(maxdim)
    while n > md:
        truncerr += p[n - 1]; n -= 1
    scale = p.sum()
    if scale == 0:
        scale = ptype(1)
    c = ptype(0 if cutoff is None else cutoff)
    while n > mindim and truncerr + p[n - 1] <= c * scale:
        truncerr += p[n - 1]; n -= 1
    truncerr = truncerr / scale
    return n, float(truncerr)


def simple_update(gate: np.ndarray, psis: List[np.ndarray], bond_axes: Optional[Tuple[int, int]],
                  envs: Optional[Tuple[List[Tuple[int, np.ndarray]], List[Tuple[int, np.ndarray]]]],
                  maxdim: Optional[int] = None, cutoff: Optional[float] = None, normalize_tensors: bool = True,
                  sqrt_cutoff: Optional[float] = None):
    """simple_update.jl:21-77.
    1-site: psis=[psi], gate d x d.   2-site: psis=[psi1, psi2], bond_axes=(axis in psi1, axis in psi2),
    envs=([(axis, M) for outer legs of psi1], [(axis, M) ... psi2]).
    returns (new tensors, S or None, truncerr)."""
    if len(psis) == 1:
        psi = psis[0]
        new = np.tensordot(gate.astype(psi.dtype), psi, axes=([1], [0]))      # out[s'] = sum_s o[s',s] psi[s]  (:27)
        out = [new]
        s_values, err = None, 0.0
    else:
        psi1, psi2 = psis
        dt = psi1.dtype
        real_eps = np.finfo(np.zeros(1, dtype=dt).real.dtype).eps
        if sqrt_cutoff is None:
            sqrt_cutoff = 10 * real_eps                                          # :32-33
        b1, b2 = bond_axes
        sq1 = [(ax,) + pseudo_sqrt_inv_sqrt(m, sqrt_cutoff) for ax, m in envs[0]]   # :38-39
        sq2 = [(ax,) + pseudo_sqrt_inv_sqrt(m, sqrt_cutoff) for ax, m in envs[1]]
        t1, t2 = psi1, psi2
        for ax, ms, _ in sq1:                                                    # :43
            t1 = _absorb(t1, ax, ms)
        for ax, ms, _ in sq2:                                                    # :44
            t2 = _absorb(t2, ax, ms)

        def qr_split(t, bax):                                                    # :45-48
            outer = [i for i in range(t.ndim) if i not in (0, bax)]
            tm = np.transpose(t, outer + [0, bax])
            oshape = tm.shape[:len(outer)]
            d, chi = t.shape[0], t.shape[bax]
            q, r = _qr_thin(np.ascontiguousarray(tm).reshape(-1, d * chi))
            return q, r.reshape(r.shape[0], d, chi), outer, oshape

        q1, r1, outer1, oshape1 = qr_split(t1, b1)
        q2, r2, outer2, oshape2 = qr_split(t2, b2)
        d1, d2 = psi1.shape[0], psi2.shape[0]
        theta = np.einsum("asb,ctb->asct", r1, r2)                               # R1 * R2 over the bond (:51)
        g4 = gate.astype(dt).reshape(d1, d2, d1, d2)                             # [s1', s2', s1, s2]
        theta = np.einsum("xyst,asct->axcy", g4, theta)                          # ITensors.apply
        rr1, rr2 = r1.shape[0], r2.shape[0]
        mat = theta.reshape(rr1 * d1, rr2 * d2)
        u, s, vh = np.linalg.svd(mat, full_matrices=False)                       # :53-59
        p = (s.astype(s.dtype) ** 2)
        n, err = truncate_spectrum(p, maxdim, cutoff)
        u, s, vh = u[:, :n], s[:n], vh[:n, :]
        sq = np.sqrt(s)
        L = (u * sq).reshape(rr1, d1, n)                                         # ortho = "none"
        R = (sq[:, None] * vh).reshape(n, rr2, d2)
        # :62-63  ungauge Q on its outer legs: Q[l] = sum_l' Q[l'] conj(M^-1/2[l, l'])
        def ungauge(q, outer, oshape, sq):
            qt = q.reshape(oshape + (q.shape[1],))
            for i, ax in enumerate(outer):
                mi = [m for (a, _, m) in sq if a == ax][0]
                qt = _absorb(qt, i, mi.conj().T)                                 # same contraction: sum_l' Q[l'] conj(M^-1/2)[l, l']
            return qt
        q1t = ungauge(q1, outer1, oshape1, sq1)
        q2t = ungauge(q2, outer2, oshape2, sq2)
        n1 = np.tensordot(q1t, L, axes=([-1], [0]))                              # [outer..., s1, u]   (:64)
        n2 = np.tensordot(q2t, R, axes=([-1], [1]))                              # [outer..., u, s2]
        n2 = np.swapaxes(n2, -1, -2)                                             # [outer..., s2, u]

        def restore(nt, outer, bax, ndim):
            # nt axes: outer..., s, u  -> original axis order with u at bax
            cur = list(outer) + [0, bax]
            perm = [cur.index(i) for i in range(ndim)]
            return np.transpose(nt, perm)
        out = [restore(n1, outer1, b1, psi1.ndim), restore(n2, outer2, b2, psi2.ndim)]
        s_values = s
        if normalize_tensors:
            s_values = s_values / np.linalg.norm(s_values)                       # :65-67
    if normalize_tensors:
        out = [t / np.linalg.norm(t) for t in out]                               # :70-74
    dt0 = psis[0].dtype
    out = [np.ascontiguousarray(t.astype(dt0)) for t in out]
    return out, s_values, err


def apply_gate(bpc: BeliefPropagationCache, gate: np.ndarray, verts: Sequence[Vertex],
               maxdim=None, cutoff=None, normalize_tensors=True, sqrt_cutoff=None) -> float:
    """apply_gate! (apply_gates.jl:101-143) -- mutates bpc."""
    g = bpc.g
    nv = len(verts)
    if not (1 <= nv <= 2):
        raise RuntimeError(f"apply_gate!: only one- and two-site gates are supported; received a gate acting on {nv} vertices: {list(verts)}.")
    if nv == 1:
        (v,) = verts
        out, _, err = simple_update(gate, [bpc.tns.tensors[v]], None, None, normalize_tensors=normalize_tensors)
        bpc.tns.tensors[v] = out[0]
        return 0.0
    v1, v2 = verts
    if not g.has_edge(v1, v2):
        raise RuntimeError(f"apply_gate!: cannot apply a two-site gate on the non-adjacent vertices {v1} and {v2}.")
    env1 = [(g.leg(v1, k), bpc.message((k, v1))) for k in g.nbrs[v1] if k != v2]      # :122
    env2 = [(g.leg(v2, k), bpc.message((k, v2))) for k in g.nbrs[v2] if k != v1]
    out, s, err = simple_update(gate, [bpc.tns.tensors[v1], bpc.tns.tensors[v2]],
                                (g.leg(v1, v2), g.leg(v2, v1)), (env1, env2),
                                maxdim=maxdim, cutoff=cutoff, normalize_tensors=normalize_tensors,
                                sqrt_cutoff=sqrt_cutoff)
    dt = bpc.tns.dtype
    md = np.diag(s).astype(dt)                                                    # :126-135
    bpc.messages[(v1, v2)] = md.copy()
    bpc.messages[(v2, v1)] = md.copy()
    bpc.tns.tensors[v1], bpc.tns.tensors[v2] = out                               # :138-140
    return err


def resolve_gate(gate, d: int = 2) -> Tuple[np.ndarray, List[Vertex]]:
    """("Rzz", [v1, v2], theta) or ("Rx", [v], theta) or (matrix, [verts]) -> (matrix, verts)."""
    name, verts = gate[0], gate[1]
    if not isinstance(verts, list):
        verts = [verts]
    if isinstance(name, np.ndarray):
        return name, list(verts)
    params = gate[2:] if len(gate) > 2 else ()
    if len(params) == 1 and isinstance(params[0], (tuple, list)):
        params = tuple(params[0])
    return gate_matrix(name, *params), list(verts)


def apply_gates(circuit: Sequence, bpc: BeliefPropagationCache, apply_kwargs: Optional[dict] = None,
                bp_update_kwargs: Optional[dict] = None, update_cache: bool = True, info: Optional[dict] = None):
    """apply_gates (apply_gates.jl:46-98): returns (new cache, truncation errors)."""
    apply_kwargs = dict(apply_kwargs or {})
    bp_kw = dict(bp_update_kwargs) if bp_update_kwargs is not None else bpc.default_bp_update_kwargs()   # :51
    bpc = bpc.copy()                                                                # :55
    affected = set()
    errs = np.zeros(len(circuit))
    n_updates = 0
    sweeps = []
    for ii, gate in enumerate(circuit):
        mat, verts = resolve_gate(gate)
        need = len(verts) >= 2 and any(v in affected for v in verts)               # :68
        if update_cache and need:
            inf = {}
            bpc = update(bpc, info=inf, **bp_kw)                                    # :76
            n_updates += 1; sweeps.append(inf.get("niter"))
            affected.clear()                                                        # :78
        errs[ii] = apply_gate(bpc, mat, verts, **apply_kwargs)                      # :87
        for v in verts:
            affected.add(v)                                                         # :88-90
    if update_cache:
        inf = {}
        bpc = update(bpc, info=inf, **bp_kw)                                        # :93-95
        n_updates += 1; sweeps.append(inf.get("niter"))
    if info is not None:
        info.update(n_updates=n_updates, sweeps=sweeps)
    return bpc, errs


def truncate(bpc: BeliefPropagationCache, maxdim: int, cutoff=None, normalize_tensors=True,
             edge_groups: Optional[List[List[Tuple[Vertex, Vertex]]]] = None,
             bp_update_kwargs: Optional[dict] = None):
    """truncate.jl:12-38 (edge_color=true branch; colouring supplied by the caller)."""
    bpc = bpc.copy()
    bp_kw = dict(bp_update_kwargs) if bp_update_kwargs is not None else bpc.default_bp_update_kwargs()    # truncate.jl:12
    if edge_groups is None:
        edge_groups = edge_color(bpc.g)
    for eg in edge_groups:
        for (a, b) in eg:
            if bpc.tns.bond_dim(a, b) == 1:                                         # truncatable_edge :5-10
                continue
            d1, d2 = bpc.tns.tensors[a].shape[0], bpc.tns.tensors[b].shape[0]
            apply_gate(bpc, np.eye(d1 * d2, dtype=complex), [a, b], maxdim=maxdim, cutoff=cutoff,
                       normalize_tensors=normalize_tensors)
        bpc = update(bpc, **bp_kw)                                                  # :28
    return bpc


# --------------------------------------------------------------------------------------
# observables / normalisation used as parity probes
# --------------------------------------------------------------------------------------
def rdm_1site(bpc: BeliefPropagationCache, v: Vertex) -> np.ndarray:
    """rho[s, s'] = sum psi[s, l..] conj(psi[s', l'..]) prod m[l_k, l_k']  (un-normalised)."""
    g = bpc.g
    psi = bpc.tns.tensors[v]
    t = psi
    for k in g.nbrs[v]:
        t = _absorb(t, g.leg(v, k), bpc.message((k, v)))
    other = list(range(1, psi.ndim))
    return np.tensordot(t, psi.conj(), axes=(other, other))


def expect_1site(bpc: BeliefPropagationCache, op: np.ndarray, v: Vertex) -> complex:
    """expect(alg"bp", cache, (op, [v])) (expect.jl:59-82): numer / denom with op[s', s]."""
    rho = rdm_1site(bpc, v)
    numer = np.einsum("ts,st->", op.astype(rho.dtype), rho)
    return complex(numer / np.trace(rho))


def steiner_vertices(g: Graph, terminals: Sequence[Vertex]) -> List[Vertex]:
    """vertices(steiner_tree(network(cache), obs_vs)) (expect.jl:67; Graphs.jl steiner_tree, [upstream, recalled]: Kou-Markowsky-Berman with
    unit weights).  Dijkstra from every terminal (heap keyed by (distance, discovery count); a vertex is relaxed only by a strictly
    shorter path, so the first parent found stays), minimum spanning tree of the terminals' distance graph, its pairs expanded into
    paths, a spanning tree of the union, non-terminal leaves pruned.  Returned in vertex order."""
    import heapq
    terms = list(dict.fromkeys(terminals))
    if len(terms) == 1:
        return terms
    dist: Dict = {}; parent: Dict = {}
    for t in terms:
        d = {t: 0}; pa = {t: None}; heap = [(0, 0, t)]; count = 1; done = set()
        while heap:
            du, _, u = heapq.heappop(heap)
            if u in done:
                continue
            done.add(u)
            for w in g.nbrs[u]:
                if w not in d or du + 1 < d[w]:
                    d[w] = du + 1; pa[w] = u
                    heapq.heappush(heap, (du + 1, count, w)); count += 1
        dist[t], parent[t] = d, pa
    if any(t not in dist[terms[0]] for t in terms):
        raise ValueError("steiner tree: terminals are not connected")

    def spanning(nodes, weighted):
        root = {v: v for v in nodes}

        def find(x):
            while root[x] != x:
                x = root[x]
            return x
        keep = []
        for (_, a, b) in sorted(weighted, key=lambda t: t[0]):
            ra, rb = find(a), find(b)
            if ra != rb:
                root[ra] = rb; keep.append((a, b))
        return keep

    pairs = [(dist[terms[i]][terms[j]], terms[i], terms[j]) for i in range(len(terms)) for j in range(i + 1, len(terms))]
    used = set()
    for (a, b) in spanning(terms, pairs):
        x = b
        while x != a:
            y = parent[a][x]
            used.add((x, y) if g.pos[x] < g.pos[y] else (y, x)); x = y
    nodes = sorted({v for e in used for v in e}, key=lambda v: g.pos[v])
    tree = spanning(nodes, [(1, a, b) for (a, b) in g.edges if (a, b) in used])
    while True:
        deg: Dict = {}
        for (a, b) in tree:
            deg[a] = deg.get(a, 0) + 1; deg[b] = deg.get(b, 0) + 1
        leaves = {v for v, k in deg.items() if k == 1 and v not in terms}
        if not leaves:
            break
        tree = [(a, b) for (a, b) in tree if a not in leaves and b not in leaves]
    return sorted({v for e in tree for v in e}, key=lambda v: g.pos[v])


def expect(bpc: BeliefPropagationCache, ops: Dict) -> complex:
    """expect(alg"bp", cache, (ops, vertices)) (expect.jl:59-82): the region is the vertex set of the Steiner tree of the support"""
    return expect_region(bpc, ops, steiner_vertices(bpc.g, list(ops.keys())))


def expect_region(bpc: BeliefPropagationCache, ops: Dict, region: Sequence[Vertex]) -> complex:
    """expect(alg"bp", cache, obs) for a multi-site observable (expect.jl:59-82): the norm network of `region` (the Steiner tree
    of the observable's support, :68) with the cache's messages on the boundary edges (:69), operators `ops[v]` (op[s', s])
    inserted on the support and identities elsewhere (:72-77); numer / denom (:79-82).  Dense einsum: small regions only."""
    g = bpc.g
    rset = set(region)

    def contract(with_ops: bool) -> complex:
        operands = []
        ids: Dict = {}

        def idx(key):
            if key not in ids:
                ids[key] = len(ids)
            return ids[key]

        for v in region:
            psi = bpc.tns.tensors[v].astype(np.complex128)
            ket_sub = [idx(("s", v, "k"))]; bra_sub = [idx(("s", v, "b"))]
            for k in g.nbrs[v]:
                e = (v, k) if g.pos[v] < g.pos[k] else (k, v)
                if k in rset:
                    ket_sub.append(idx(("e", e, "k"))); bra_sub.append(idx(("e", e, "b")))
                else:
                    ket_sub.append(idx(("m", v, k, "k"))); bra_sub.append(idx(("m", v, k, "b")))
                    operands += [bpc.message((k, v)).astype(np.complex128), [idx(("m", v, k, "k")), idx(("m", v, k, "b"))]]
            operands += [psi, ket_sub, psi.conj(), bra_sub]
            o = ops.get(v) if with_ops else None
            m = np.eye(psi.shape[0], dtype=np.complex128) if o is None else np.asarray(o, dtype=np.complex128)
            operands += [m, [idx(("s", v, "b")), idx(("s", v, "k"))]]          # op[s', s]: s' contracts with the bra, s with the ket
        return complex(np.einsum(*operands, [], optimize=True))

    return contract(True) / contract(False)


def vertex_scalar(bpc: BeliefPropagationCache, v: Vertex) -> complex:   # abstract...:22-28
    return complex(np.trace(rdm_1site(bpc, v)))


def edge_scalar(bpc: BeliefPropagationCache, e: DEdge) -> complex:     # beliefpropagationcache.jl:47-49
    return complex(np.sum(bpc.message(e) * bpc.message((e[1], e[0]))))


def rescale(bpc: BeliefPropagationCache) -> BeliefPropagationCache:
    """rescale! (abstract...:318-322) = rescale_messages! (bpc.jl:127-140) then rescale_vertices! (:82-101)."""
    bpc = bpc.copy()
    for (a, b) in bpc.g.edges:
        me = bpc.message((a, b)); mer = bpc.message((b, a))
        me = me / np.linalg.norm(me); mer = mer / np.linalg.norm(mer)
        n = np.sum(me * mer)
        if abs(n.imag) == 0:
            sgn = np.sign(n.real)
            me = me * sgn; n = n * sgn
        bpc.messages[(a, b)] = (me / np.sqrt(n)).astype(bpc.tns.dtype)
        bpc.messages[(b, a)] = (mer / np.sqrt(n)).astype(bpc.tns.dtype)
    for v in bpc.g.vertices:
        vn = vertex_scalar(bpc, v)
        s = np.sign(vn.real) if vn.imag == 0 else 1.0
        bpc.tns.tensors[v] = (bpc.tns.tensors[v] * s / np.sqrt(vn)).astype(bpc.tns.dtype)
    return bpc


def symmetric_gauge(bpc: BeliefPropagationCache, regularization: Optional[float] = None) -> BeliefPropagationCache:
    """symmetric_gauge (src/symmetric_gauge.jl:1-62): per edge e = (src, dst), X = message(e), Y = message(reverse(e)):
    psi_src <- psi_src X^{-1/2} U sqrt(S), psi_dst <- psi_dst Y^{-1/2} V sqrt(S) with U S V = svd(X^{1/2} (Y^{1/2})^T), both
    messages of the edge := diag(S).  Eigenvalues are regularised by 10 eps before the roots (:1,:15-16)."""
    bpc = bpc.copy()
    g = bpc.g
    real = np.float32 if bpc.tns.dtype == np.complex64 else np.float64
    reg = 10 * np.finfo(real).eps if regularization is None else regularization
    dt = bpc.tns.dtype
    for (a, b) in g.edges:
        la, lb = g.leg(a, b), g.leg(b, a)
        roots = []
        for m in (bpc.message((a, b)), bpc.message((b, a))):
            w, q = np.linalg.eigh(m.astype(np.complex128), UPLO="U")            # safe_eigen, ishermitian = true (utils.jl:94-108)
            w = w + reg
            if np.any(w < 0):
                raise ValueError("DomainError: sqrt of a negative (regularised) message eigenvalue")
            # ITensors.eigen without explicit index sets takes the PRIMED index as the row index (A_{l' l} U_{l j} = U_{l' j} D_j),
            # i.e. it diagonalises M^T; X_U * f(D) * prime(dag(X_U)) is therefore f(M)^T = conj(f(M)) as a tensor [l, l']
            roots.append((((q * np.sqrt(w)) @ q.conj().T).conj(), ((q / np.sqrt(w)) @ q.conj().T).conj()))
        (rx, irx), (ry, iry) = roots
        ce = (rx @ ry.T).astype(dt)
        u, sv_, vh = np.linalg.svd(ce, full_matrices=False)
        xs = (irx @ u.astype(np.complex128)) * np.sqrt(sv_.astype(np.float64))
        xd = (iry @ vh.T.astype(np.complex128)) * np.sqrt(sv_.astype(np.float64))
        bpc.tns.tensors[a] = _absorb(bpc.tns.tensors[a].astype(np.complex128), la, xs).astype(dt)
        bpc.tns.tensors[b] = _absorb(bpc.tns.tensors[b].astype(np.complex128), lb, xd).astype(dt)
        smat = np.diag(sv_).astype(dt)
        bpc.messages[(a, b)] = smat
        bpc.messages[(b, a)] = smat.copy()
    return bpc


def partitionfunction(bpc: BeliefPropagationCache) -> complex:          # abstract...:289-304
    num = [vertex_scalar(bpc, v) for v in bpc.g.vertices]
    den = [edge_scalar(bpc, e) for e in bpc.g.edges]
    return complex(np.exp(np.sum(np.log(np.array(num, dtype=complex))) - np.sum(np.log(np.array(den, dtype=complex)))))


def bond_entropy(bpc: BeliefPropagationCache, e: DEdge) -> float:
    """von Neumann bond entropy from BP messages (entanglement.jl:73-86, used only to pin the GHZ known answer)."""
    m1, m2 = bpc.message(e).astype(complex), bpc.message((e[1], e[0])).astype(complex)
    w1, q1 = np.linalg.eigh((m1 + m1.conj().T) / 2)
    w1 = np.clip(w1, 0, None)
    s1 = (q1 * np.sqrt(w1)) @ q1.conj().T
    rho = s1 @ m2.T @ s1
    ev = np.linalg.eigvalsh((rho + rho.conj().T) / 2)
    ev = np.clip(ev.real, 0, None)
    ev = ev / ev.sum()
    ev = ev[ev > 1e-300]
    return float(-np.sum(ev * np.log(ev)))
