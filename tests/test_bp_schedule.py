"""Host-side logic of the BP update (no GPU): the library's default sweep order and the dependency levels it is launched in, read through the host-only
entry point tnqs_dbg_default_sequence_graph (include/tnqs_debug.h; csrc/engine_bp.cpp default_sequence / sequence_levels).  The order must be an ordinary
sequential one -- every directed edge exactly once (abstractbeliefpropagationcache.jl:204-218 sweeps `edge_sequence` in order) -- and the levels a valid
schedule of it: a message that reads the NEW value of another one runs in a later level.  On top of that the structure the order is built for: both
messages a site sends into a linear forest / an edge set that closes cycles leave in one level (one pass over the site tensor for two messages)."""
import ctypes as C

import numpy as np
import pytest

import tnqs_amd as tn


def default_schedule(g):
    lib = C.CDLL(tn.LIB_PATH)
    fn = lib.tnqs_dbg_default_sequence_graph
    fn.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
    fn.restype = C.c_int
    idx = {v: i for i, v in enumerate(g.vertices)}
    es = np.array([idx[a] for (a, b) in g.edges], dtype=np.int32); ed = np.array([idx[b] for (a, b) in g.edges], dtype=np.int32)
    cap = 2 * g.ne(); src = (C.c_int * cap)(); dst = (C.c_int * cap)(); lev = (C.c_int * cap)(); n = C.c_int(0)
    rc = fn(g.nv(), g.ne(), es.ctypes.data_as(C.POINTER(C.c_int32)), ed.ctypes.data_as(C.POINTER(C.c_int32)), src, dst, lev, cap, C.byref(n))
    assert rc == 0 and n.value == cap
    return [(g.vertices[src[i]], g.vertices[dst[i]]) for i in range(cap)], [lev[i] for i in range(cap)]


LATTICES = {
    "grid5x5": lambda: tn.named_grid((5, 5)),
    "grid20x20": lambda: tn.named_grid((20, 20)),
    "heavyhex5x5": lambda: tn.heavy_hexagonal_lattice(5, 5),
    "hexagonal3x3": lambda: tn.named_hexagonal_lattice_graph(3, 3),
    "torus4x4": lambda: tn.named_grid((4, 4), periodic=True),
    "torus8x8": lambda: tn.named_grid((8, 8), periodic=True),
    "cubic3p": lambda: tn.named_grid((3, 3, 3), periodic=True),
    "cubic5p": lambda: tn.named_grid((5, 5, 5), periodic=True),
    "cubic4": lambda: tn.named_grid((4, 4, 4)),
    "cubic10p": lambda: tn.named_grid((10, 10, 10), periodic=True),      # BASELINE configs[3] at full size (1000 sites, 3000 edges)
    "grid32x32": lambda: tn.named_grid((32, 32)),                        # BASELINE configs[4]
    "ring7": lambda: tn.named_grid((7,), periodic=True),
    "comb": lambda: tn.named_comb_tree((4, 3)),
}


@pytest.mark.parametrize("name", sorted(LATTICES))
def test_default_order_is_a_sequential_order_with_a_valid_level_schedule(name):
    g = LATTICES[name]()
    seq, lev = default_schedule(g)
    assert sorted(seq) == sorted([(a, b) for (a, b) in g.edges] + [(b, a) for (a, b) in g.edges])       # every message once
    pos = {m: t for t, m in enumerate(seq)}
    for t, (s, d) in enumerate(seq):
        for k in g.neighbors(s):
            if k != d and pos[(k, s)] < t:                    # reads the new value of (k -> s): must run in a later level
                assert lev[pos[(k, s)]] < lev[t], (name, (k, s), (s, d))
    assert min(lev) == 0 and sorted(set(lev)) == list(range(max(lev) + 1))


@pytest.mark.parametrize("name,levels,passes", [
    ("grid5x5", 2, None), ("grid20x20", 2, None),          # rows and columns: two levels per sweep
    ("heavyhex5x5", 2, None),
    ("torus4x4", 4, 2 * 16), ("torus8x8", 4, 2 * 64),      # two sets of cycles: every site sends two messages per set in ONE level
    ("cubic3p", 6, 3 * 27), ("cubic5p", 6, 3 * 125), ("cubic10p", 6, 3 * 1000),       # three sets, two levels each (all sites of a ring but one, then that one)
    ("grid32x32", 2, None),
    ("ring7", 2, 7)])
def test_two_messages_per_site_and_level_on_lattices(name, levels, passes):
    g = LATTICES[name]()
    seq, lev = default_schedule(g)
    assert max(lev) + 1 == levels, (name, max(lev) + 1)
    per = {}
    for (s, _d), l in zip(seq, lev):
        per[(s, l)] = per.get((s, l), 0) + 1
    assert max(per.values()) <= 2                            # never more than the two messages of one set: the plane kernels' both-messages pass
    if passes is not None:
        assert len(per) == passes, (name, len(per))          # = (site, level) passes over the site tensors per sweep


def test_tree_order_is_exact_in_one_sweep():
    """on a tree the default is the forest-cover order (post-order edges towards the root, then their reverses): every message is listed AFTER the messages it
    depends on, which is what makes the reference's tree defaults -- one sweep, no tolerance (beliefpropagationcache.jl:39,110-113) -- exact"""
    g = LATTICES["comb"]()
    seq, _lev = default_schedule(g)
    pos = {m: t for t, m in enumerate(seq)}
    for (s, d) in seq:
        for k in g.neighbors(s):
            if k != d:
                assert pos[(k, s)] < pos[(s, d)]
