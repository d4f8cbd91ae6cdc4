"""Host-side logic (no GPU): graphs, edge colouring, gate registry -- checked against the oracle's independent
restatement where both exist."""
import numpy as np
import pytest

import tnqs_amd as tn
import tnqs_oracle as o


@pytest.mark.parametrize("g,k", [(tn.named_grid((20, 20)), 4), (tn.heavy_hexagonal_lattice(5, 5), 3),
                                 (tn.named_grid((10, 10, 10), periodic=True), 6), (tn.named_grid((5, 5)), 4),
                                 (tn.named_grid((3, 3, 3), periodic=True), None)])
def test_edge_color_is_proper(g, k):
    groups = tn.edge_color(g, k)
    assert sum(len(x) for x in groups) == g.ne()
    for grp in groups:
        vs = [v for e in grp for v in e]
        assert len(vs) == len(set(vs))
        assert all(g.has_edge(a, b) for (a, b) in grp)


def test_lattice_sizes_match_baseline_configs():
    g = tn.named_grid((20, 20)); assert (g.nv(), g.ne()) == (400, 760)
    g = tn.heavy_hexagonal_lattice(5, 5); assert (g.nv(), g.ne()) == (164, 188)
    assert sorted(set(g.degree(v) for v in g.vertices)) == [2, 3]
    g = tn.named_grid((10, 10, 10), periodic=True); assert (g.nv(), g.ne()) == (1000, 3000)
    g = tn.named_grid((32, 32)); assert (g.nv(), g.ne()) == (1024, 1984)
    g = tn.named_grid((5, 5)); assert (g.nv(), g.ne()) == (25, 40)
    og = o.named_grid((5, 5))
    assert og.vertices == g.vertices and set(map(frozenset, og.edges)) == set(map(frozenset, g.edges))


def test_forest_cover_sequence_covers_every_directed_edge_once():
    for g in (tn.named_grid((4, 4)), tn.heavy_hexagonal_lattice(1, 1), tn.named_comb_tree((3, 3))):
        seq = tn.forest_cover_edge_sequence(g)
        assert len(seq) == 2 * g.ne() and len(set(seq)) == len(seq)
        assert all(g.has_edge(a, b) for (a, b) in seq)


def test_gate_matrices_match_oracle_and_are_unitary():
    cases = [("Rx", (0.3,)), ("Ry", (0.7,)), ("Rz", (1.1,)), ("P", (0.4,)), ("H", ()), ("CNOT", ()), ("CY", ()),
             ("CZ", ()), ("SWAP", ()), ("iSWAP", ()), ("Rxx", (0.5,)), ("Ryy", (0.5,)), ("Rzz", (0.9,)),
             ("CRx", (0.2,)), ("CRz", (0.2,)), ("CPHASE", (-0.3,)), ("Rxxyy", (0.3,)), ("Rxxyyzz", (0.6,)),
             ("xx_plus_yy", (0.3, 0.8))]
    for n, p in cases:
        m = tn.gate_matrix(n, *p)
        assert np.allclose(m, o.gate_matrix(n, *p)), n
        assert np.allclose(m @ m.conj().T, np.eye(m.shape[0])), n
    assert np.allclose(tn.gate_matrix("√SWAP") @ tn.gate_matrix("√SWAP"), tn.gate_matrix("SWAP"))
    assert np.allclose(tn.gate_matrix("√iSWAP") @ tn.gate_matrix("√iSWAP"), tn.gate_matrix("iSWAP"))


def test_gate_registry_behaviour():
    """test/test_apply.jl:56-106: custom registration, aliases, locked built-ins, unknown-gate suggestions"""
    tn.register_gate("MyZRot", lambda t: tn.gate_matrix("Rz", t), nparams=1)
    assert np.allclose(tn.gate_matrix("MyZRot", 0.37), tn.gate_matrix("Rz", 0.37))
    tn.register_alias("myz", "MyZRot")
    assert np.allclose(tn.gate_matrix("myz", 0.1), tn.gate_matrix("Rz", 0.1))
    tn.unregister_gate("MyZRot")
    with pytest.raises(ValueError):
        tn.gate_matrix("myz", 0.1)
    with pytest.raises(ValueError, match="built-in"):
        tn.register_gate("Rz", lambda t: np.eye(2), nparams=1)
    with pytest.raises(ValueError, match="built-in"):
        tn.unregister_gate("CNOT")
    with pytest.raises(ValueError, match="Did you mean"):
        tn.gate_matrix("Rzx", 0.1)
    assert np.allclose(tn.gate_matrix("rzz", 0.3), tn.gate_matrix("Rzz", 0.3))
    assert np.allclose(tn.gate_matrix("cp", 0.3), tn.gate_matrix("CPHASE", 0.3))
    assert np.allclose(tn.gate_matrix("XZ"), np.kron(tn.gate_matrix("X"), tn.gate_matrix("Z")))


def test_gate_matrix_cache_follows_the_registry():
    """gate matrices are memoised per (name, params) (a layer repeats two distinct gates 1160 times); the cache must never serve a
    matrix of a gate that was re-registered, unregistered, aliased away or replaced directly in the public GATES dict"""
    a = tn.gate_matrix("Rzz", 0.3)
    assert tn.gate_matrix("Rzz", 0.3) is a and not a.flags.writeable
    assert tn.gate_matrix("Rzz", 0.31) is not a
    assert np.allclose(tn.gate_matrix("Rzz", 0.3), o.gate_matrix("Rzz", 0.3))
    assert tn.gate_matrix("Rzz", (0.3,)) is a                                    # tuple-wrapped parameter (:137-139)
    tn.register_gate("MyPhase", lambda t: np.diag([1.0, np.exp(1j * t)]), nparams=1)
    try:
        m1 = tn.gate_matrix("MyPhase", 0.5)
        assert np.allclose(m1, np.diag([1.0, np.exp(0.5j)]))
        tn.unregister_gate("MyPhase")
        with pytest.raises(ValueError):
            tn.gate_matrix("MyPhase", 0.5)
        tn.register_gate("MyPhase", lambda t: np.diag([np.exp(-1j * t), 1.0]), nparams=1)
        assert np.allclose(tn.gate_matrix("MyPhase", 0.5), np.diag([np.exp(-0.5j), 1.0]))
        spec = tn.GATES["MyPhase"]
        tn.GATES["MyPhase"] = type(spec)(lambda t: np.eye(2) * t, 1)                 # replaced behind the registry's back
        assert np.allclose(tn.gate_matrix("MyPhase", 0.5), 0.5 * np.eye(2))
        tn.register_alias("Ph", "MyPhase")
        assert np.allclose(tn.gate_matrix("Ph", 2.0), 2.0 * np.eye(2))
    finally:
        tn.unregister_gate("MyPhase")
    with pytest.raises(ValueError):
        tn.gate_matrix("Ph", 2.0)
    assert np.allclose(tn.gate_matrix("XZ"), np.kron(o.gate_matrix("X"), o.gate_matrix("Z")))      # Pauli-string sugar is cached too
    assert tn.gate_matrix("XZ") is tn.gate_matrix("XZ")
    with pytest.raises(ValueError):
        tn.gate_matrix("Rzz")                                                    # wrong parameter count still raises


def test_bp_kwargs_follow_the_reference_defaults():
    """`update(bpc; maxiter = 10)` has NO tolerance in the reference (`default_tolerance(::Algorithm"bp") = nothing`,
    beliefpropagationcache.jl:62-67); only `default_bp_update_kwargs` -- what apply_gates / truncate / normalize use when
    `bp_update_kwargs` is omitted -- carries the dtype tolerance (:110-117)."""
    from tnqs_amd.core import _bp_opts
    g = tn.named_grid((3, 3))
    o1, _ = _bp_opts(g, dict(maxiter=10))
    assert o1.maxiter == 10 and o1.tolerance < 0
    o2, _ = _bp_opts(g, {})
    assert o2.maxiter == 0 and o2.tolerance < 0
    o3, _ = _bp_opts(g, None, dict(maxiter=25, tolerance=1e-5))
    assert o3.maxiter == 25 and o3.tolerance == 1e-5
    o4, _ = _bp_opts(g, dict(maxiter=7, tolerance=1e-9), dict(maxiter=25, tolerance=1e-5))
    assert o4.maxiter == 7 and o4.tolerance == 1e-9


def test_upstream_only_registry_entries_resolve_but_do_not_guess_a_matrix():
    """"Rz+" / "Rz+z+" (gate_definitions.jl:33,58): registered, aliased, locked -- their matrices live in an upstream ITensors.op the
    reference does not carry, so building them explains that instead of inventing a convention"""
    assert "Rz+" in tn.GATES and "Rz+z+" in tn.GATES and tn.ALIASES["rz+"] == "Rz+" and tn.ALIASES["rz+z+"] == "Rz+z+"
    with pytest.raises(ValueError, match="built-in"):
        tn.register_gate("Rz+", lambda t: np.eye(2))
    with pytest.raises(ValueError, match="upstream"):
        tn.gate_matrix("Rz+", 0.3)


def test_steiner_tree_of_an_observable_support():
    """steiner_region (graphs.py; src/expect.jl:67 -> Graphs.steiner_tree): terminals are in the tree, the tree is connected and minimal where the
    answer is unambiguous, ties are resolved deterministically, and the oracle's own statement of the rule picks the same vertices."""
    import tnqs_oracle as o
    g = tn.named_grid((4, 4)); og = o.named_grid((4, 4))
    region, parent = tn.steiner_region(g, [(1, 1), (1, 4)])
    assert region == [(1, 1), (1, 2), (1, 3), (1, 4)] and parent == [-1, 0, 1, 2]
    region, parent = tn.steiner_region(g, [(1, 1), (2, 2)])                       # two shortest paths: one of them, always the same
    assert len(region) == 3 and region[0] == (1, 1) and region[-1] == (2, 2) and region == tn.steiner_region(g, [(1, 1), (2, 2)])[0]
    region, parent = tn.steiner_region(g, [(2, 2), (2, 3), (3, 3), (3, 2)])       # a plaquette: its four vertices, three tree edges
    assert sorted(region) == [(2, 2), (2, 3), (3, 2), (3, 3)] and sorted(parent)[0] == -1 and sum(1 for q in parent if q >= 0) == 3
    import random
    rnd = random.Random(5)
    for hg, ho in ((g, og), (tn.heavy_hexagonal_lattice(2, 2), o.heavy_hexagonal_lattice(2, 2)), (tn.named_grid((3, 3, 2)), o.named_grid((3, 3, 2)))):
        for _ in range(60):
            vs = rnd.sample(list(hg.vertices), rnd.choice([2, 3, 4]))
            region, parent = tn.steiner_region(hg, vs)
            assert set(vs) <= set(region) and region[0] == vs[0]
            for i, q in enumerate(parent):
                assert (q == -1) == (i == 0) and (q < 0 or hg.has_edge(region[i], region[q]))
            leaves = set(region) - {region[q] for q in parent if q >= 0}
            assert leaves <= set(vs) or len(region) == 1                               # no dangling non-terminal
            assert sorted(region, key=hg.index.__getitem__) == o.steiner_vertices(ho, vs)
    with pytest.raises(ValueError):
        tn.steiner_region(tn.NamedGraph([1, 2, 3], [(1, 2)]), [1, 3])


def test_marshalled_circuit_cache_keys_on_gate_identity_and_registry():
    """core._marshal_circuit: the flat arrays of a circuit are reused for the SAME gate tuples on the same graph only, and not after the registry entry of
    one of its names changed; flat vertex lists are cached with a snapshot, array parameters and explicit matrices never."""
    from tnqs_amd import core
    g = tn.named_grid((2, 2))
    layer = [("Rx", (v,), 0.3) for v in g.vertices] + [("Rzz", (a, b), 0.2) for (a, b) in g.edges]      # vertex TUPLES: nothing of a gate can be edited in place
    a1 = core._marshal_circuit(layer, g); a2 = core._marshal_circuit(layer, g)
    assert a1[5] is a2[5] and a1[0] == len(layer)                                   # same arrays
    a3 = core._marshal_circuit(list(layer), g)                                      # another list of the same tuples: same circuit
    assert a3[5] is a1[5]
    rebuilt = [("Rx", (v,), 0.3) for v in g.vertices] + [("Rzz", (a, b), 0.2) for (a, b) in g.edges]
    a4 = core._marshal_circuit(rebuilt, g)
    assert a4[5] is not a1[5] and np.array_equal(a4[5], a1[5])
    # a vertex LIST -- the reference's own calling form, ("Rzz", [a, b], theta) -- can be mutated behind the identity key (round-4 advisor finding): it is cached
    # WITH a frozen snapshot of the list that every lookup compares (round-5 advisor finding: only tuple-form circuits were cached), so an in-place edit is seen
    mutable = [("Rx", [v], 0.3) for v in g.vertices] + [("Rzz", [a, b], 0.2) for (a, b) in g.edges]
    b1 = core._marshal_circuit(mutable, g); b2 = core._marshal_circuit(mutable, g)
    assert b1[5] is b2[5] and np.array_equal(b1[5], a1[5]) and np.array_equal(b1[3], a1[3])
    mutable[-1][1][0], mutable[-1][1][1] = mutable[-1][1][1], mutable[-1][1][0]      # swap the two vertices of the last gate in place
    b3 = core._marshal_circuit(mutable, g)
    assert b3[5] is not b1[5] and not np.array_equal(b3[3], b1[3])                    # the edit is seen: resolved again
    mutable[-1][1][0], mutable[-1][1][1] = mutable[-1][1][1], mutable[-1][1][0]      # ... and back: the ORIGINAL entry's snapshot matches again
    assert np.array_equal(core._marshal_circuit(mutable, g)[3], b1[3])
    mutable[0][1].append(g.vertices[1])                                              # a list that grew in place (now an invalid one-site gate on two vertices)
    assert core._marshal_circuit(mutable, g)[0] == len(mutable) and core._marshal_circuit(mutable, g)[3].size == b1[3].size + 1
    nested = [("Rzz", [[g.edges[0][0]], g.edges[0][1]], 0.2)]                         # anything stranger than a flat list is never cached (and rejected downstream)
    assert core._frozen_probe(nested[0]) is None
    arr_param = [("Rx", (g.vertices[0],), np.array(0.3))]
    assert core._marshal_circuit(arr_param, g)[5] is not core._marshal_circuit(arr_param, g)[5]
    assert core._marshal_circuit(layer, tn.named_grid((2, 2)))[5] is not a1[5]       # another graph object
    tn.register_gate("MyG", lambda: np.eye(2))
    try:
        c = [("MyG", [g.vertices[0]])]
        m1 = core._marshal_circuit(c, g)[5]
        tn.unregister_gate("MyG"); tn.register_gate("MyG", lambda: np.diag([1.0, -1.0]))
        m2 = core._marshal_circuit(c, g)[5]
        assert not np.array_equal(m1, m2)
    finally:
        tn.unregister_gate("MyG")
    mat = np.eye(2, dtype=complex)
    assert core._marshal_circuit([(mat, [g.vertices[0]])], g)[5] is not core._marshal_circuit([(mat, [g.vertices[0]])], g)[5]
