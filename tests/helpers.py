"""shared helpers for the parity tests: conversions between the host package and the CPU oracle"""
import numpy as np

import tnqs_oracle as o


def to_oracle_graph(g):
    return o.Graph(list(g.vertices), list(g.edges))


def to_oracle_state(tns):
    og = to_oracle_graph(tns.graph)
    return o.TensorNetworkState(og, {v: np.array(tns.tensors[v]) for v in tns.graph.vertices})


def oracle_cache_from_device(bpc):
    """download a device cache into an oracle BeliefPropagationCache (tensors + every message)"""
    tns = bpc.network()
    oc = o.BeliefPropagationCache(to_oracle_state(tns), edge_sequence=[])
    for (a, b) in bpc.graph.edges:
        oc.messages[(a, b)] = bpc.message((a, b))
        oc.messages[(b, a)] = bpc.message((b, a))
    return oc


def colour_sequence(g, groups):
    """the library's default sweep order expressed with an explicit colouring: per colour, forward then reverse"""
    seq = []
    for grp in groups:
        seq += [(a, b) for (a, b) in grp]
        seq += [(b, a) for (a, b) in grp]
    return seq


def tfim_layer(g, groups, dt=0.25, hx=1.0, hz=0.8, J=0.5):
    layer = [("Rx", [v], 2 * hx * dt) for v in g.vertices]
    layer += [("Rz", [v], 2 * hz * dt) for v in g.vertices]
    for grp in groups:
        layer += [("Rzz", [a, b], 2 * J * dt) for (a, b) in grp]
    return layer


def msg_close(a, b, tol):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) <= tol * max(1.0, np.max(np.abs(b)))


def htse_free_energy(mod, nsteps=25, dbeta=0.01, J=1.0, maxdim=16, dtype=np.complex128, every=5, bp_update_kwargs=None, cutoff=1e-14):
    """the reference's thermal-state example (examples/hexagonal_heisenbergmodel_thermalstate.jl:7-40) on either side
    (`mod` = the oracle module or the device package): identity operator on two site indices per vertex (here one d = 4
    site, index (s, ancilla)), imaginary-time Heisenberg gates Rxxyyzz(theta = -i J dbeta / 2) on the first index, log Z
    accumulated from freenergy + rescale! after every layer.  The 2x2 periodic hexagonal lattice of the example is an
    8-vertex 3-regular graph; BP sees only the degree, so any 3-regular graph gives the same density -- built here as
    the periodic 4x2 grid (two rings of 4 joined by rungs).  Returns [(beta, f_bp, f_htse4)]."""
    pauli = [np.array([[0, 1], [1, 0]], complex), np.array([[0, -1j], [1j, 0]]), np.diag([1.0, -1.0]).astype(complex)]
    h = 0.5 * sum(np.kron(np.kron(p, np.eye(2)), np.kron(p, np.eye(2))) for p in pauli)      # gate_definitions.jl:276-279
    w, q = np.linalg.eigh(h)
    gate = (q * np.exp(-0.5 * J * dbeta * w)) @ q.conj().T                                   # exp(-i theta h), theta = -i J dbeta / 2
    g = mod.named_grid((4, 2), periodic=True)
    assert all(g.degree(v) == 3 for v in g.vertices)
    tensors = {v: np.eye(2, dtype=dtype).reshape((4, 1, 1, 1)) for v in g.vertices}         # tensornetworkstate_constructors.jl:21-39
    bpkw = dict(bp_update_kwargs(g)) if bp_update_kwargs is not None else {}     # the example runs on the defaults
    bpc = mod.update(mod.BeliefPropagationCache(mod.TensorNetworkState(g, tensors)), **bpkw)
    gates = [(gate, [a, b]) for grp in mod.edge_color(g) for (a, b) in grp]
    kw = dict(maxdim=maxdim, cutoff=cutoff, normalize_tensors=False)
    logz = -np.log(complex(mod.partitionfunction(bpc)))
    bpc = mod.rescale(bpc)
    out = []
    for i in range(1, nsteps + 1):
        bpc, _ = mod.apply_gates(gates, bpc, apply_kwargs=kw, bp_update_kwargs=bpkw or None)
        logz -= np.log(complex(mod.partitionfunction(bpc)))
        bpc = mod.rescale(bpc)
        if i % every == 0:
            b = 2 * i * dbeta
            out.append((b, (logz / len(g.vertices)).real,
                        -np.log(2) - 9 / 64 * (J * b) ** 2 - 3 / 128 * (J * b) ** 3 + 27 / 2048 * (J * b) ** 4))
    return out


def two_site_tensor(og, a, b, ta, tb):
    """psi_a psi_b contracted over their shared bond: [s_a, outer legs of a..., s_b, outer legs of b...] -- invariant under the gauge
    freedom of the bond (SVD phases, rotations inside degenerate clusters)"""
    return np.tensordot(ta, tb, axes=([og.leg(a, b)], [og.leg(b, a)]))


def gauged_two_site(bo, a, b, T):
    """T with sqrt(incoming message) on every outer leg: the metric in which simple update truncates (simple_update.jl:27-44).
    `bo` is the oracle cache BEFORE the gate (its messages define the metric)."""
    og = bo.g
    pos = 1
    for site, other in ((a, b), (b, a)):
        for k in og.nbrs[site]:
            if k == other:
                continue
            m = np.asarray(bo.message((k, site)), dtype=np.complex128)
            ev, q = np.linalg.eigh((m + m.conj().T) / 2)
            T = np.moveaxis(np.tensordot(T, (q * np.sqrt(np.clip(ev, 0, None))) @ q.conj().T, axes=([pos], [0])), -1, pos)
            pos += 1
        pos += 1
    return T


def exact_two_site(bo, a, b, gate):
    """the untruncated gate application on the two-site tensor; gate is (d_a d_b) x (d_a d_b), first vertex most significant"""
    ta, tb = np.asarray(bo.tns.tensors[a], dtype=np.complex128), np.asarray(bo.tns.tensors[b], dtype=np.complex128)
    T0 = two_site_tensor(bo.g, a, b, ta, tb)
    da, db = ta.shape[0], tb.shape[0]
    na = ta.ndim - 1
    T0 = np.moveaxis(T0, na, 1)                                           # [s_a, s_b, outer a..., outer b...]
    Tex = np.tensordot(np.asarray(gate, dtype=np.complex128).reshape(da, db, da, db), T0, axes=([2, 3], [0, 1]))
    return np.moveaxis(Tex, 1, na)


def c64_errs_close(errs, oerrs, rel=2e-3, floor=3e-7):
    """ComplexF32 truncation errors against the oracle: relative (an absolute 1e-5 would pass a 10 % error on a truncation error of 1e-4),
    with a floor at the f32 rounding level of the normalised spectrum"""
    e, f = np.asarray(errs, dtype=float), np.asarray(oerrs, dtype=float)
    return bool(np.all(np.abs(e - f) < rel * np.maximum(np.abs(e), np.abs(f)) + floor))


def bond_dims_agree(dims_dev, dims_or, errs_dev, errs_or, gate_of_edge, cutoff):
    """bond dimensions of a ComplexF32 run against the oracle's.  A relative cutoff of 1e-12 on sigma^2 asks whether a singular value of 1e-6 sigma_max is
    kept -- f32 arithmetic resolves singular values to ~1e-7 sigma_max (LAPACK's backward error as much as the device's), so two correct implementations
    disagree on a value that sits AT the cutoff.  Equal everywhere, except that a bond may differ by one where the two truncation errors differ by less than the
    cutoff itself -- i.e. where the singular value in question carries a weight within rounding of the threshold.  Returns (ok, list of such bonds)."""
    noise = []
    for k, (a, b) in enumerate(zip(dims_dev, dims_or)):
        if a == b:
            continue
        g = gate_of_edge[k]
        if abs(a - b) > 1 or g is None or not abs(float(errs_dev[g]) - float(errs_or[g])) <= cutoff:
            return False, [k]
        noise.append(k)
    return True, noise

