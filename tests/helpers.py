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
