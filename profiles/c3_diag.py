import sys, os
ROOT = "/root/repo"
sys.path[:0] = [ROOT, ROOT + "/oracle", ROOT + "/tests"]
import numpy as np, tnqs_amd as tn, tnqs_oracle as o
from helpers import to_oracle_state
g = tn.heavy_hexagonal_lattice(5, 5)
groups = tn.edge_color(g, 3)
layer = [("Rx", [v], 0.4) for v in g.vertices]
for grp in groups:
    layer += [("Rzz", [a, b], np.pi / 2) for (a, b) in grp]
psi = tn.tensornetworkstate(np.complex64, lambda v: "↑", g)
bpkw = dict(edge_sequence=tn.forest_cover_edge_sequence(g), maxiter=6, tolerance=None)
kw = dict(maxdim=16, cutoff=1e-12, normalize_tensors=True)
bd = tn.update(tn.BeliefPropagationCache(psi), **bpkw)
bo = o.update(o.BeliefPropagationCache(to_oracle_state(psi)), **bpkw)
for it in range(5):
    info = {}
    bd, ed = tn.apply_gates(layer, bd, apply_kwargs=kw, bp_update_kwargs=bpkw, info=info)
    bo, eo = o.apply_gates(layer, bo, apply_kwargs=kw, bp_update_kwargs=bpkw)
    dd = [bd.bond_dim(a, b) for (a, b) in g.edges]; do = [bo.tns.bond_dim(a, b) for (a, b) in g.edges]
    bad = [i for i in range(len(dd)) if dd[i] != do[i]]
    print("layer", it, "mismatching bonds", len(bad), "svd sweeps max", info.get("n_svd_sweeps_max"), "avg", info.get("n_svd_sweeps") / max(1, info.get("n_two_site", 1)))
    # the two-site gate index of each edge in the layer, to look up truncation errors
    eidx = {}
    for k, gt in enumerate(layer):
        if len(gt[1]) == 2: eidx[frozenset(gt[1])] = k
    for i in bad[:6]:
        a, b = g.edges[i]
        k = eidx[frozenset((a, b))]
        print("  bond", (a, b), "dims dev/oracle", dd[i], do[i], " truncerr dev/oracle", ed[k], eo[k], " diff", ed[k] - eo[k])
    if bad: break
