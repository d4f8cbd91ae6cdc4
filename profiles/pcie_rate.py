"""What the boundary costs when the state does NOT live on the device: BASELINE configs[1] (20 x 20, chi = 32, ComplexF32: 6.4 GB of site tensors) handed over as host
arrays (tnqs_set_site_tensor, pageable numpy memory), one Trotter layer, tensors read back (tnqs_get_site_tensor).  bench.py's `value` never includes this: a
simulation keeps its state in HBM for thousands of layers.  python profiles/pcie_rate.py [L] [chi]  -> one JSON line"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import tnqs_amd as tn  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 20
chi = int(sys.argv[2]) if len(sys.argv) > 2 else 32
g = tn.named_grid((L, L))
groups = tn.edge_color(g, 4)
layer = [("Rx", [v], 2 * 2.5 * 0.01) for v in g.vertices] + [("Rzz", [a, b], 2 * 1.0 * 0.01) for grp in groups for (a, b) in grp]
rng = np.random.default_rng(1)
host = {}
for v in g.vertices:
    shp = (2,) + (chi,) * g.degree(v); n = int(np.prod(shp))
    host[v] = (rng.standard_normal(2 * n, dtype=np.float32).view(np.complex64).reshape(shp) / np.float32(np.sqrt(n)))
nbytes = sum(t.nbytes for t in host.values())
bpc = tn.BeliefPropagationCache(tn.tensornetworkstate(np.complex64, lambda v: "↑", g))
kw = dict(maxdim=chi, cutoff=1e-10, normalize_tensors=True)
res = {}
for rep in range(2):                       # the second pass is the one reported (first touch of the pinned staging buffers, kernels loaded)
    t0 = time.perf_counter()
    for v in g.vertices:
        bpc._set_tensor(v, host[v])
    t1 = time.perf_counter()
    bpc = tn.update(bpc)
    out, _ = tn.apply_gates(layer, bpc, apply_kwargs=kw)
    _ = out.bond_dim(*g.edges[0])          # (blocks until the layer is through)
    tn.expect(out, ("Z", [g.vertices[0]]))
    t2 = time.perf_counter()
    back = [out.tensor(v) for v in g.vertices]
    t3 = time.perf_counter()
    res = {"L": L, "chi": chi, "state_GB": round(nbytes / 1e9, 2), "upload_s": round(t1 - t0, 3), "upload_GBps": round(nbytes / (t1 - t0) / 1e9, 2),
           "update_plus_layer_s": round(t2 - t1, 3), "download_s": round(t3 - t2, 3), "download_GBps": round(sum(b.nbytes for b in back) / (t3 - t2) / 1e9, 2),
           "two_site_gates": len(g.edges), "gates_per_s_with_the_state_crossing_PCIe_both_ways": round(len(g.edges) / (t3 - t0), 1)}
    del back
print(json.dumps(res))
