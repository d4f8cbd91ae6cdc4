"""BP sweeps alone on the 3x3x3 periodic cubic lattice at chi = 16 (configs[3] per-site shape): three sweeps without a tolerance, kernel
classes by HIP events.   python profiles/bp16_bench.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tnqs_amd as tn
g = tn.named_grid((3, 3, 3), periodic=True); chi = 16
rng = np.random.default_rng(1)
bpc = tn.BeliefPropagationCache(tn.tensornetworkstate(np.complex64, lambda v: "↑", g))
for v in g.vertices:
    shp = (2,) + (chi,) * g.degree(v); n = int(np.prod(shp))
    bpc._set_tensor(v, rng.standard_normal(2 * n, dtype=np.float32).view(np.complex64).reshape(shp) / np.float32(np.sqrt(n)))
bpc = tn.update(bpc, maxiter=1, tolerance=None)
tn.profile_enable(bpc, True); tn.profile_reset(bpc)
t0 = time.perf_counter()
bpc = tn.update(bpc, maxiter=3, tolerance=None)
dt = time.perf_counter() - t0
pr = tn.profile_get(bpc)
print(json.dumps({"ms": round(dt * 1e3, 1), **{k: dict(ms=round(v["ms"], 2), n=v["launches"], tbps=round(v["bytes"] / v["ms"] / 1e9, 2)) for k, v in pr.items() if v["launches"]}}))
