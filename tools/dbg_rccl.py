import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, tnqs_amd as tn
mode = sys.argv[1]
if mode in ("self", "both"):
    tn.dist.rccl_selftest(0, 1 << 20); print("selftest ok", flush=True)
if mode in ("shard", "both"):
    g = tn.named_grid((2, 2))
    b = tn.BeliefPropagationCache(tn.random_tensornetworkstate(np.complex64, g, bond_dimension=2, seed=1))
    sh = tn.shard(b, 0, 1, transport="rccl", exch_bytes=1 << 20)
    b2 = tn.update(b, maxiter=3, tolerance=None)
    print("shard ok", tn.expect(b2, ("Z", [g.vertices[0]])), flush=True)
    if len(sys.argv) > 2:
        del b2, b, sh
        import gc; gc.collect(); print("deleted", flush=True)
print("exiting", flush=True)
