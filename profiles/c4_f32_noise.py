"""Whose ComplexF32 noise is the 6e-6 of the C4 test?  The 3x3x3 periodic cubic lattice at chi = 16 (four colour groups of a 3-D Ising layer, as tests/test_gpu_fullsize.py::
test_c4_periodic_cubic_layer_matches_oracle): <Z> of the device in ComplexF32, of the device in ComplexF64 and of the oracle in ComplexF32.  Measured (round 4): dev32 - orc32 6.0e-6,
dev32 - dev64 1.2e-7, orc32 - dev64 6.0e-6 -- the distance between device and oracle is the ORACLE's f32 accumulation noise (numpy sums 1.7e7 terms in f32; the device accumulates
Gram matrices in f64 and its f32 matrix-core sums are blocked).  python profiles/c4_f32_noise.py   (4 minutes, mostly the oracle)"""
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import tnqs_amd as tn, tnqs_oracle as o, cpu_layer
from helpers import to_oracle_state
from test_gpu_fullsize import small_norm_state
g = tn.named_grid((3, 3, 3), periodic=True); chi = 16
psi = small_norm_state(g, chi, seed=33)
groups = tn.edge_color(g)
seq = []
for grp in groups: seq += list(grp) + [(b, a) for (a, b) in grp]
J, h, dt = -1.0, -1.0, 0.04
one_site = [("Rz", [v], h * dt) for v in g.vertices]
groups = groups[:4]
cg = [[("Rxx", [a, b], 2 * J * dt) for (a, b) in grp] for grp in groups]
layer = one_site + [gt for grp in cg for gt in grp]
kw = dict(maxdim=chi, cutoff=1e-10, normalize_tensors=True)
bk = dict(edge_sequence=seq, maxiter=1, tolerance=None)
res = {}
for name, dtp in (("dev32", np.complex64), ("dev64", np.complex128)):
    p = tn.TensorNetworkState(g, {v: psi.tensors[v].astype(dtp) for v in g.vertices})
    b = tn.update(tn.BeliefPropagationCache(p), **bk)
    b, e = tn.apply_gates(layer, b, apply_kwargs=kw, bp_update_kwargs=bk)
    res[name] = tn.expect_all(b, "Z").real; print(name, "done", flush=True)
    del b
zop = np.diag([1.0, -1.0]).astype(complex)
with cpu_layer.parallel_oracle() as pool:
    bo = o.BeliefPropagationCache(to_oracle_state(psi), edge_sequence=seq)
    bo = cpu_layer.update(bo, pool, maxiter=1, tolerance=None)
    bo, eo, _ = cpu_layer.apply_layer(bo, one_site, cg, pool, kw, dict(maxiter=1, tolerance=None))
res["orc32"] = np.array([o.expect_1site(bo, zop, v).real for v in g.vertices])
for a, b in (("dev32", "orc32"), ("dev32", "dev64"), ("orc32", "dev64")):
    print(f"C4 shape max|dZ| {a} - {b}: {np.max(np.abs(res[a] - res[b])):.2e}")
