"""Generates tests/golden/theta_factors.npz: the (R1, R2, gate) triples the theta SVD of a two-site gate starts from, harvested from ORACLE runs
(oracle/tnqs_oracle.py simple_update: the R factors of the two thin QR factorisations, simple_update.jl:47-48, and the gate matrix):
  rand32_*  : 4 x 4 grid, iid complex-normal chi = 32 ComplexF32 state (the benchmark's synthetic state), TFIM layer at dt = 0.01, gates of the fourth layer
              (bulk-bulk gates: 64 x 2 x 32 factors; rand32_b*: a corner-edge gate, 32 / 64 rows);
  evol16_*  : 4 x 4 grid evolved from the product state at dt = 0.1 with maxdim 16, gates of the sixteenth layer (bonds saturated, truncation live).
Used by tests/test_gpu_kernels.py::test_theta_svd_pre_kernel_on_harvested_factors and profiles/svd_bench.py.     python tests/golden/make_theta_factors.py
(about two minutes on a desktop CPU; deterministic: seeded state, LAPACK)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "oracle")]
import tnqs_oracle as o

rec = []
_svd = np.linalg.svd


def svd_spy(a, *args, **kw):          # simple_update calls np.linalg.svd on theta with r1, r2, gate in scope
    f = sys._getframe(1)
    if "r1" in f.f_locals and "r2" in f.f_locals and "gate" in f.f_locals:
        rec.append((f.f_locals["r1"].copy(), f.f_locals["r2"].copy(), f.f_locals["gate"].copy()))
    return _svd(a, *args, **kw)


np.linalg.svd = svd_spy


def tfim_layer(g, groups, dt, J=1.0, hx=2.5):
    layer = [("Rx", [v], 2 * hx * dt) for v in g.vertices]
    for grp in groups:
        layer += [("Rzz", [a, b], 2 * J * dt) for (a, b) in grp]
    return layer


def harvest(mode, chi, nlayers):
    g = o.named_grid((4, 4)); groups = o.edge_color(g)
    kw = dict(maxdim=chi, cutoff=1e-10, normalize_tensors=True)
    if mode == "random":
        psi, dt = o.random_state(np.complex64, g, chi, seed=1234), 0.01
    else:
        psi, dt = o.product_state(np.complex64, lambda v: "↑", g), 0.1
    bpc = o.update(o.BeliefPropagationCache(psi))
    for _ in range(nlayers):
        rec.clear()
        bpc, _errs = o.apply_gates(tfim_layer(g, groups, dt), bpc, apply_kwargs=kw)
    return list(rec)


out = {}
recs = harvest("random", 32, 4)
big = [t for t in recs if t[0].shape[0] == 64 and t[1].shape[0] == 64]
small = [t for t in recs if t[0].shape[0] != 64 or t[1].shape[0] != 64]
for i, (r1, r2, gt) in enumerate(big[:4]):
    out[f"rand32_{i}_r1"], out[f"rand32_{i}_r2"], out[f"rand32_{i}_gate"] = r1.astype(np.complex64), r2.astype(np.complex64), gt.astype(np.complex128)
for i, (r1, r2, gt) in enumerate(small[:2]):
    out[f"rand32_b{i}_r1"], out[f"rand32_b{i}_r2"], out[f"rand32_b{i}_gate"] = r1.astype(np.complex64), r2.astype(np.complex64), gt.astype(np.complex128)
recs = harvest("evolved", 16, 16)
big = [t for t in recs if t[0].shape[0] == 32 and t[1].shape[0] == 32]
for i, (r1, r2, gt) in enumerate(big[:3]):
    out[f"evol16_{i}_r1"], out[f"evol16_{i}_r2"], out[f"evol16_{i}_gate"] = r1.astype(np.complex64), r2.astype(np.complex64), gt.astype(np.complex128)
np.savez(os.path.join(ROOT, "tests", "golden", "theta_factors.npz"), **out)
print({k: v.shape for k, v in out.items() if k.endswith("_r1")})
