"""Per-site shapes of the BASELINE configurations that need 8 GPUs at full size, measured on one MI355X on a lattice that fits:
    python profiles/shape_bench.py cubic16     3x3x3 periodic cubic, chi = 16 (configs[3] per-site shape: degree 6, 268 MB tensors)
    python profiles/shape_bench.py heavyhex    heavy-hex (5,5), chi = 16 (configs[2]);   c1: 5x5, chi = 10, ComplexF64 (configs[0]) -- both latency-bound
    python profiles/shape_bench.py c128        LxL grid (L = 8), chi = 32 (CHI), ComplexF64: the reference's default element type at the bulk shape
    python profiles/shape_bench.py chi64       5x5 grid, chi = 64 (configs[4] per-site shape: degree 4, 268 MB bulk tensors, 256 x 256 theta)
Prints one JSON line: ms per layer, gates/s, BP sweeps, and per kernel class the HIP-event time, algorithmic TFLOP/s and TB/s (the
engine accumulates algorithmic flops / minimum bytes per class, include/tnqs.h tnqs_profile_get)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import tnqs_amd as tn  # noqa: E402


def unit_state(g, chi, seed):
    rng = np.random.default_rng(seed)
    for v in g.vertices:
        shp = (2,) + (chi,) * g.degree(v)
        n = int(np.prod(shp))
        yield v, rng.standard_normal(2 * n, dtype=np.float32).view(np.complex64).reshape(shp) / np.float32(np.sqrt(n))


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "cubic16"
    nrep = int(os.environ.get("NREP", 2))
    if mode == "cubic16":
        g = tn.named_grid((3, 3, 3), periodic=True); chi = 16
        groups = tn.edge_color(g)
        layer = [("Rz", (v,), -0.04) for v in g.vertices]
        for grp in groups:
            layer += [("Rxx", (a, b), -0.08) for (a, b) in grp]
        z = 6
    elif mode == "heavyhex":      # BASELINE configs[2]: heavy-hex (5,5), chi = 16, degrees 2 / 3 (examples/heavyhexIsing_dynamics.jl circuit); latency-bound
        g = tn.heavy_hexagonal_lattice(5, 5); chi = 16
        groups = tn.edge_color(g, 3)
        layer = [("Rx", (v,), 0.4) for v in g.vertices]
        for grp in groups:
            layer += [("Rzz", (a, b), 0.1) for (a, b) in grp]
        z = 3
    elif mode == "c1":            # BASELINE configs[0]: 5x5 TFIM, chi = 10, ComplexF64 (the reference's CPU-runnable case); latency-bound
        g = tn.named_grid((5, 5)); chi = 10; groups = tn.edge_color(g, 4)
        layer = [("Rx", (v,), 2 * 2.5 * 0.01) for v in g.vertices]
        for grp in groups:
            layer += [("Rzz", (a, b), 2 * 1.0 * 0.01) for (a, b) in grp]
        z = 4
    elif mode == "c128":
        L = int(os.environ.get("L", 8)); chi = int(os.environ.get("CHI", 32))
        g = tn.named_grid((L, L)); groups = tn.edge_color(g, 4)
        layer = [("Rx", (v,), 2 * 2.5 * 0.01) for v in g.vertices]
        for grp in groups:
            layer += [("Rzz", (a, b), 2 * 1.0 * 0.01) for (a, b) in grp]
        z = 4
    else:
        L = int(os.environ.get("L", 5)); chi = 64
        g = tn.named_grid((L, L)); groups = tn.edge_color(g, 4)
        layer = [("Rx", (v,), 2 * 2.5 * 0.01) for v in g.vertices]
        for grp in groups:
            layer += [("Rzz", (a, b), 2 * 1.0 * 0.01) for (a, b) in grp]
        z = 4
    dt_ = np.complex128 if mode in ("c128", "c1") else np.complex64
    bpc = tn.BeliefPropagationCache(tn.tensornetworkstate(dt_, lambda v: "↑", g))
    for v, t in unit_state(g, chi, 1234):
        bpc._set_tensor(v, t.astype(dt_))
    kw = dict(maxdim=chi, cutoff=1e-10, normalize_tensors=True)
    for _ in range(2):
        bpc, _ = tn.apply_gates(layer, bpc, apply_kwargs=kw)
    tn.profile_enable(bpc, True); tn.profile_reset(bpc)
    t0 = time.perf_counter()
    sweeps = 0
    for _ in range(nrep):
        info = {}
        bpc, errs = tn.apply_gates(layer, bpc, apply_kwargs=kw, info=info)
        sweeps += info["n_sweeps"]
    dt = (time.perf_counter() - t0) / nrep
    pr = tn.profile_get(bpc)
    # SURVEY 8d algorithmic work of the BULK site shape (upper bound on open lattices): per gate 2 (2 n d + 3 d^2) chi^(z+1) cMAC, per
    # message (n + 1) d chi^(z+1) cMAC, n = z - 1, d = 2; 8 flop per cMAC
    n = z - 1
    flops = 8.0 * (g.ne() * 2 * (2 * n * 2 + 12) * chi ** (z + 1) + (sweeps / nrep) * 2 * g.ne() * (n + 1) * 2 * chi ** (z + 1))
    classes = {}
    for k, v in pr.items():
        if v["launches"]:
            ms = v["ms"] / nrep
            classes[k] = dict(ms=round(ms, 2), launches=v["launches"] // nrep,
                              tflops=round(v["flops"] / nrep / ms / 1e9, 1) if ms > 0 else None,
                              tbps=round(v["bytes"] / nrep / ms / 1e9, 2) if ms > 0 else None)
    print(json.dumps(dict(mode=mode, lattice=f"{g.nv()} sites / {g.ne()} edges", chi=chi, ms_per_layer=round(dt * 1e3, 1),
                          gates_per_s=round(g.ne() / dt, 1), bp_sweeps_per_layer=sweeps / nrep, colours=len(groups),
                          bulk_formula_tflops=round(flops / dt / 1e12, 1), classes=classes)))


if __name__ == "__main__":
    main()
