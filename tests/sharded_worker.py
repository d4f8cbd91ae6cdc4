"""worker of tests/test_gpu_sharded.py: run as `python -m torch.distributed.run --nproc-per-node 2 tests/sharded_worker.py out.npz`
(gloo backend, both ranks on GPU 0): a sharded 2-rank apply_gates must reproduce the single-rank result."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tnqs_amd as tn


def run(bpc_factory, layer, kw, bpkw, nlayers):
    bpc = bpc_factory()
    bpc = tn.update(bpc, **bpkw)
    errs_all = []
    for _ in range(nlayers):
        bpc, errs = tn.apply_gates(layer, bpc, apply_kwargs=kw, bp_update_kwargs=bpkw)
        errs_all.append(errs)
    return bpc, np.array(errs_all)


def main():
    out = sys.argv[1]
    backend = os.environ.get("TNQS_WORKER_BACKEND", "gloo")      # "nccl": one rank per GPU, the library's own RCCL transport
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    mode = sys.argv[2] if len(sys.argv) > 2 else "c128"
    dtype = np.complex64 if mode in ("c64", "chi32") else np.complex128
    # "chi32": 4x4 grid at bond dimension 32 -- the degree-4 sites run the plane kernels (shared pair products, epilogue kernel,
    # Cholesky factors, deferred normalisation) under sharding
    big = mode == "chi32"
    g = tn.named_grid((4, 4)) if big else tn.named_grid((4, 3))
    groups = tn.edge_color(g, 4)
    layer = [("Rx", [v], 0.5) for v in g.vertices] + [("Rz", [v], 0.4) for v in g.vertices]
    seq = []
    for grp in groups:
        layer += [("Rzz", [a, b], 0.25) for (a, b) in grp]
        seq += list(grp) + [(b, a) for (a, b) in grp]
    kw = dict(maxdim=32 if big else 3, cutoff=1e-10, normalize_tensors=True)
    bpkw = dict(edge_sequence=seq, maxiter=3 if big else 12, tolerance=None)
    psi0 = tn.random_tensornetworkstate(dtype, g, bond_dimension=32 if big else 2, seed=11)
    if big:
        for v in g.vertices:
            psi0.tensors[v] = (psi0.tensors[v] / np.linalg.norm(psi0.tensors[v])).astype(dtype)
    nlayers = 1 if big else 2
    if mode in ("z6chi16", "z4chi64"):
        # the per-site shapes of BASELINE configs[3] / [4] under sharding: two adjacent degree-6 hubs at chi = 16 on different ranks (the
        # bulk gate of the cubic lattice as a cross-rank gate; tests/test_gpu_fullsize.py double_wheel), and the 3x3 grid at chi = 64
        # (268 MB centre tensor, 256 x 256 theta) -- one layer, sharded against single rank
        dtype = np.complex64
        if mode == "z6chi16":
            g = tn.NamedGraph(list(range(12)), [(0, 6)] + [(0, 1 + i) for i in range(5)] + [(6, 7 + i) for i in range(5)] + [(1 + i, 7 + i) for i in range(5)])
            chi = 16; groups = tn.edge_color(g)
            layer = [("Rz", [v], -0.04) for v in g.vertices]; seq = []
            for grp in groups:
                layer += [("Rxx", [a, b], -0.08) for (a, b) in grp]
                seq += list(grp) + [(b, a) for (a, b) in grp]
        else:
            g = tn.named_grid((3, 3)); chi = 64; groups = tn.edge_color(g, 4)
            layer = [("Rx", [v], 0.05) for v in g.vertices]; seq = []
            for grp in groups:
                layer += [("Rzz", [a, b], 0.02) for (a, b) in grp]
                seq += list(grp) + [(b, a) for (a, b) in grp]
        kw = dict(maxdim=chi, cutoff=1e-10, normalize_tensors=True)
        bpkw = dict(edge_sequence=seq, maxiter=2, tolerance=None)
        psi0 = tn.random_tensornetworkstate(dtype, g, bond_dimension=chi, seed=13)
        for v in g.vertices:
            psi0.tensors[v] = (psi0.tensors[v] / np.linalg.norm(psi0.tensors[v])).astype(dtype)
        nlayers = 1; big = True
    if mode == "illc128":
        # ComplexF64 with cutoff = 1e-14 from a product state: kept singular values span 1e-7, the regime in which the gate path needs its
        # second factorisation pass -- sharded and single-rank runs must both take it
        g = tn.named_grid((4, 4)); groups = tn.edge_color(g, 4)
        layer = [("Rx", [v], 2 * 2.5 * 0.05) for v in g.vertices]; seq = []
        for grp in groups:
            layer += [("Rzz", [a, b], 2 * 1.0 * 0.05) for (a, b) in grp]
            seq += list(grp) + [(b, a) for (a, b) in grp]
        kw = dict(maxdim=8, cutoff=1e-14, normalize_tensors=True)
        bpkw = dict(edge_sequence=seq, maxiter=100, tolerance=1e-14)
        psi0 = tn.tensornetworkstate(dtype, lambda v: "↑", g)
        nlayers = 8

    if mode == "pendshard":
        # round-3 advisor finding: one-site gates deferred on an UNSHARDED handle (State::pend1) must be applied when the handle is sharded
        # afterwards -- otherwise the owner of a vertex applies and clears them in its accessors while the partner rank of a later cross-rank
        # gate still folds them into the gate matrix.  Evolve unsharded, end on a one-site layer (deferred: the tensors have unit norm after
        # the two-site gates), shard, read <Z> on the owners (accessor), then two more layers.
        dtype = np.complex64
        psi0 = tn.random_tensornetworkstate(dtype, g, bond_dimension=2, seed=11)
        tail = [("Rx", [v], 0.3 + 0.01 * i) for i, v in enumerate(g.vertices)]

        def pre(shard_it):
            b = tn.update(tn.BeliefPropagationCache(psi0, device=dev), **bpkw)
            info = {}
            b, _ = tn.apply_gates(layer + tail, b, apply_kwargs=kw, bp_update_kwargs=bpkw, info=info)
            assert info["n_deferred_1site"] >= len(tail), info
            if shard_it:
                tn.shard(b, rank, world, exch_bytes=8 << 20, max_chi=64)
                tn.expect_all(b, "Z")          # an accessor that only touches owned vertices
            return b

    def sharded_factory():
        if mode == "pendshard":
            return pre(True)
        b = tn.BeliefPropagationCache(tn.tensornetworkstate(dtype, lambda v: "↑", g), device=dev)
        tn.shard(b, rank, world, exch_bytes=(64 << 20) if big else (8 << 20), max_chi=64)
        for v in g.vertices:
            if b.owns(v):
                b._set_tensor(v, psi0.tensors[v])
            else:
                b._declare_dims(v, psi0.tensors[v].shape)
        return b

    bs, es = run(sharded_factory, layer, kw, bpkw, nlayers)
    # every rank holds all messages and bond dims; site tensors / <Z> only for owned vertices
    ez = tn.expect_all(bs, "Z")
    ez_t = torch.from_numpy(np.nan_to_num(ez.view(np.float64), nan=0.0).copy())
    dist.all_reduce(ez_t)                                   # owners contribute, others contribute zeros
    ez_full = ez_t.numpy().view(np.complex128)
    spectra = []
    for (a, b) in g.edges:
        for e in ((a, b), (b, a)):
            m = bs.message(e).astype(np.complex128)
            w = np.linalg.eigvalsh((m + m.conj().T) / 2)
            spectra.append(w / w.sum())
    dims = np.array([bs.bond_dim(a, b) for (a, b) in g.edges])
    # multi-site observables and the symmetric gauge on the sharded handle (every rank takes part in the region's exchanges)
    extra = mode in ("c128", "c64")
    vs = list(g.vertices)
    # regions with a unique Steiner tree: neighbours, and vertices on one lattice line (also across the rank boundary)
    line = [v for v in vs if v[0] == vs[0][0]] if extra else []
    regions = [[vs[0], vs[1]], [line[0], line[-1]], [line[0], line[1], line[-1]], [(1, 1), (2, 2)], [(2, 1), (2, 2), (3, 2), (3, 1)]] if extra else []      # + a diagonal pair (tie) and a plaquette (loop) across the rank boundary
    def multi(b):
        out = []
        for r in regions:
            try:
                out.append(complex(tn.expect(b, ("Z" * len(r), r))))
            except (ValueError, tn.TnqsError):      # no unique Steiner tree for this pair: skipped on both sides
                out.append(complex("nan"))
        return np.array(out)
    zz_sh = multi(bs) if extra else np.zeros(0)
    mraw_sh = [bs.message(e) for e in ((vs[0], vs[1]), (vs[1], vs[0]))] if extra else []
    if extra:
        bg = tn.symmetric_gauge(bs)
        ezg = tn.expect_all(bg, "Z")
        ezg_t = torch.from_numpy(np.nan_to_num(ezg.view(np.float64), nan=0.0).copy())
        dist.all_reduce(ezg_t)
        ezg_full = ezg_t.numpy().view(np.complex128)
        sg = np.sort(np.abs(np.diag(bg.message((vs[0], vs[1])))))
    else:
        ezg_full, sg = np.zeros(0), np.zeros(0)
    nex = bs._shard.n_exchanges
    if rank == 0:
        bu, eu = run((lambda: pre(False)) if mode == "pendshard" else (lambda: tn.BeliefPropagationCache(psi0, device=dev)), layer, kw, bpkw, nlayers)
        ezu = tn.expect_all(bu, "Z")
        spu = []
        for (a, b) in g.edges:
            for e in ((a, b), (b, a)):
                m = bu.message(e).astype(np.complex128)
                w = np.linalg.eigvalsh((m + m.conj().T) / 2)
                spu.append(w / w.sum())
        if extra:
            zz_un = multi(bu); bgu = tn.symmetric_gauge(bu); ezg_un = tn.expect_all(bgu, "Z"); sg_un = np.sort(np.abs(np.diag(bgu.message((vs[0], vs[1])))))
        else:
            zz_un, ezg_un, sg_un = np.zeros(0), np.zeros(0), np.zeros(0)
        if extra:
            mraw_un = [bu.message(e) for e in ((vs[0], vs[1]), (vs[1], vs[0]))]
            print("raw message difference before the gauge:", [float(np.max(np.abs(a - b))) for a, b in zip(mraw_sh, mraw_un)],
                  " sums:", [complex(a.sum()) for a in mraw_sh], [complex(b.sum()) for b in mraw_un], flush=True)
            print("S sharded:", sg[-4:], " S single:", sg_un[-4:], " normalised difference:", float(np.max(np.abs(sg / np.linalg.norm(sg) - sg_un / np.linalg.norm(sg_un)))), flush=True)
        np.savez(out, zz_sh=zz_sh, zz_un=zz_un, ezg_sh=ezg_full, ezg_un=ezg_un, sg_sh=sg, sg_un=sg_un, errs_sh=es, errs_un=eu, ez_sh=ez_full, ez_un=ezu, sp_sh=np.concatenate(spectra), sp_un=np.concatenate(spu),
                 dims_sh=dims, dims_un=np.array([bu.bond_dim(a, b) for (a, b) in g.edges]), n_exchanges=nex,
                 transport=type(bs._shard).__name__)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
