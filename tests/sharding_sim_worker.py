"""CPU worker (gloo, world_size 2) of tests/test_sharding_cpu.py: simulates the library's vertex-sharded protocol
with the numpy pieces (oracle message update, tests/gram_update_ref.py) and the product's own partition helper,
exchanging exactly what the C++ engine exchanges: raw messages per BP level, Gram matrices per gate batch, and the
per-gate record (chi', truncerr, S, X2).  Rank 0 compares with the serial oracle."""
import os
import pickle
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import tnqs_oracle as o
import gram_update_ref as gr
import tnqs_amd as tn            # host logic under test: partition_vertices (the .so loads on CPU, nothing is launched)


def allgather_obj(x):
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, x)
    return out


def sharded_bp_sweep(g, owner, rank, tensors, messages, levels, dtype):
    """one Gauss-Seidel sweep executed level by level; owners compute, everyone normalises (engine.cpp bp_update_t)"""
    for lev in levels:
        mine = {}
        for (u, v) in lev:
            if owner[g.pos[u]] != rank:
                continue
            loc = o.BeliefPropagationCache(o.TensorNetworkState.__new__(o.TensorNetworkState), messages, edge_sequence=[])
            loc.tns.g, loc.tns.tensors = g, tensors
            mine[(u, v)] = o.updated_message(loc, (u, v), normalize=False)
        for part in allgather_obj(mine):                       # exchange #: raw messages of the level
            for e, m in part.items():
                s = m.sum()
                messages[e] = (m / s if s != 0 else m).astype(dtype)
    return messages


def main():
    out = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dtype = np.complex128
    g = o.named_grid((4, 3))
    owner = tn.partition_vertices(len(g.vertices), world)
    groups = o.edge_color(g)
    psi = o.random_state(dtype, g, 2, seed=11)
    # each rank keeps ONLY its own site tensors
    tensors = {v: (psi.tensors[v] if owner[g.pos[v]] == rank else None) for v in g.vertices}
    dims = {frozenset(e): psi.bond_dim(*e) for e in g.edges}
    messages = {}
    for (a, b) in g.edges:
        messages[(a, b)] = np.eye(dims[frozenset((a, b))], dtype=dtype)
        messages[(b, a)] = np.eye(dims[frozenset((a, b))], dtype=dtype)
    levels = [[(a, b) for (a, b) in grp] + [(b, a) for (a, b) in grp] for grp in groups]
    seq = [e for lev in levels for e in lev]
    nsweeps = 8
    kw = dict(maxdim=3, cutoff=1e-10, normalize_tensors=True)
    gate = o.gate_matrix("Rzz", 0.25)

    def bp(messages):
        for _ in range(nsweeps):
            messages = sharded_bp_sweep(g, owner, rank, tensors, messages, levels, dtype)
        return messages

    messages = bp(messages)
    errs = []
    for grp in groups:
        # --- gate batch: pieces run where the data lives ---------------------------------------------------------
        grams, projs = {}, {}
        for (a, b) in grp:
            for (v, w) in ((a, b), (b, a)):
                if owner[g.pos[v]] != rank:
                    continue
                env = [(g.leg(v, k), messages[(k, v)]) for k in g.nbrs[v] if k != w]
                grams[(v, w)], projs[(v, w)] = gr.site_gram(tensors[v], g.leg(v, w), env, 10 * np.finfo(np.float64).eps)
        allg = {}
        for part in allgather_obj(grams):                       # exchange #1: Gram matrices
            allg.update(part)
        recs = {}
        for (a, b) in grp:
            if owner[g.pos[a]] != rank:
                continue                                        # owner of the first vertex runs theta / SVD / truncation
            chi = dims[frozenset((a, b))]
            n, err, sv_, x1, x2 = gr.gate_algebra(gate, allg[(a, b)], allg[(b, a)], 2, 2, chi, kw["maxdim"], kw["cutoff"], True, np.float64)
            recs[(a, b)] = dict(n=n, err=err, S=sv_, x2=x2, x1=x1 if owner[g.pos[a]] == rank else None)
        allr = {}
        for part in allgather_obj({k: {kk: vv for kk, vv in v.items() if kk != "x1"} for k, v in recs.items()}):   # exchange #2
            allr.update(part)
        for (a, b) in grp:
            r = allr[(a, b)]
            if owner[g.pos[a]] == rank:
                tensors[a] = gr.site_apply(tensors[a], g.leg(a, b), projs[(a, b)], recs[(a, b)]["x1"], True)
            if owner[g.pos[b]] == rank:
                tensors[b] = gr.site_apply(tensors[b], g.leg(b, a), projs[(b, a)], r["x2"], True)
            dims[frozenset((a, b))] = r["n"]
            messages[(a, b)] = np.diag(r["S"]).astype(dtype)
            messages[(b, a)] = np.diag(r["S"]).astype(dtype)
            errs.append(r["err"])
        messages = bp(messages)
    # <Z> of owned vertices
    zop = np.diag([1.0, -1.0]).astype(complex)
    ez = {}
    for v in g.vertices:
        if owner[g.pos[v]] == rank:
            loc = o.BeliefPropagationCache(o.TensorNetworkState.__new__(o.TensorNetworkState), messages, edge_sequence=[])
            loc.tns.g, loc.tns.tensors = g, tensors
            ez[v] = o.expect_1site(loc, zop, v)
    allz = {}
    for part in allgather_obj(ez):
        allz.update(part)
    if rank == 0:
        bpkw = dict(edge_sequence=seq, maxiter=nsweeps, tolerance=None)
        oc = o.update(o.BeliefPropagationCache(psi), **bpkw)
        oerrs = []
        for grp in groups:
            for (a, b) in grp:
                oerrs.append(o.apply_gate(oc, gate, [a, b], **kw))
            oc = o.update(oc, **bpkw)
        ref = np.array([o.expect_1site(oc, zop, v) for v in g.vertices])
        got = np.array([allz[v] for v in g.vertices])
        with open(out, "wb") as f:
            pickle.dump(dict(errs=np.array(errs), oerrs=np.array(oerrs), ez=got, oez=ref,
                             dims=[dims[frozenset(e)] for e in g.edges], odims=[oc.tns.bond_dim(*e) for e in g.edges],
                             owner=owner), f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
