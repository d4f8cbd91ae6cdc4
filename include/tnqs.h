/*
 * tnqs.h -- C ABI of libtnqs_hip.so: MI355X (gfx950) implementation of the BP-gauged
 * gate-application hot path of TensorNetworkQuantumSimulator.jl.
 *
 * The reference has no FFI seam (it is pure Julia); these entry points are what a
 * Julia shim type `HipBeliefPropagationCache <: AbstractBeliefPropagationCache` would `ccall`
 * (INTEGRATION.md shows the shim).  Each entry cites the reference function it replaces
 * (paths relative to the reference repository root).
 *
 * Conventions
 *   - every function returns an int status (0 = TNQS_OK, <0 = error); the message is available from
 *     tnqs_last_error() (thread-local).  No C++ exception crosses the ABI.
 *   - vertices are 0..nv-1, undirected edges are 0..ne-1 in the order given to tnqs_create; a directed
 *     edge is named by (src vertex, dst vertex).
 *   - host pointers are borrowed for the duration of the call only.
 *   - complex numbers are interleaved (re, im); matrices/tensors are column-major (Julia order).
 *   - canonical device layout of the site tensor of v:  [site d][leg to nbr_0]...[leg to nbr_{z-1}]
 *     column-major, neighbours in ascending vertex id.  Messages are chi x chi, index order (ket, bra).
 *   - one handle must not be used from two threads at once; every call is synchronous on return.
 */
#ifndef TNQS_H
#define TNQS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tnqs_state_s* tnqs_handle;

/* Element types (the reference's default is Float64, README.md:86; its BP tests run all four, test/test_beliefpropagation.jl:14,34).
 * A handle created as TNQS_F32 / TNQS_F64 takes and returns REAL site tensors and messages (tnqs_set/get_site_tensor,
 * tnqs_set/get_message); the device stores them as complex numbers with zero imaginary parts and runs the complex kernels.  Applying
 * a gate with a non-zero imaginary part promotes the handle to the complex type of the same precision, as in the reference
 * (`adapt_gate`, src/Apply/apply_gates.jl:41-44: a complex gate stays complex, and contracting it promotes the site tensors);
 * tnqs_scalartype tells which type the getters / setters currently speak. */
enum { TNQS_C64 = 0, TNQS_C128 = 1, TNQS_F32 = 2, TNQS_F64 = 3 };

enum {
    TNQS_OK = 0,
    TNQS_ERR_INVALID = -1,     /* bad argument; mirrors `error(...)` / ArgumentError in the reference */
    TNQS_ERR_UNSUPPORTED = -2,
    TNQS_ERR_HIP = -3,         /* HIP runtime failure (includes "no GPU") */
    TNQS_ERR_NUMERIC = -4,     /* e.g. sqrt of a negative message eigenvalue (Julia DomainError, src/utils.jl:21) */
    TNQS_ERR_COMM = -5
};

/* BP update options: kwargs of `update(bpc; maxiter, tolerance, edge_sequence, ...)`
 * (src/MessagePassing/abstractbeliefpropagationcache.jl:223-259, beliefpropagationcache.jl:51-72,103-117). */
typedef struct {
    int maxiter;          /* <= 0: reference default (25 on loopy graphs, 1 on trees) */
    double tolerance;     /* < 0: none (no convergence check) -- what `update(bpc; maxiter = 10)` means in the reference, whose
                           * default_tolerance(::Algorithm"bp") is `nothing`;  NaN: the tolerance of default_bp_update_kwargs
                           * (1e-5 f32 / c64, 1e-8 f64 / c128, none on trees), i.e. what apply_gates / truncate / normalize use when
                           * bp_update_kwargs is omitted.  A NULL opts pointer = default_bp_update_kwargs altogether */
    int normalize;        /* message_update_alg "contract" kwarg `normalize` (default true): m <- m / sum(m) */
    int n_sequence;       /* 0: library default edge sequence (linear forests -- on periodic lattices edge sets that close cycles --, DESIGN.md 4.3: the order the plane kernels share products on);
                           * -1: the reference's default, forest_cover_edge_sequence(graph) (beliefpropagationcache.jl:28), built by the library;
                           * > 0: the explicit sequence below */
    const int32_t* seq_src; /* explicit `edge_sequence` kwarg: directed edges, swept sequentially (Gauss-Seidel) */
    const int32_t* seq_dst;
} tnqs_bp_opts;

/* apply_kwargs of apply_gates / simple_update (src/Apply/simple_update.jl:21-24, forwarded to factorize_svd :58). */
typedef struct {
    int maxdim;             /* <= 0: no cap (`maxdim = nothing`) */
    double cutoff;          /* < 0: `cutoff = nothing` (behaves as 0: only exactly-zero weight is cut) */
    int normalize_tensors;  /* default true */
    double sqrt_cutoff;     /* < 0: default 10*eps(real(eltype)) (src/Apply/simple_update.jl:32-33) */
    int update_cache;       /* apply_gates kwarg `update_cache` (default true) */
} tnqs_apply_opts;

/* run statistics of the last tnqs_apply_gates / tnqs_truncate call (the reference only prints these when verbose) */
typedef struct {
    int n_bp_updates;       /* number of `update` calls triggered (c+1 for a TFIM layer) */
    int n_bp_sweeps;        /* total sweeps over all of them */
    int n_batches;          /* batched launches of pairwise-disjoint gates */
    int n_two_site;         /* two-site gates applied */
    int bp_not_converged;   /* updates that hit maxiter (reference: @warn, abstract...:245-252) */
    double last_bp_diff;
    int n_chol_fallbacks;   /* gate batches in which at least one site had a numerically rank-deficient Gram matrix (those sites take the eigen path) */
    int n_qr2_sites;        /* ComplexF64 sites that went through the second factorisation pass (ill-conditioned psi~) */
    int n_lowrank_svd;      /* two-site gates whose theta SVD ran on the low-rank factor (gate of operator Schmidt rank kappa, kappa chi < d chi) */
    int n_tall_svd;         /* theta SVDs that went through the Cholesky-QR preprocessing (matrix too tall for the LDS-resident Jacobi: 256 x 128 at chi = 64) */
    int n_svd_sweeps;       /* Jacobi sweeps of the theta SVDs, summed over the two-site gates of the call (diagnostic: sweeps per gate = this / n_two_site) */
    int n_svd_sweeps_max;   /* ... and the largest count of any single gate: a colour batch's SVD launch lasts as long as its slowest gate */
    int n_deferred_1site;   /* unitary one-site gates that were only recorded and later absorbed by a two-site gate on the vertex (or applied when the tensor was read) */
    int n_bp_products_reused;  /* BP: partial products (site tensor x messages of some legs) taken from an earlier level instead of recomputed */
    int n_bp_products_evicted; /* ... dropped again by the per-site bound (3) or the byte bound (TNQS_BP_CACHE_MB) before anything could reuse them or after */
    int n_lowrank_fallbacks; /* gates that qualified for the low-rank theta SVD but whose Cholesky / CholeskyQR2 of B refused a pivot: they took the SVD of the full theta */
    int n_spec_batches;     /* (library version 101) gate batches enqueued without their host round trip: every bond already at its cap, outcome verified afterwards */
    int n_spec_redone;      /* ... steps (gate batches, BP updates) whose deferred verification failed: state put back, step run again the careful way */
} tnqs_apply_stats;

/* ---- library ---------------------------------------------------------------------------------------- */
int tnqs_version(void);
const char* tnqs_last_error(void);
int tnqs_device_count(int* count);

/* ---- state handle: TensorNetworkState + BeliefPropagationCache(psi)
 *      (src/TensorNetworks/tensornetworkstate.jl:12-15, src/MessagePassing/beliefpropagationcache.jl:9-15,27-31).
 *      Created as the all-"up" product state with bond dimension 1 and unset (= identity) messages
 *      (tensornetworkstate.jl:72-75,141-161). */
int tnqs_create(int nv, int ne, const int32_t* edge_src, const int32_t* edge_dst, const int32_t* site_dim,
                int dtype, int device, tnqs_handle* out);
int tnqs_destroy(tnqs_handle h);
/* Base.copy(::BeliefPropagationCache) (beliefpropagationcache.jl:35-37): shallow, buffers are shared and
 * never mutated in place, so the copy is O(nv + ne). */
int tnqs_copy(tnqs_handle h, tnqs_handle* out);
/* scalartype(bpc) (abstractbeliefpropagationcache.jl:13): the element type host buffers of this handle are read / written in */
int tnqs_scalartype(tnqs_handle h, int* dtype);
/* use an existing HIP stream (e.g. torch's current stream) instead of the handle's own */
int tnqs_set_stream(tnqs_handle h, void* hip_stream);

/* ---- tensors / messages  (setindex_preserve! abstracttensornetwork.jl:40-43; psi[v]; setmessage!/message
 *      abstractbeliefpropagationcache.jl:93-102) ------------------------------------------------------------ */
/* leg_role[k] = -1 for the site leg, otherwise the neighbour vertex that leg k connects to.  The library
 * permutes into the canonical layout bit-exactly.  Bond dimensions of the incident edges are updated; messages
 * on an edge whose dimension changes are reset to the identity. */
int tnqs_set_site_tensor(tnqs_handle h, int v, const void* host, int ndim, const int64_t* dims, const int32_t* leg_role);
int tnqs_get_site_tensor(tnqs_handle h, int v, void* host, int ndim, const int32_t* leg_role);
/* Synthetic site tensor generated ON THE DEVICE (the benchmark states of the 8-GPU configurations are ~250 GiB: host random numbers plus PCIe
 * would take minutes): iid complex-normal entries (real-normal for a real handle), scaled by `scale`, in the canonical layout; bond_dims[j] =
 * dimension of the leg to the j-th neighbour in ascending vertex order.  Counter-based: entry e of vertex v depends on (seed, v, e) only, so
 * a sharded and an unsharded run build the same state.  Sharded handles: a vertex owned by another rank only records the dimensions. */
int tnqs_set_site_random(tnqs_handle h, int v, int n_neighbours, const int64_t* bond_dims, uint64_t seed, double scale);
int tnqs_site_tensor_size(tnqs_handle h, int v, int64_t* nelem);
/* chi x chi message src -> dst, axes (ket, bra), column major.  Every message the path itself produces is Hermitian to rounding, and tnqs_update absorbs some of them on
 * the bra side as their own conjugate transpose (csrc/engine_bp.cpp); a message handed in here that is NOT Hermitian (1e-5 / 1e-12 of its largest entry) makes the
 * handle (and its copies) absorb everything on the ket side, as abstractbeliefpropagationcache.jl:162-190 does. */
int tnqs_set_message(tnqs_handle h, int src, int dst, const void* host, int chi);
int tnqs_get_message(tnqs_handle h, int src, int dst, void* host, int chi);
int tnqs_bond_dim(tnqs_handle h, int u, int v, int* chi);           /* dim(virtualinds) */
int tnqs_maxvirtualdim(tnqs_handle h, int* chi);                     /* abstracttensornetwork.jl:27-29 */

/* ---- BP update (abstractbeliefpropagationcache.jl:223-259) -------------------------------------------------
 * Non-convergence is not an error (the reference only warns): status OK, *niter == maxiter, diff returned. */
int tnqs_bp_update(tnqs_handle h, const tnqs_bp_opts* opts, int* niter, double* final_diff);

/* ---- apply_gates (src/Apply/apply_gates.jl:46-98 incl. the update scheduling rule :68-90 and apply_gate!
 *      :101-143, simple_update src/Apply/simple_update.jl:21-77).  Mutates h (callers wanting the reference's
 *      value semantics call tnqs_copy first).
 *      gate_nverts[g] in {1,2}; gate_verts is the concatenation of the vertex lists; gate_mats is the
 *      concatenation of the d^k x d^k complex128 matrices, column-major, first listed vertex = most significant
 *      index.  out_truncerr is caller-allocated double[ngates] (apply_gates.jl:61). */
int tnqs_apply_gates(tnqs_handle h, int ngates, const int32_t* gate_nverts, const int32_t* gate_verts,
                     const double* gate_mats, const tnqs_apply_opts* opts, const tnqs_bp_opts* bp_opts,
                     double* out_truncerr, tnqs_apply_stats* stats);

/* ---- truncate (src/truncate.jl:12-38, edge_color = true branch).  Edge colour groups are supplied by the
 *      caller (the reference gets them from SimpleGraphAlgorithms.edge_color); n_groups = 0 selects the
 *      non-coloured branch (:31-35): one update after every edge in edge order. */
int tnqs_truncate(tnqs_handle h, int maxdim, double cutoff, int normalize_tensors, int n_groups,
                  const int32_t* group_offsets, const int32_t* edge_u, const int32_t* edge_v,
                  const tnqs_bp_opts* bp_opts, tnqs_apply_stats* stats);

/* ---- parity probes (src/expect.jl:59-82) ------------------------------------------------------------------
 * rho[s,s'] = sum psi[s,l..] conj psi[s',l'..] prod m[l,l'] (un-normalised, d x d complex128 col-major);
 * expect = tr(op rho)/tr(rho), op[s',s] column-major complex128. */
int tnqs_rdm_1site(tnqs_handle h, int v, double* out_rho);
int tnqs_expect_1site(tnqs_handle h, int v, const double* op, double* out_re_im);
/* all-vertex <op_v>: ops is nv consecutive d x d matrices; out is nv complex128 */
int tnqs_expect_all(tnqs_handle h, const double* ops, double* out_re_im);

/* ---- multi-site observables (src/expect.jl:59-82) ------------------------------------------
 * The caller passes the region = the vertices of the Steiner tree of the observable's support (expect.jl:68), as a rooted
 * tree: region_parent[i] = index (into region_verts) of the parent of vertex i, -1 for the single root.  What is contracted is the
 * INDUCED region, as in the reference (norm_factors over the Steiner vertices share every internal bond): a bond between two region
 * vertices that is not a tree edge (a plaquette's closing bond) is summed over as well -- chi^2 tree contractions per such bond, at
 * most 2^16 terms in all (TNQS_ERR_UNSUPPORTED beyond: one closing bond up to chi = 256, two up to chi = 16).  ops = one d x d complex128 column-major matrix op[s',s] per region vertex
 * (identity off the support).  out = {Re, Im numerator, Re, Im denominator}; <O> = coeff * numer / denom. */
int tnqs_expect_region(tnqs_handle h, int n_region, const int32_t* region_verts, const int32_t* region_parent,
                       const double* ops, double* out_numer_denom);

/* ---- BP scalars and normalisation --------------------------------------------------------
 * vertex scalar  tr(rho_v)  = [psi_v, conj psi_v, incoming messages] contracted (abstractbeliefpropagationcache.jl:22-28),
 * edge scalar    sum_ij m_e[i,j] m_rev(e)[i,j]                                    (beliefpropagationcache.jl:47-49);
 * free energy = sum log(vertex scalars) - sum log(edge scalars) (abstract...:289-304) is left to the host.
 * out_vertex: nv complex128 (NaN for vertices owned by another rank); out_edge: ne complex128, edges in tnqs_create order.
 * tnqs_rescale: rescale_messages! then rescale_vertices! (beliefpropagationcache.jl:82-140, abstract...:318-322) in place:
 * afterwards every vertex and edge scalar is 1 (the BP norm of the state is 1). */
int tnqs_vertex_scalars(tnqs_handle h, double* out_vertex);
int tnqs_edge_scalars(tnqs_handle h, double* out_edge);
int tnqs_rescale(tnqs_handle h);
/* the two halves on their own, as the abstract cache interface has them (abstractbeliefpropagationcache.jl:11-20,306-316): generic callers
 * invoke either one.  tnqs_rescale_messages: rescale_messages!(bpc, edges) (beliefpropagationcache.jl:127-140) on the listed edges
 * (edge_u[i], edge_v[i]), both directions each; edge_u == NULL: every edge.  tnqs_rescale_vertices: rescale_vertices!(bpc, vertices)
 * (beliefpropagationcache.jl:82-101) with the vertex scalars under the CURRENT messages; vertices == NULL: every vertex. */
int tnqs_rescale_messages(tnqs_handle h, int n_edges, const int32_t* edge_u, const int32_t* edge_v);
int tnqs_rescale_vertices(tnqs_handle h, int n_vertices, const int32_t* vertices);

/* ---- symmetric (Vidal) gauge (src/symmetric_gauge.jl:1-62) ---------------------------------
 * per edge: psi_src <- psi_src X^-1/2 U S^1/2, psi_dst <- psi_dst Y^-1/2 V S^1/2 with U S V = svd(X^1/2 (Y^1/2)^T), X, Y the two
 * messages of the edge (eigenvalues + regularization before the roots); both messages := diag(S).  The state is unchanged and
 * diag(S) is a BP fixed point when the input messages were one.  regularization < 0: default 10 eps(real(eltype)). */
int tnqs_symmetric_gauge(tnqs_handle h, double regularization);

/* ---- multi-GPU sharding (no reference analogue).  A rank owns a vertex subset: it holds only
 *      those site tensors and does all per-vertex work for them; messages are replicated.  The library calls
 *      the host-supplied all-gather at the exchange points (host side: torch.distributed over RCCL). --------- */
/* all-gather over the ranks of `bytes_per_rank` bytes: rank r's block sits at exch_base + r*bytes_per_rank of the
 * exchange buffer handed to tnqs_set_sharding (in place).  Must be complete (device-visible) on return. */
typedef int (*tnqs_allgather_fn)(void* ctx, void* exch_base, int64_t bytes_per_rank, int nranks);
/* vertex_owner[v] in [0, nranks).  exch_dev/exch_bytes: a device buffer owned by the host side (e.g. a torch tensor)
 * that the library packs its exchange payloads into.  Call right after tnqs_create on every rank; site tensors of
 * vertices owned by other ranks are dropped, and tnqs_set_site_tensor on them only records the bond dimensions
 * (host may be NULL).  Exchanges per apply_gates call: one per BP level (raw messages, <= 2|E| chi^2 elements per
 * sweep in total) and two per gate batch (the d*chi x d*chi Gram matrices; chi', truncation error, S and X2). */
int tnqs_set_sharding(tnqs_handle h, int rank, int nranks, const int32_t* vertex_owner, tnqs_allgather_fn fn, void* ctx,
                      void* exch_dev, int64_t exch_bytes);

/* The same sharding with the transport INSIDE the library: RCCL (librccl.so, loaded at run time) over xGMI.  One rank calls
 * tnqs_rccl_unique_id and hands the 128 opaque bytes (an ncclUniqueId) to every rank by whatever means the host has (MPI, a file,
 * torch.distributed's store); every rank then calls tnqs_set_sharding_rccl right after tnqs_create -- it joins the communicator
 * (collective: blocks until all nranks ranks have called) and allocates the exchange buffer (exch_bytes, on the handle's device).
 * From then on every exchange point is ONE in-place ncclAllGather enqueued on the handle's stream: no stream synchronisation, no
 * host callback.  Copies of the handle share the communicator; it is destroyed with the last of them.  nranks == 1 is allowed
 * (nothing is exchanged). */
int tnqs_rccl_unique_id(void* out_128_bytes);
int tnqs_set_sharding_rccl(tnqs_handle h, int rank, int nranks, const int32_t* vertex_owner, const void* unique_id_128_bytes,
                           int64_t exch_bytes);
/* all-gathers issued through RCCL by this handle and its copies, and the bytes they gathered (0 in callback mode) */
int tnqs_sharding_stats(tnqs_handle h, int64_t* n_exchanges, int64_t* bytes_exchanged);
/* one-rank round trip through RCCL on `device` (unique id, communicator, in-place all-gather of `bytes` bytes, teardown): lets a
 * single-GPU box check that the transport loads and runs -- RCCL refuses two ranks on one GPU */
int tnqs_rccl_selftest(int device, int64_t bytes);
/* LOCAL preflight, no collective: librccl.so loads and exports every entry point the transport uses.  A host that is about to call
 * tnqs_set_sharding_rccl on N ranks runs this (or tnqs_rccl_selftest) on every rank first and lets the ranks AGREE on the outcome over
 * a channel it already trusts, so that no rank enters the communicator set-up while another one has already failed. */
int tnqs_rccl_preflight(void);

/* ---- profiling: HIP-event timing of the kernel classes on the handle's stream ---------------------------- */
enum { TNQS_PROF_BP_MODEPROD = 0, TNQS_PROF_BP_GRAM = 1, TNQS_PROF_GATE_MODEPROD = 2, TNQS_PROF_GATE_GRAM = 3,
       TNQS_PROF_GATE_APPLY = 4, TNQS_PROF_JACOBI = 5, TNQS_PROF_SMALL = 6, TNQS_PROF_BP_FUSED = 7, TNQS_PROF_BP_PAIR = 8, TNQS_PROF_BP_PAIRGRAM = 9,
       /* whole phases, first to last kernel on the handle's stream (side streams join it before a phase ends): the CRITICAL-PATH time of the BP updates (launches =
        * sweeps) and of the batches of two-site gates (launches = batches) -- the kernel classes above overlap each other where a phase runs on two streams */
       TNQS_PROF_PHASE_BP_UPDATE = 10, TNQS_PROF_PHASE_GATE_BATCH = 11, TNQS_PROF_NCLASSES = 12 };
int tnqs_profile_enable(tnqs_handle h, int on);
/* launches, total ms, algorithmic bytes (min traffic: operands read once + result written once) and flops */
int tnqs_profile_get(tnqs_handle h, int cls, int64_t* launches, double* total_ms, double* alg_bytes, double* alg_flops);
int tnqs_profile_reset(tnqs_handle h);

#ifdef __cplusplus
}
#endif
#endif /* TNQS_H */
