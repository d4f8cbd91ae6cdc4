// engine.hpp -- host-side engine behind the C ABI: state container (TensorNetworkState + BeliefPropagationCache),
// BP update driver, apply_gates scheduler, truncate, parity probes (implemented in engine_*.cpp, see engine_internal.hpp).
// Threading: a handle AND ALL ITS COPIES (tnqs_copy shares the buffer pool, the profiler and the graph's cached BP plan) must be used
// from one thread at a time; different handle families are independent.  Mirrors (paths relative to the reference):
//   src/MessagePassing/beliefpropagationcache.jl:9-15   struct BeliefPropagationCache {network, messages, edge_sequence}
//   src/TensorNetworks/tensornetworkstate.jl:12-15       TensorNetworkState
//   src/Apply/apply_gates.jl:46-143                      apply_gates / apply_gate!
//   src/MessagePassing/abstractbeliefpropagationcache.jl:223-259   update
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstdint>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/tnqs.h"

namespace tnqs {

struct Err : std::runtime_error {
    int code;
    Err(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
void hipchk(hipError_t e, const char* what);

// ---- caching device allocator ---------------------------------------------------------------------------------
// Reuse is stream-ordered: a buffer that goes back to the free list may be handed out again at once, which is safe while all work of a
// handle is enqueued on ONE stream.  While the boundary sites of a BP level run on the side stream (engine_bp.cpp, fork / join), releases are
// DEFERRED instead: nothing freed inside that region is handed out before the join has ordered both streams again.
class Pool {
public:
    explicit Pool(int device) : device_(device) {}
    ~Pool();
    void* alloc(size_t bytes, size_t* rounded);
    void release(void* p, size_t rounded);
    size_t bytes_live() const { return live_; }
    size_t bytes_cached() const { return cached_; }
    void trim();
    void set_defer(bool on);         // off: everything released meanwhile becomes available
private:
    int device_ [[maybe_unused]];
    std::mutex mu_;
    std::map<size_t, std::vector<void*>> free_;
    std::vector<std::pair<void*, size_t>> deferred_; bool defer_ = false;
    size_t live_ = 0, cached_ = 0;
};
struct DevBuf {
    void* p = nullptr; size_t bytes = 0; size_t rounded = 0; std::shared_ptr<Pool> pool;
    std::shared_ptr<DevBuf> parent;      // set for a view into another buffer (no pool: nothing to release)
    ~DevBuf() { if (p && pool) pool->release(p, rounded); }
};
using Buf = std::shared_ptr<DevBuf>;
inline Buf sub_buffer(const Buf& parent, size_t off, size_t bytes) {
    Buf b = std::make_shared<DevBuf>(); b->p = reinterpret_cast<char*>(parent->p) + off; b->bytes = bytes; b->parent = parent; return b;
}

struct Graph {
    int nv = 0, ne = 0;
    std::vector<int> esrc, edst;
    std::vector<std::vector<int>> nbr, nbr_e;     // neighbours of v in ascending id, and the edge to each
    std::unordered_map<uint64_t, int> emap;
    bool is_tree = false;
    std::vector<int> ecolor; int ncolors = 0;     // deterministic greedy proper edge colouring
    mutable std::vector<int> default_seq;         // default BP sweep order (engine.cpp default_sequence), built on first use
    mutable std::vector<int> default_set_starts;  // positions of default_seq where a set that may close cycles begins (empty: linear forests): a set's levels follow the previous set's
    mutable std::shared_ptr<const void> default_plan;   // its level schedule (engine.cpp BPPlan), built on first use
    mutable std::shared_ptr<const void> forest_plan;    // level schedule of the forest-cover order (n_sequence = -1), built on first use
    mutable int spec_penalty = 0;                       // apply_gates: steps to run one step deep after a failed deferred verification (engine_gates.cpp); shared by the copies of a handle
    int edge(int u, int v) const;                 // -1 if absent
    int leg(int v, int w) const;                  // position of neighbour w in nbr[v], -1 if absent
    int dedge(int src, int dst) const;            // directed edge id 2*e + (src == edst[e]), -1 if absent
};

struct ProfClass { int64_t launches = 0; double ms = 0, bytes = 0, flops = 0; };
struct Prof {
    bool on = false; ProfClass cls[TNQS_PROF_NCLASSES];
    struct Pending { int cls; hipEvent_t a, b; bool own_a; };
    std::vector<Pending> pending; std::vector<hipEvent_t> ev_free;
    // A scope that directly follows another one (no host synchronisation in between) starts at the previous scope's end event instead of
    // recording its own: half the events in the launch chains.  `chain` is cleared wherever the host waits for the stream.
    hipEvent_t last_b = nullptr; bool chain = false; hipStream_t last_stream = nullptr;
    ~Prof() { for (auto& p : pending) { if (p.own_a) (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); } for (auto e : ev_free) (void)hipEventDestroy(e); }
};

// RCCL transport of the exchange step (sharding.cpp): communicator + the exchange buffer the library owns in that mode; shared by the
// copies of a handle, released with the last of them
struct RcclComm {
    void* comm = nullptr; int device = 0; void* exch_owned = nullptr;
    long long n_exchanges = 0, bytes_exchanged = 0;
    ~RcclComm();
};

struct HostArena {   // pinned staging for descriptor uploads, reset at host sync points
    char* base = nullptr; size_t cap = 0, off = 0;
    // pinned ring behind the arena proper: the staged read-backs of pending checks (State::checks) -- they outlive the arena's own resets
    char* ring = nullptr; size_t ring_cap = 0, ring_off = 0;
    hipEvent_t cev[16] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; unsigned cevn = 0;
};

// ---- deferred verification (round 6) ---------------------------------------------------------------------------------------------------
// apply_gates runs ahead of the device: a gate batch whose outcome is predictable (every bond already at its cap: the new bond dimension is the cap, no
// factorisation falls back) and a BP update that is expected to converge in its first sweep are ENQUEUED on those assumptions -- no host round trip in the
// dependent launch chains -- and leave a Check behind: the staged copy of what decides the assumption, an event behind that copy, and the decision.  Site tensors
// and messages are never mutated in place, so the state before any step is a vector of references (Snapshot); a check that fails (rare: a cutoff that bites at a
// saturated bond, a collapsed pivot, a sweep that misses the tolerance) puts the snapshot back, drains the stream and runs the step again the careful way.
// Checks are settled in order; every host synchronisation point of the path settles what is pending, and apply_gates settles everything before it returns.
struct State;
struct Snapshot {
    std::vector<int> chi; std::vector<Buf> site, sscale, msg; std::vector<std::vector<double>> pend1; std::vector<char> unit_norm;
    tnqs_apply_stats stats{}; bool real_io = false;
};
struct Check {
    int kind = 0;                          // 0: gate batch enqueued on assumptions; 1: BP update whose verdict is pending
    int step = 0;                          // step of the apply_gates schedule it belongs to
    int iters_done = 0;                    // BP: sweeps enqueued so far
    hipEvent_t ev = nullptr;               // recorded behind the staged copy (HostArena::cev, not owned)
    std::function<bool(State*)> eval;      // true: the assumption held (results and statistics booked); false: roll back
};
struct SpecFailed { int kind, step, iters_done; };       // thrown by settle() for the first check that does not hold

struct State {
    std::shared_ptr<Graph> g;
    int dtype = TNQS_C64;          // STORAGE / arithmetic type: TNQS_C64 or TNQS_C128 (real element types are stored as complex numbers
                                   // with zero imaginary parts and computed by the complex kernels)
    // element type at the boundary (tnqs_scalartype): a handle created as TNQS_F32 / TNQS_F64 takes and returns REAL tensors and messages
    // until a complex gate is applied to it -- from then on it is ComplexF32 / ComplexF64, exactly like the reference's network after
    // `adapt_gate` kept a complex gate complex (src/Apply/apply_gates.jl:41-44) and the contraction promoted the site tensors
    bool real_io = false;
    int device = 0;
    std::vector<int> d;            // site dims
    std::vector<int> chi;          // bond dim per edge
    std::vector<Buf> site;         // canonical layout; null when not owned (sharding)
    std::vector<Buf> sscale;       // pending real scale factor of a site tensor (one device double; null = 1): the tensor the
                                   // reference holds is site[v] * (*sscale[v]).  Normalisation after a gate only records the factor;
                                   // every consumer on the hot path is scale-invariant, the others call materialize_scale() first
    // pending one-site gate of a vertex (d x d complex128, column-major [s' + d s]; empty = none): the tensor the reference holds is
    // (G applied to the site leg of) site[v] * sscale[v].  A UNITARY one-site gate is only recorded: BP messages sum over the site index of
    // ket and bra, so they do not see it; the next two-site gate on the vertex absorbs it exactly (g . (G1 (x) G2) is the gate simple_update
    // then applies: same theta); everything that reads the tensor itself materialises it first (materialize_pending).  Host-side metadata,
    // identical on every rank of a sharded handle.  unit_norm[v]: the last operation on v left ||psi_v|| = 1 (normalize_tensors), so a
    // deferred unitary gate with normalize_tensors has nothing to normalise
    std::vector<std::vector<double>> pend1; std::vector<char> unit_norm;
    // A sharded handle defers only INSIDE one apply_gates call (every rank walks the same gate list, so the pending sets stay identical) and
    // applies what is left before the call returns: between calls an accessor of one rank only touches the vertices that rank owns
    bool in_apply = false;
    std::vector<Buf> msg;          // 2*ne, null = unset = identity (tensornetworkstate.jl:72-75)
    std::shared_ptr<Pool> pool;
    hipStream_t stream = nullptr; bool own_stream = false;
    // side stream (+ the two events that order it against `stream`): the early small-SVD launches of a gate batch (engine_gates.cpp 2b) and the
    // boundary sites of a BP level (engine_bp.cpp) run on it next to the tensor passes of the main stream; created on first use, owned by this State
    hipStream_t aux_stream = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // sharding
    int rank = 0, nranks = 1; std::vector<int> owner; tnqs_allgather_fn ag_fn = nullptr; void* ag_ctx = nullptr;
    void* exch = nullptr; size_t exch_bytes = 0;      // device exchange buffer (nranks equal blocks): the host's (callback mode) or comm->exch_owned
    std::shared_ptr<RcclComm> comm;                   // set: the all-gather is an ncclAllGather enqueued on the handle's stream
    // profiling (shared by copies of a handle, so a loop `bpc = apply_gates(layer, bpc)` accumulates)
    std::shared_ptr<Prof> prof;
    std::vector<Buf> keepalive;    // descriptor buffers and workspaces kept until the next host sync
    size_t keep_mark = 0;          // keepalive[0, keep_mark) belongs to phases that have ended: released at the next stream synchronisation (soft_sync)
    HostArena arena;               // this handle's pinned staging arena (taken from / returned to a small free list, engine_core.cpp)
    tnqs_apply_stats stats{};
    std::deque<Check> checks;          // pending, oldest first (see Check above)
    int cur_step = -1;                 // apply_gates: the schedule step being executed (labels the checks it leaves behind)

    size_t esz() const { return dtype == TNQS_C64 ? 8 : 16; }
    int scalartype() const { return real_io ? (dtype == TNQS_C64 ? TNQS_F32 : TNQS_F64) : dtype; }
    size_t io_esz() const { return real_io ? esz() / 2 : esz(); }
    bool owns(int v) const { return nranks == 1 || owner[v] == rank; }
    // the sharded code path: more than one rank -- or ONE rank made to take it (TNQS_FORCE_EXCHANGE=1 when tnqs_set_sharding_rccl is called): every exchange point packs
    // its block, runs the communicator's all-gather on the handle's stream and reads the gathered block back, which is all a single GPU can exercise of the RCCL transport
    bool force_exchange = false;
    bool msg_hermitian = true;                       // no message handed in through set_message was non-Hermitian (engine_bp.cpp: products absorbed on the bra side)
    bool sharded() const { return nranks > 1 || force_exchange; }
    ~State();
};

// ---- operations (implemented in engine.cpp, templated internally on the real type) ---------------------------
State* state_create(int nv, int ne, const int32_t* es, const int32_t* ed, const int32_t* sd, int dtype, int device);
State* state_copy(const State* s);
void state_set_site(State* s, int v, const void* host, int ndim, const int64_t* dims, const int32_t* role);
void state_get_site(State* s, int v, void* host, int ndim, const int32_t* role);
void state_set_site_random(State* s, int v, int nn, const int64_t* bond_dims, uint64_t seed, double scale);
int64_t state_site_size(const State* s, int v);
void state_set_message(State* s, int src, int dst, const void* host, int chi);
void state_get_message(State* s, int src, int dst, void* host, int chi);
void bp_update(State* s, const tnqs_bp_opts* opts, int* niter, double* diff);
void apply_gates(State* s, int ngates, const int32_t* nverts, const int32_t* verts, const double* mats,
                 const tnqs_apply_opts* opts, const tnqs_bp_opts* bp, double* errs);
void truncate_bp(State* s, int maxdim, double cutoff, int normalize, int ngroups, const int32_t* offs,
                 const int32_t* eu, const int32_t* ev, const tnqs_bp_opts* bp);
void rdm_1site(State* s, int v, double* out);
void expect_all(State* s, const double* ops, double* out);
void expect_region(State* s, int nr, const int32_t* rv, const int32_t* parent, const double* ops, double* out4);
void vertex_scalars(State* s, double* out);
void edge_scalars(State* s, double* out);
void rescale(State* s);
void rescale_messages(State* s, int n, const int32_t* eu, const int32_t* ev);     // null lists: all edges / all vertices
void rescale_vertices(State* s, int n, const int32_t* verts);
void symmetric_gauge(State* s, double regularization);
void prof_collect(State* s);
void materialize_pending_all(State* s);      // apply every deferred one-site gate (State::pend1)
// sharding.cpp
void rccl_unique_id(void* out128);
void set_sharding_rccl(State* s, int rank, int nranks, const int32_t* owner, const void* unique_id128, int64_t exch_bytes);
void rccl_allgather(State* s, size_t bytes_per_rank);
void rccl_selftest(int device, int64_t bytes);
void rccl_preflight();

}  // namespace tnqs
