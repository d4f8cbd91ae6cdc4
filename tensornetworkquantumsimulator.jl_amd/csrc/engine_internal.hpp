// engine_internal.hpp -- declarations shared by the translation units of the host engine:
//   engine_core.cpp   pool, graph, state container, tensor / message I/O
//   engine_batch.cpp  batched building blocks: mode-product chains, Gram jobs, the SVD batch
//   engine_bp.cpp     BP update (default sequence, level schedule, message kernels)
//   engine_gates.cpp  apply_gates scheduler, one- and two-site gate batches, truncate
//   engine_obs.cpp    observables, BP scalars / rescale, symmetric gauge
//   sharding.cpp      exchange step (RCCL or host callback)
#pragma once
#include "engine.hpp"
#include "kernels.hpp"
#include <algorithm>
#include <array>
#include <chrono>
#include <climits>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <numeric>
#include <set>
#include <type_traits>
#include <unordered_map>

namespace tnqs {

#define HIPCHK(x) hipchk((x), #x)

// ---- environment switches (all default off; read once per process).  ONE alternative route per kernel family (round 6: the table had grown to ~45 switches,
// most of them A/B levers of decisions that are long made -- each one a route that had to stay correct): they serve the route-equivalence tests
// (tests/test_gpu_toggles.py) and the A/B legs of bench.py; nothing else steers the hot path. ------------------------------------------------------------
inline bool envflag(const char* name) { const char* e = std::getenv(name); return e && e[0] == '1'; }
#define TNQS_SWITCH(fn, expr) inline bool fn() { static const bool v = (expr); return v; }
TNQS_SWITCH(use_mfma, !envflag("TNQS_NO_MFMA"))                  // no matrix-core kernel at all: the generic tiled kernels (kernels.hip), any dims / element type
TNQS_SWITCH(use_pair, !envflag("TNQS_NO_PAIR"))                  // no plane kernels (two legs per pass, pair-Gram, chi = 16 / 32): single-leg matrix-core products
TNQS_SWITCH(use_bra_products, !envflag("TNQS_NO_BRA_PRODUCTS"))  // BP: every message absorbed on the ket side (engine_bp.cpp: sites of degree >= 6 absorb half of them on the bra side)
TNQS_SWITCH(use_prodcache, !envflag("TNQS_NO_PRODCACHE"))        // BP: no partial product kept from one level to the next (engine_bp.cpp ProdCache)
TNQS_SWITCH(use_chol, !envflag("TNQS_NO_CHOL"))                  // R factor from the eigen factorisation of the Gram matrix instead of Cholesky (the route a collapsed pivot falls back to)
TNQS_SWITCH(use_qr2, !envflag("TNQS_NO_QR2"))                    // ComplexF64: no second factorisation pass
TNQS_SWITCH(use_lowrank, !envflag("TNQS_NO_LOWRANK"))            // theta SVD on the full theta instead of the low-rank factor (the route a refused pivot falls back to)
TNQS_SWITCH(use_precond_svd, !envflag("TNQS_NO_PRECOND_SVD"))    // low-rank theta SVD on the plain LDS Jacobi instead of the preconditioned one-kernel route (kernels.hip theta_svd_pre_kernel)
TNQS_SWITCH(use_small_svd, !envflag("TNQS_NO_SMALLSVD"))         // sites with fewer fibers than columns: Gram + eigen instead of the direct SVD
TNQS_SWITCH(defer_site1, !envflag("TNQS_NO_DEFER_1SITE"))        // unitary one-site gates are applied in a pass of their own instead of being carried to the next two-site gate
TNQS_SWITCH(use_chi64, !envflag("TNQS_NO_CHI64"))                // the chi = 64 kernel family (kernels_chi64.hip: register-direct fiber GEMM, 64 x 64 / 128 x 128 Grams, packed Cholesky, Cholesky-QR theta SVD)
#undef TNQS_SWITCH
// TNQS_NO_BF16X3=1 (launch_util.hpp): the chi = 32 / 64 plane kernels on v_mfma_f32_32x32x2_f32 instead of the bf16 matrix cores with exact three-way operand splits
//                   (kernels_x3.hip; bench.py's A/B leg);
// TNQS_NO_SPECULATION=1 (engine_gates.cpp): apply_gates never runs ahead of the device -- every batch reads its results back, every BP update waits for its verdict;
// TNQS_NO_SMALL_SITE_BP=1 (engine_bp.cpp): sites of at most 8192 elements take the generic chain + Gram route instead of the one-kernel LDS-resident message;
// TNQS_JACOBI_GLOBAL=1: every Jacobi factorisation in the global-memory kernel (the route matrices beyond the LDS take);
// tunables: TNQS_ARENA_KB (pinned staging arena; tests of its overflow path), TNQS_BP_WS_MB (workspace bound of a BP sub-batch), TNQS_BP_CACHE_MB, TNQS_RCCL_LIB (sharding.cpp),
//           TNQS_FORCE_EXCHANGE (a one-rank RCCL handle takes the sharded path); diagnostics: TNQS_HOST_TIMING, TNQS_DEBUG_SWEEPS.
// Kernel experiments are NOT in the shipped library: they only exist in a build with -DTNQS_EXPERIMENTS (csrc/build.sh EXPERIMENTS=1); the kernel-level entry points of
// include/tnqs_debug.h (debug.cpp) read TNQS_DBG_* themselves and are not part of the hot path.
// TNQS_BP_CACHE_MB: bound on the partial products kept across BP levels (MiB, default 49152)
inline size_t bp_cache_budget() { static const size_t v = [] { const char* e = std::getenv("TNQS_BP_CACHE_MB"); return (e ? (size_t)std::atoll(e) : (size_t)49152) << 20; }(); return v; }
inline size_t bp_ws_budget() { static const size_t v = [] { const char* e = std::getenv("TNQS_BP_WS_MB"); return (e ? (size_t)std::atoll(e) : (size_t)24576) << 20; }(); return v; }
inline size_t jacobi_lds(size_t bytes) { static const bool g = envflag("TNQS_JACOBI_GLOBAL"); return (g || bytes > 160 * 1024 - 2048) ? 0 : bytes; }
inline int mmax_of(const std::vector<JacobiItem>& ji) { int m = 1; for (auto& j : ji) m = std::max(m, std::max(j.m, j.n)); return m; }

// optional host-side phase timing (TNQS_HOST_TIMING=1): printed when the process exits
struct HostTimer {
    static double acc[16]; static long cnt[16];
    int k; std::chrono::steady_clock::time_point t0; bool on;
    explicit HostTimer(int kk) : k(kk), t0(std::chrono::steady_clock::now()), on(true) {}
    void stop() { if (on) { static std::mutex mu; std::lock_guard<std::mutex> lk(mu); acc[k] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); cnt[k]++; on = false; } }
    ~HostTimer() { stop(); }
};

void sync(State* s);                                   // stream synchronise + release of the batch's descriptor buffers
hipStream_t aux_stream_of(State* s);                  // the State's side stream (+ its events), created on first use / recycled with the State
void recycle_arena(HostArena ar);                     // back to the free list (nothing on the device may still read it)
HostArena acquire_arena();                             // a recycled or freshly pinned 32 MiB arena (engine_core.cpp)
// The staging arena is about to be reused from its start: everything that may still read it must have run -- on the current stream and on the other
// stream this State enqueues on (the side stream of the early small-SVD launches and of the BP levels' boundary sites)
inline void drain_for_arena_reuse(State* s) {
    HIPCHK(hipStreamSynchronize(s->stream));
    if (s->aux_stream && s->aux_stream != s->stream) HIPCHK(hipStreamSynchronize(s->aux_stream));
}
// Read-back through the pinned staging arena: the copy is enqueued and the staged host pointer returned; it holds the data once the stream
// has been synchronised (a read-back into pageable memory is staged by the runtime and BLOCKS per call).  The staged bytes stay valid until the
// next upload() / readback() after a sync(s) reuses the arena.
template <class X> const X* readback(State* s, const void* dsrc, size_t count) {
    const size_t bytes = count * sizeof(X);
    HostArena& ar = s->arena;
    if (!ar.base) ar = acquire_arena();
    const size_t aligned = (std::max<size_t>(bytes, 1) + 255) & ~size_t(255);
    if (aligned > ar.cap) throw Err(TNQS_ERR_UNSUPPORTED, "read-back too large for the staging arena");
    if (ar.off + aligned > ar.cap) { drain_for_arena_reuse(s); ar.off = 0; }     // pending uploads have been copied: the host side is free
    char* h = ar.base + ar.off; ar.off += aligned;
    if (bytes) HIPCHK(hipMemcpyAsync(h, dsrc, bytes, hipMemcpyDeviceToHost, s->stream));
    return reinterpret_cast<const X*>(h);
}
// room for `bytes` of staged read-backs without a wrap in between (several read-backs that are consumed after one synchronisation)
inline void reserve_readback(State* s, size_t bytes) {
    HostArena& ar = s->arena;
    if (!ar.base) ar = acquire_arena();
    if (bytes > ar.cap) throw Err(TNQS_ERR_UNSUPPORTED, "read-backs of one batch too large for the staging arena");
    if (ar.off + bytes > ar.cap) { drain_for_arena_reuse(s); ar.off = 0; }
}
// End of a phase WITHOUT draining the stream: everything in the keep-alive list so far may go once the stream has been synchronised for some
// other reason (the next read-back), so the host can start preparing the next phase while this one's last kernels still run.
inline void soft_sync(State* s) { s->keep_mark = s->keepalive.size(); }
// after a raw hipStreamSynchronize(s->stream): release what soft_sync() marked (later entries belong to the phase in progress)
inline void drained(State* s) {
    s->prof->chain = false;                    // the host waited: the next profiled scope records its own start event
    s->arena.off = 0;                          // every staged upload has been copied; staged read-backs are consumed by the caller before its next upload
    if (s->keep_mark) { s->keepalive.erase(s->keepalive.begin(), s->keepalive.begin() + (std::ptrdiff_t)std::min(s->keep_mark, s->keepalive.size())); s->keep_mark = 0; }
}
void materialize_scale(State* s, const std::vector<int>& verts);
void materialize_scale_all(State* s);
void materialize_pending(State* s, const std::vector<int>& verts);      // apply the pending one-site gates of these vertices (State::pend1)
void materialize_pending_all(State* s);
inline Buf dalloc(State* s, size_t bytes) {
    auto b = std::make_shared<DevBuf>();
    b->pool = s->pool; b->bytes = bytes;
    b->p = s->pool->alloc(bytes ? bytes : 1, &b->rounded);
    return b;
}
// descriptor upload through the handle's pinned staging arena (reset at host sync points)
template <class Item> const Item* upload(State* s, const std::vector<Item>& v) {
    if (v.empty()) return nullptr;
    size_t bytes = v.size() * sizeof(Item);
    HostArena& ar = s->arena;
    if (!ar.base) ar = acquire_arena();
    size_t aligned = (bytes + 255) & ~size_t(255);
    if (aligned > ar.cap) throw Err(TNQS_ERR_UNSUPPORTED, "descriptor batch too large");
    // arena full: wait for the copies that still read it and start over.  ONLY the host staging is recycled here -- the device buffers of
    // earlier uploads (descriptor arrays whose kernels are not launched yet) stay in the keep-alive list: a sync(s) at this point would hand
    // them back to the pool in the middle of a phase and the next dalloc of the same size class could alias them
    if (ar.off + aligned > ar.cap) { drain_for_arena_reuse(s); ar.off = 0; s->prof->chain = false; }
    char* h = ar.base + ar.off; ar.off += aligned;
    std::memcpy(h, v.data(), bytes);
    Buf b = dalloc(s, bytes);
    HIPCHK(hipMemcpyAsync(b->p, h, bytes, hipMemcpyHostToDevice, s->stream));
    s->keepalive.push_back(b);
    return reinterpret_cast<const Item*>(b->p);
}

// The same for the descriptor arrays of the ONE-WORKGROUP-PER-ITEM kernels (items[blockIdx.x]: env / Cholesky / theta / Jacobi / finish / diag ...):
// no copy at all -- the kernel reads its item straight from the pinned arena (hipHostMalloc memory is mapped into the device's address space under the
// same pointer).  A dependent chain of 20-60 us kernels paid ~5 us of stream time and ~8 us of host time per descriptor copy (two dozen per gate
// batch); a workgroup's single ~100-byte read over the host link costs it 1-2 us.  NOT for kernels that binary-search their item list per workgroup
// (the tensor passes, reduce) or re-read the data in inner loops (gate matrices).  The bytes stay valid until the arena is reset, which only
// happens after the stream has been synchronised (drained / sync / wrap below) -- by then every kernel that reads them has run.  Hence: only for
// arrays whose kernels are ALL launched before the next host synchronisation of the phase (the GateItem array of a batch is read again after the
// read-back of the ranks in the two-trip flow: it keeps its device copy).
template <class Item> const Item* upload_small(State* s, const std::vector<Item>& v) {
    if (v.empty()) return nullptr;
    const size_t bytes = v.size() * sizeof(Item);
    HostArena& ar = s->arena;
    if (!ar.base) ar = acquire_arena();
    const size_t aligned = (bytes + 255) & ~size_t(255);
    if (aligned > ar.cap) throw Err(TNQS_ERR_UNSUPPORTED, "descriptor batch too large");
    if (ar.off + aligned > ar.cap) { drain_for_arena_reuse(s); ar.off = 0; s->prof->chain = false; }
    char* h = ar.base + ar.off; ar.off += aligned;
    std::memcpy(h, v.data(), bytes);
    return reinterpret_cast<const Item*>(h);
}

struct ProfScope {
    State* s; int cls; hipEvent_t a = nullptr, b = nullptr; bool own_a = true;
    ProfScope(State* st, int c, double bytes, double flops) : s(st), cls(c) {
        Prof& P = *s->prof;
        if (!P.on) return;
        P.cls[c].bytes += bytes; P.cls[c].flops += flops; P.cls[c].launches += 1;
        auto get = [&]() { hipEvent_t e; if (!P.ev_free.empty()) { e = P.ev_free.back(); P.ev_free.pop_back(); } else HIPCHK(hipEventCreate(&e)); return e; };
        if (P.chain && P.last_b && P.last_stream == s->stream) { a = P.last_b; own_a = false; }
        else { a = get(); own_a = true; HIPCHK(hipEventRecord(a, s->stream)); }
        b = get();
    }
    ~ProfScope() {
        if (!a) return;
        Prof& P = *s->prof;
        (void)hipEventRecord(b, s->stream);
        P.pending.push_back({cls, a, b, own_a});
        P.last_b = b; P.chain = true; P.last_stream = s->stream;
    }
};

// a whole phase on the handle's stream (TNQS_PROF_PHASE_*): own events, outside the chaining of the kernel-class scopes; `count` is added to the class's launches
struct PhaseScope {
    State* s; int cls; hipEvent_t a = nullptr, b = nullptr; long count = 1;
    PhaseScope(State* st, int c) : s(st), cls(c) {
        Prof& P = *s->prof;
        if (!P.on) return;
        auto get = [&]() { hipEvent_t e; if (!P.ev_free.empty()) { e = P.ev_free.back(); P.ev_free.pop_back(); } else HIPCHK(hipEventCreate(&e)); return e; };
        a = get(); b = get();
        HIPCHK(hipEventRecord(a, s->stream));
    }
    ~PhaseScope() {
        if (!a) return;
        Prof& P = *s->prof;
        (void)hipEventRecord(b, s->stream);
        P.cls[cls].launches += count;
        P.pending.push_back({cls, a, b, true});
    }
};

// bond dimensions of a site's legs without a heap allocation for degrees up to 8 (site_dims is called several times per message and per gate on the host:
// on the latency-bound lattices the allocations were a measurable share of the BP preparation)
struct ChiVec {
    int inl[8]; int n = 0; std::vector<int> big;          // big: only when the degree exceeds 8
    void push_back(int c) { if (n < 8 && big.empty()) inl[n++] = c; else { if (big.empty()) big.assign(inl, inl + n); big.push_back(c); ++n; } }
    int* data() { return big.empty() ? inl : big.data(); }
    const int* data() const { return big.empty() ? inl : big.data(); }
    int& operator[](size_t i) { return data()[i]; }
    const int& operator[](size_t i) const { return data()[i]; }
    size_t size() const { return (size_t)n; }
    bool empty() const { return n == 0; }
    const int* begin() const { return data(); }
    const int* end() const { return data() + n; }
};
struct SD {       // dims of a site tensor in canonical layout
    int z = 0, d = 1; ChiVec chi; size_t n = 1;
    size_t pre(int j) const { size_t p = d; for (int i = 0; i < j; ++i) p *= chi[i]; return p; }
    size_t post(int j) const { size_t p = 1; for (int i = j + 1; i < z; ++i) p *= chi[i]; return p; }
};
inline SD site_dims(const State* s, int v) {
    SD r; const Graph& g = *s->g;
    r.z = (int)g.nbr[v].size(); r.d = s->d[v]; r.n = r.d;
    for (int j = 0; j < r.z; ++j) { int c = s->chi[g.nbr_e[v][j]]; r.chi.push_back(c); r.n *= c; }
    return r;
}

// number of elements of a site tensor (no allocation: site_dims builds a vector)
inline size_t site_nelem(const State* s, int v) { size_t n = (size_t)s->d[v]; for (int e : s->g->nbr_e[v]) n *= (size_t)s->chi[e]; return n; }
inline size_t round256(size_t b) { return (b + 255) & ~size_t(255); }
void exchange(State* s, size_t bytes_per_rank);                       // sharding.cpp
void check_exchange(const State* s, size_t bytes_per_rank);           // call BEFORE enqueuing anything that writes into s->exch

inline int pick_TR(size_t KK, size_t esz, int copies) {
    for (int tr : {64, 32, 16, 8, 4}) if (KK * tr * esz * copies <= 64 * 1024) return tr;
    throw Err(TNQS_ERR_UNSUPPORTED, "bond dimension too large for the fiber-tile kernels (d*chi*16*elemsize must fit 64 KiB of LDS)");
}
inline void tile_params(size_t PA, size_t PB, int TR, int& TA, int& TB, int& nta, int& ntb) {
    TA = (int)std::min<size_t>(PA, TR); TB = std::max(1, TR / TA); TB = (int)std::min<size_t>(TB, PB);
    nta = (int)((PA + TA - 1) / TA); ntb = (int)((PB + TB - 1) / TB);
}


// a chain = one site tensor pushed through several mode products (leg j with matrix X_j, chi_j x chi_j)
struct Chain {
    int v = -1; const void* src = nullptr; SD sd;
    const void* y = nullptr;                     // the untouched site tensor when src is a shared partial product of it (BP prefix sharing)
    std::vector<std::pair<int, const void*>> steps;
    const void* result = nullptr; Buf tmp[2];
    bool ordered = false;                        // steps are in the caller's priority order: the two-leg stages take them from the front
    std::vector<std::vector<int>> trail;         // legs absorbed by each pass, in order; pass k wrote tmp[k & 1] (the last two are intact afterwards)
};


struct GramJob {      // out[i,j] = sum X[i,.] conj(Y[j,.]) over everything but the kept index (s and/or leg)
    const void* X; const void* Y; SD sd; int leg;  /* -1: keep the site index only */ bool keep_site;
    Buf partial; int nchunks = 0; int KK = 0;
    const void* M = nullptr;       // fused path: message absorbed on the first row leg inside the Gram kernel
    Buf final_msg;                 // set: the kernel that computed the message normalised and diffed it too (bp_small_site_kernel, matrix-core form): no msg_finalize item
};

template <class T> void run_chains(State* s, std::vector<Chain>& chains, int cls, int cls_pair = -1);
template <class T, class Acc> void run_grams(State* s, std::vector<GramJob>& jobs, int cls);
template <class T> void svd_batch(State* s, const std::vector<JacobiItem>& all, bool with_v);
// optimistic: (apply_gates, a tolerance given) return after ENQUEUING the first sweep with its verdict left as a Check (engine.hpp); iters_before: sweeps this
// update has already run (the continuation after a verdict turned out negative)
template <class T> void bp_update_t(State* s, const tnqs_bp_opts* o, int* niter_out, double* diff_out, bool optimistic = false, int iters_before = 0);

// ---- deferred verification (engine.hpp: Check, Snapshot) ---------------------------------------------------------------------------------
// Evaluate the pending checks, oldest first.  block: wait for each; otherwise stop at the first whose event has not fired.  The first check that does not
// hold is removed and thrown as SpecFailed -- the caller (apply_gates_t) drains the stream, drops the younger checks and puts the snapshot back.  Call it only
// where nothing of the State has been replaced since the last consistent point, or where a snapshot covers what has.
inline void settle(State* s, bool block) {
    while (!s->checks.empty()) {
        Check& c = s->checks.front();
        if (block) HIPCHK(hipEventSynchronize(c.ev));
        else { const hipError_t q = hipEventQuery(c.ev); if (q == hipErrorNotReady) return; HIPCHK(q); }
        const bool ok = c.eval(s);
        const SpecFailed f{c.kind, c.step, c.iters_done};
        s->checks.pop_front();
        if (s->checks.empty()) s->arena.ring_off = 0;
        if (!ok) throw f;
    }
}
// `bytes` of pinned staging for a check's read-back (valid until the check is settled); makes room by settling what is pending when the ring is full
inline char* ring_alloc(State* s, size_t bytes) {
    HostArena& ar = s->arena;
    if (!ar.base) ar = acquire_arena();
    const size_t b = round256(std::max<size_t>(bytes, 1));
    if (b > ar.ring_cap) return nullptr;                                   // (the caller takes the careful route)
    if (ar.ring_off + b > ar.ring_cap) { settle(s, true); ar.ring_off = 0; }
    char* p = ar.ring + ar.ring_off; ar.ring_off += b; return p;
}
// the event a new check records behind its staged copy (a ring of 16: apply_gates never leaves more than 12 checks pending)
inline hipEvent_t check_event(State* s) {
    HostArena& ar = s->arena;
    if (!ar.base) ar = acquire_arena();
    hipEvent_t& e = ar.cev[ar.cevn++ & 15];
    if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return e;
}
inline Snapshot take_snapshot(const State* s) {
    Snapshot n; n.chi = s->chi; n.site = s->site; n.sscale = s->sscale; n.msg = s->msg; n.pend1 = s->pend1; n.unit_norm = s->unit_norm; n.stats = s->stats; n.real_io = s->real_io; return n;
}
inline void restore_snapshot(State* s, const Snapshot& n) {
    s->chi = n.chi; s->site = n.site; s->sscale = n.sscale; s->msg = n.msg; s->pend1 = n.pend1; s->unit_norm = n.unit_norm; s->stats = n.stats; s->real_io = n.real_io;
}
// drop every pending check unevaluated (the stream has been drained, a snapshot is about to be put back)
inline void drop_checks(State* s) { s->checks.clear(); s->arena.ring_off = 0; }

}  // namespace tnqs
