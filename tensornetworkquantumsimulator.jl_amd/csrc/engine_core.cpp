// engine_core.cpp -- see engine.hpp / engine_internal.hpp.  Pool, graph, state container, tensor and message I/O.
#include "engine_internal.hpp"

namespace tnqs {

double HostTimer::acc[16] = {0}; long HostTimer::cnt[16] = {0};
static struct HostTimerReport { ~HostTimerReport() { if (envflag("TNQS_HOST_TIMING")) for (int k = 0; k < 16; ++k) if (HostTimer::cnt[k])
    std::fprintf(stderr, "[tnqs host timing] phase %d: %.2f ms total, %ld calls, %.1f us each\n", k, HostTimer::acc[k], HostTimer::cnt[k], 1e3 * HostTimer::acc[k] / HostTimer::cnt[k]); } } g_host_timer_report;

void hipchk(hipError_t e, const char* what) {
    if (e != hipSuccess) throw Err(TNQS_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

// ---------------------------------------------------------------------------------------------------------------
// pool
// ---------------------------------------------------------------------------------------------------------------
static size_t round_size(size_t b) {
    if (b < 256) return 256;
    if (b <= (1u << 20)) return (b + 255) & ~size_t(255);
    // above 1 MiB: 8 size classes per power of two, so buffers of nearby sizes are reusable
    size_t p = size_t(1) << (63 - __builtin_clzll(b));
    size_t step = p >> 3;
    return (b + step - 1) / step * step;
}
Pool::~Pool() { set_defer(false); trim(); }
void Pool::trim() {
    std::lock_guard<std::mutex> lk(mu_);
    for (auto& kv : free_) for (void* p : kv.second) (void)hipFree(p);
    free_.clear(); cached_ = 0;
}
void* Pool::alloc(size_t bytes, size_t* rounded) {
    size_t r = round_size(bytes);
    *rounded = r;
    {
        std::lock_guard<std::mutex> lk(mu_);
        auto it = free_.find(r);
        if (it != free_.end() && !it->second.empty()) {
            void* p = it->second.back(); it->second.pop_back(); cached_ -= r; live_ += r; return p;
        }
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, r);
    if (e != hipSuccess) { trim(); e = hipMalloc(&p, r); }
    hipchk(e, "hipMalloc");
    std::lock_guard<std::mutex> lk(mu_);
    live_ += r;
    return p;
}
void Pool::release(void* p, size_t rounded) {
    std::lock_guard<std::mutex> lk(mu_);
    live_ -= rounded; cached_ += rounded;
    if (defer_) deferred_.push_back({p, rounded}); else free_[rounded].push_back(p);
}
void Pool::set_defer(bool on) {
    std::lock_guard<std::mutex> lk(mu_);
    defer_ = on;
    if (!on) { for (auto& d : deferred_) free_[d.second].push_back(d.first); deferred_.clear(); }
}
// ---------------------------------------------------------------------------------------------------------------
// graph
// ---------------------------------------------------------------------------------------------------------------
static uint64_t ekey(int a, int b) { if (a > b) std::swap(a, b); return (uint64_t(uint32_t(a)) << 32) | uint32_t(b); }
int Graph::edge(int u, int v) const { auto it = emap.find(ekey(u, v)); return it == emap.end() ? -1 : it->second; }
int Graph::leg(int v, int w) const {
    const auto& n = nbr[v];
    auto it = std::lower_bound(n.begin(), n.end(), w);
    return (it != n.end() && *it == w) ? int(it - n.begin()) : -1;
}
int Graph::dedge(int src, int dst) const { int e = edge(src, dst); if (e < 0) return -1; return 2 * e + (src == edst[e] ? 1 : 0); }

static std::shared_ptr<Graph> make_graph(int nv, int ne, const int32_t* es, const int32_t* ed) {
    auto g = std::make_shared<Graph>();
    g->nv = nv; g->ne = ne; g->esrc.assign(es, es + ne); g->edst.assign(ed, ed + ne);
    g->nbr.resize(nv); g->nbr_e.resize(nv);
    for (int e = 0; e < ne; ++e) {
        int a = es[e], b = ed[e];
        if (a < 0 || a >= nv || b < 0 || b >= nv || a == b) throw Err(TNQS_ERR_INVALID, "tnqs_create: bad edge endpoints");
        if (g->emap.count(ekey(a, b))) throw Err(TNQS_ERR_INVALID, "tnqs_create: duplicate edge");
        g->emap[ekey(a, b)] = e;
    }
    for (int v = 0; v < nv; ++v) {
        std::vector<std::pair<int, int>> tmp;
        for (int e = 0; e < ne; ++e) { if (es[e] == v) tmp.push_back({ed[e], e}); else if (ed[e] == v) tmp.push_back({es[e], e}); }
        std::sort(tmp.begin(), tmp.end());
        for (auto& p : tmp) { g->nbr[v].push_back(p.first); g->nbr_e[v].push_back(p.second); }
    }
    // forest test (union-find)
    std::vector<int> par(nv); std::iota(par.begin(), par.end(), 0);
    auto find = [&](int x) { while (par[x] != x) { par[x] = par[par[x]]; x = par[x]; } return x; };
    g->is_tree = true;
    for (int e = 0; e < ne; ++e) { int a = find(es[e]), b = find(ed[e]); if (a == b) { g->is_tree = false; break; } par[a] = b; }
    // greedy proper edge colouring in edge order
    g->ecolor.assign(ne, -1);
    std::vector<std::vector<char>> used(nv);
    for (int e = 0; e < ne; ++e) {
        int a = es[e], b = ed[e], c = 0;
        for (;; ++c) {
            bool ua = c < (int)used[a].size() && used[a][c], ub = c < (int)used[b].size() && used[b][c];
            if (!ua && !ub) break;
        }
        if ((int)used[a].size() <= c) used[a].resize(c + 1, 0);
        if ((int)used[b].size() <= c) used[b].resize(c + 1, 0);
        used[a][c] = used[b][c] = 1; g->ecolor[e] = c; g->ncolors = std::max(g->ncolors, c + 1);
    }
    return g;
}

// (host only: the graph of a handle without the handle -- include/tnqs_debug.h tnqs_dbg_default_sequence_graph)
std::shared_ptr<Graph> dbg_make_graph(int nv, int ne, const int32_t* es, const int32_t* ed) { return make_graph(nv, ne, es, ed); }

// ---------------------------------------------------------------------------------------------------------------
// state plumbing
// ---------------------------------------------------------------------------------------------------------------
// apply_gates has value semantics (apply_gates.jl:55): every call works on a copy of the handle, so a Trotter loop creates and destroys
// one State per layer.  A stream and a pinned staging arena cost milliseconds to create and to release; the ones of destroyed States are
// recycled through these small free lists instead (a State still owns its stream and arena exclusively while it lives).
static std::mutex g_recycle_mu;
static std::vector<HostArena> g_spare_arenas;                              // pinned, device independent
struct SpareStream { int device; int hi; hipStream_t st; };
static std::vector<SpareStream> g_spare_streams;                           // idle streams (hi = 1: created with the highest priority)
static const size_t kMaxSpares = 16;
HostArena acquire_arena() {
    HostArena ar{};
    { std::lock_guard<std::mutex> lk(g_recycle_mu); if (!g_spare_arenas.empty()) { ar = g_spare_arenas.back(); g_spare_arenas.pop_back(); ar.off = 0; ar.ring_off = 0; } }
    if (!ar.base) {
        // TNQS_ARENA_KB: a small arena makes every phase overflow it (tests/test_gpu_toggles.py drives the overflow path that way)
        static const size_t cap = [] { const char* e = std::getenv("TNQS_ARENA_KB"); return e ? std::max<size_t>(16, (size_t)std::atoll(e)) << 10 : size_t(32) << 20; }();
        const size_t ring = size_t(2) << 20;
        HIPCHK(hipHostMalloc((void**)&ar.base, cap + ring, hipHostMallocDefault)); ar.cap = cap - 256;
        ar.ring = ar.base + cap; ar.ring_cap = ring; ar.ring_off = 0;                               // (behind the arena proper: staged read-backs of pending checks)
    }
    return ar;
}
static void arena_free(HostArena& ar) {
    for (hipEvent_t& e : ar.cev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    if (ar.base) { (void)hipHostFree(ar.base); ar.base = nullptr; }
}
void recycle_arena(HostArena ar) {
    if (!ar.base) return;
    { std::lock_guard<std::mutex> lk(g_recycle_mu); if (g_spare_arenas.size() < kMaxSpares) { ar.off = 0; ar.ring_off = 0; g_spare_arenas.push_back(ar); ar.base = nullptr; } }
    if (ar.base) arena_free(ar);
}
static hipStream_t acquire_stream(int device, int hi = 0) {
    {
        std::lock_guard<std::mutex> lk(g_recycle_mu);
        for (size_t i = 0; i < g_spare_streams.size(); ++i)
            if (g_spare_streams[i].device == device && g_spare_streams[i].hi == hi) { hipStream_t st = g_spare_streams[i].st; g_spare_streams.erase(g_spare_streams.begin() + (std::ptrdiff_t)i); return st; }
    }
    hipStream_t st = nullptr;
    if (hi) { int least = 0, greatest = 0; HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest)); HIPCHK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, greatest)); }
    else HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    return st;
}

State::~State() {
    if (own_stream && stream) (void)hipStreamSynchronize(stream);           // nothing of this State is in flight past this point
    if (aux_stream) (void)hipStreamSynchronize(aux_stream);
    keepalive.clear(); site.clear(); msg.clear();
    HostArena ar = arena; arena = HostArena{};
    SpareStream st[2] = {{device, 0, (own_stream && stream) ? stream : nullptr}, {device, 0, aux_stream}};
    {
        std::lock_guard<std::mutex> lk(g_recycle_mu);
        if (ar.base && g_spare_arenas.size() < kMaxSpares) { ar.off = 0; g_spare_arenas.push_back(ar); ar.base = nullptr; }
        for (auto& q : st) if (q.st && g_spare_streams.size() < kMaxSpares) { g_spare_streams.push_back(q); q.st = nullptr; }
    }
    if (ar.base) arena_free(ar);
    for (auto& q : st) if (q.st) (void)hipStreamDestroy(q.st);
    for (hipEvent_t e : {ev_fork, ev_join}) if (e) (void)hipEventDestroy(e);
}
hipStream_t aux_stream_of(State* s) {
    if (!s->aux_stream) {
        s->aux_stream = acquire_stream(s->device);
        HIPCHK(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
    }
    return s->aux_stream;
}
void sync(State* s) {
    HIPCHK(hipStreamSynchronize(s->stream));
    s->keepalive.clear(); s->keep_mark = 0;
    s->arena.off = 0;
    if (s->prof) s->prof->chain = false;
}

void prof_collect(State* s) {
    Prof& P = *s->prof;
    if (P.pending.empty()) return;
    HIPCHK(hipStreamSynchronize(s->stream));
    for (auto& p : P.pending) {
        float ms = 0; (void)hipEventElapsedTime(&ms, p.a, p.b);
        P.cls[p.cls].ms += ms;
    }
    for (auto& p : P.pending) { if (p.own_a) P.ev_free.push_back(p.a); P.ev_free.push_back(p.b); }
    P.pending.clear(); P.last_b = nullptr; P.chain = false;
}

int64_t state_site_size(const State* s, int v) { return (int64_t)site_dims(s, v).n; }

template <class T> static void fill_product_up(State* s, int v) {
    // |up> = (1, 0, ...) with all bonds of dimension 1 (tensornetworkstate.jl:141-161)
    std::vector<T> h(2 * s->d[v], T(0)); h[0] = T(1);
    Buf b = dalloc(s, s->d[v] * s->esz());
    HIPCHK(hipMemcpyAsync(b->p, h.data(), s->d[v] * s->esz(), hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    s->site[v] = b;
}

State* state_create(int nv, int ne, const int32_t* es, const int32_t* ed, const int32_t* sd, int dtype, int device) {
    if (nv <= 0 || ne < 0) throw Err(TNQS_ERR_INVALID, "tnqs_create: nv must be > 0 and ne >= 0");
    if (dtype != TNQS_C64 && dtype != TNQS_C128 && dtype != TNQS_F32 && dtype != TNQS_F64) throw Err(TNQS_ERR_INVALID, "tnqs_create: unknown dtype");
    const bool real_io = (dtype == TNQS_F32 || dtype == TNQS_F64);
    if (dtype == TNQS_F32) dtype = TNQS_C64; else if (dtype == TNQS_F64) dtype = TNQS_C128;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw Err(TNQS_ERR_HIP, "tnqs_create: no HIP device available (the HIP path has no CPU fallback)");
    if (device < 0 || device >= ndev) throw Err(TNQS_ERR_INVALID, "tnqs_create: bad device index");
    HIPCHK(hipSetDevice(device));
    auto s = std::make_unique<State>();
    s->g = make_graph(nv, ne, es, ed);
    s->dtype = dtype; s->real_io = real_io; s->device = device;
    s->d.assign(nv, 2);
    if (sd) for (int v = 0; v < nv; ++v) { if (sd[v] < 1 || sd[v] > 16) throw Err(TNQS_ERR_INVALID, "tnqs_create: site dimension out of range"); s->d[v] = sd[v]; }
    s->chi.assign(ne, 1);
    s->site.resize(nv); s->sscale.assign(nv, nullptr); s->msg.assign(2 * (size_t)ne, nullptr);
    s->pend1.assign(nv, {}); s->unit_norm.assign(nv, 0);
    s->pool = std::make_shared<Pool>(device);
    s->prof = std::make_shared<Prof>();
    s->stream = acquire_stream(device); s->own_stream = true;
    for (int v = 0; v < nv; ++v) { if (dtype == TNQS_C64) fill_product_up<float>(s.get(), v); else fill_product_up<double>(s.get(), v); }
    return s.release();
}

State* state_copy(const State* o) {
    auto s = std::make_unique<State>();
    s->g = o->g; s->dtype = o->dtype; s->real_io = o->real_io; s->device = o->device; s->d = o->d; s->chi = o->chi;
    s->site = o->site; s->sscale = o->sscale; s->msg = o->msg; s->pool = o->pool; s->prof = o->prof;
    s->pend1 = o->pend1; s->unit_norm = o->unit_norm;
    s->rank = o->rank; s->nranks = o->nranks; s->owner = o->owner; s->ag_fn = o->ag_fn; s->ag_ctx = o->ag_ctx;
    s->exch = o->exch; s->exch_bytes = o->exch_bytes; s->comm = o->comm; s->force_exchange = o->force_exchange; s->msg_hermitian = o->msg_hermitian;
    HIPCHK(hipSetDevice(o->device));
    if (o->own_stream) { HIPCHK(hipStreamSynchronize(o->stream)); s->stream = acquire_stream(o->device); s->own_stream = true; }
    else { s->stream = o->stream; s->own_stream = false; }
    return s.release();
}

template <class T> static void permute_dispatch(State* s, const PermItem& it) { launch_permute<T>(s->stream, it); }

// real element types at the boundary: the caller's real array becomes (re, 0) pairs on the way in and loses its (zero) imaginary parts on
// the way out; everything in between is the complex path
static std::vector<char> widen_real(const State* s, const void* host, size_t n) {
    std::vector<char> out(n * s->esz());
    if (s->dtype == TNQS_C64) { const float* p = static_cast<const float*>(host); float* q = reinterpret_cast<float*>(out.data()); for (size_t i = 0; i < n; ++i) { q[2 * i] = p[i]; q[2 * i + 1] = 0.f; } }
    else { const double* p = static_cast<const double*>(host); double* q = reinterpret_cast<double*>(out.data()); for (size_t i = 0; i < n; ++i) { q[2 * i] = p[i]; q[2 * i + 1] = 0.0; } }
    return out;
}
// (a real handle only ever holds zero imaginary parts: real tensors, real gates -- a complex gate promotes it first; anything else is a bug
// in the library, reported instead of being dropped silently)
static void narrow_real(const State* s, const std::vector<char>& cplx, void* host, size_t n) {
    bool bad = false;
    if (s->dtype == TNQS_C64) { const float* q = reinterpret_cast<const float*>(cplx.data()); float* p = static_cast<float*>(host); for (size_t i = 0; i < n; ++i) { p[i] = q[2 * i]; bad = bad || q[2 * i + 1] != 0.f; } }
    else { const double* q = reinterpret_cast<const double*>(cplx.data()); double* p = static_cast<double*>(host); for (size_t i = 0; i < n; ++i) { p[i] = q[2 * i]; bad = bad || q[2 * i + 1] != 0.0; } }
    if (bad) throw Err(TNQS_ERR_NUMERIC, "internal: a real-typed handle holds a non-zero imaginary part");
}

void state_set_site(State* s, int v, const void* host, int ndim, const int64_t* dims, const int32_t* role) {
    const Graph& g = *s->g;
    if (v < 0 || v >= g.nv) throw Err(TNQS_ERR_INVALID, "set_site_tensor: bad vertex");
    const int z = (int)g.nbr[v].size();
    if (ndim != z + 1 || ndim > 8) throw Err(TNQS_ERR_INVALID, "set_site_tensor: tensor must have one site leg and one leg per neighbour (<= 7 neighbours)");
    // caller axis k -> canonical axis
    std::vector<int> canon_of(ndim, -1); std::vector<int> src_of(ndim, -1);
    for (int k = 0; k < ndim; ++k) {
        int c;
        if (role[k] < 0) c = 0;
        else { int j = g.leg(v, role[k]); if (j < 0) throw Err(TNQS_ERR_INVALID, "set_site_tensor: leg_role names a non-neighbour"); c = 1 + j; }
        if (src_of[c] >= 0) throw Err(TNQS_ERR_INVALID, "set_site_tensor: duplicate leg role");
        canon_of[k] = c; src_of[c] = k;
    }
    if (dims[src_of[0]] != s->d[v]) throw Err(TNQS_ERR_INVALID, "set_site_tensor: site dimension mismatch");
    size_t n = 1; std::vector<long long> stride_caller(ndim);
    for (int k = 0; k < ndim; ++k) { stride_caller[k] = (long long)n; if (dims[k] < 1) throw Err(TNQS_ERR_INVALID, "set_site_tensor: bad dim"); n *= (size_t)dims[k]; }
    HIPCHK(hipSetDevice(s->device));
    s->pend1[v].clear(); s->unit_norm[v] = 0;                   // a new tensor: nothing pending on it
    if (!s->owns(v)) {          // sharded, not ours: only the bond dimensions are recorded (host may be null)
        s->site[v] = nullptr; s->sscale[v] = nullptr;
        for (int j = 0; j < z; ++j) {
            int e = g.nbr_e[v][j]; int c = (int)dims[src_of[1 + j]];
            if (s->chi[e] != c) { s->chi[e] = c; s->msg[2 * e] = nullptr; s->msg[2 * e + 1] = nullptr; }
        }
        return;
    }
    if (!host) throw Err(TNQS_ERR_INVALID, "set_site_tensor: null data for an owned vertex");
    Buf raw = dalloc(s, n * s->esz());
    std::vector<char> widened; if (s->real_io) { widened = widen_real(s, host, n); host = widened.data(); }
    HIPCHK(hipMemcpyAsync(raw->p, host, n * s->esz(), hipMemcpyHostToDevice, s->stream));
    Buf out = dalloc(s, n * s->esz());
    PermItem it{}; it.in = raw->p; it.out = out->p; it.ndim = ndim; it.n = n;
    for (int c = 0; c < ndim; ++c) { it.dims_out[c] = (int)dims[src_of[c]]; it.stride_in[c] = stride_caller[src_of[c]]; }
    if (s->dtype == TNQS_C64) permute_dispatch<float>(s, it); else permute_dispatch<double>(s, it);
    HIPCHK(hipStreamSynchronize(s->stream));
    s->site[v] = out; s->sscale[v] = nullptr;
    for (int j = 0; j < z; ++j) {
        int e = g.nbr_e[v][j]; int c = it.dims_out[1 + j];
        if (s->chi[e] != c) { s->chi[e] = c; s->msg[2 * e] = nullptr; s->msg[2 * e + 1] = nullptr; }
    }
}

void state_set_site_random(State* s, int v, int nn, const int64_t* bond_dims, uint64_t seed, double scale) {
    const Graph& g = *s->g;
    if (v < 0 || v >= g.nv) throw Err(TNQS_ERR_INVALID, "set_site_random: bad vertex");
    const int z = (int)g.nbr[v].size();
    if (nn != z) throw Err(TNQS_ERR_INVALID, "set_site_random: one bond dimension per neighbour expected");
    size_t n = (size_t)s->d[v];
    for (int j = 0; j < z; ++j) { if (bond_dims[j] < 1 || bond_dims[j] > 4096) throw Err(TNQS_ERR_INVALID, "set_site_random: bad bond dimension"); n *= (size_t)bond_dims[j]; }
    HIPCHK(hipSetDevice(s->device));
    s->pend1[v].clear(); s->unit_norm[v] = 0;
    if (s->owns(v)) {
        Buf out = dalloc(s, n * s->esz());
        const unsigned long long sd = seed * 0x9E3779B97F4A7C15ull + (unsigned long long)v * 0xC2B2AE3D27D4EB4Full;
        if (s->dtype == TNQS_C64) launch_random_fill<float>(s->stream, out->p, n, sd, scale, s->real_io); else launch_random_fill<double>(s->stream, out->p, n, sd, scale, s->real_io);
        HIPCHK(hipStreamSynchronize(s->stream));
        s->site[v] = out; s->sscale[v] = nullptr;
    } else { s->site[v] = nullptr; s->sscale[v] = nullptr; }
    for (int j = 0; j < z; ++j) {
        const int e = g.nbr_e[v][j], c = (int)bond_dims[j];
        if (s->chi[e] != c) { s->chi[e] = c; s->msg[2 * e] = nullptr; s->msg[2 * e + 1] = nullptr; }
    }
}

void state_get_site(State* s, int v, void* host, int ndim, const int32_t* role) {
    const Graph& g = *s->g;
    if (v < 0 || v >= g.nv) throw Err(TNQS_ERR_INVALID, "get_site_tensor: bad vertex");
    if (!s->site[v]) throw Err(TNQS_ERR_INVALID, "get_site_tensor: vertex not owned by this rank");
    materialize_pending(s, {v});
    SD sd = site_dims(s, v);
    if (ndim != sd.z + 1 || ndim > 8) throw Err(TNQS_ERR_INVALID, "get_site_tensor: ndim mismatch");
    // consistency: the neighbour tensors must agree on bond dims; verify buffer size
    if (s->site[v]->bytes != sd.n * s->esz()) throw Err(TNQS_ERR_INVALID, "get_site_tensor: bond dimensions are inconsistent with the stored tensor (set all neighbours first)");
    std::vector<int> cdims(ndim); std::vector<long long> cstride(ndim);
    cdims[0] = sd.d; for (int j = 0; j < sd.z; ++j) cdims[1 + j] = sd.chi[j];
    { long long st = 1; for (int c = 0; c < ndim; ++c) { cstride[c] = st; st *= cdims[c]; } }
    PermItem it{}; it.in = s->site[v]->p; it.ndim = ndim; it.n = sd.n;
    std::vector<char> seen(ndim, 0);
    for (int k = 0; k < ndim; ++k) {
        int c;
        if (role[k] < 0) c = 0; else { int j = g.leg(v, role[k]); if (j < 0) throw Err(TNQS_ERR_INVALID, "get_site_tensor: leg_role names a non-neighbour"); c = 1 + j; }
        if (seen[c]) throw Err(TNQS_ERR_INVALID, "get_site_tensor: duplicate leg role"); seen[c] = 1;
        it.dims_out[k] = cdims[c]; it.stride_in[k] = cstride[c];
    }
    HIPCHK(hipSetDevice(s->device));
    if (s->sscale[v]) { materialize_scale(s, {v}); it.in = s->site[v]->p; }       // the caller sees the normalised tensor
    Buf out = dalloc(s, sd.n * s->esz()); it.out = out->p;
    if (s->dtype == TNQS_C64) permute_dispatch<float>(s, it); else permute_dispatch<double>(s, it);
    std::vector<char> tmp; void* dst = host; if (s->real_io) { tmp.resize(sd.n * s->esz()); dst = tmp.data(); }
    HIPCHK(hipMemcpyAsync(dst, out->p, sd.n * s->esz(), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (s->real_io) narrow_real(s, tmp, host, sd.n);
}

void state_set_message(State* s, int src, int dst, const void* host, int chi) {
    int de = s->g->dedge(src, dst);
    if (de < 0) throw Err(TNQS_ERR_INVALID, "set_message: not an edge");
    if (chi != s->chi[de / 2]) throw Err(TNQS_ERR_INVALID, "set_message: dimension does not match the bond");
    HIPCHK(hipSetDevice(s->device));
    Buf b = dalloc(s, (size_t)chi * chi * s->esz());
    std::vector<char> widened; if (s->real_io) { widened = widen_real(s, host, (size_t)chi * chi); host = widened.data(); }
    {   // BP may absorb a message on the bra side as its own conjugate transpose (engine_bp.cpp): true of every message the path itself produces (identity, diag(S),
        // updates from Hermitian messages), to rounding; a caller's message that is not Hermitian switches that route off for this handle
        auto herm = [&](auto* m, double tol) {
            double big = 0, dev = 0;
            for (int i = 0; i < chi; ++i) for (int j = 0; j <= i; ++j) {
                const double ar = m[2 * ((size_t)i * chi + j)], ai = m[2 * ((size_t)i * chi + j) + 1], br = m[2 * ((size_t)j * chi + i)], bi = m[2 * ((size_t)j * chi + i) + 1];
                big = std::max(big, std::max(std::fabs(ar) + std::fabs(ai), std::fabs(br) + std::fabs(bi))); dev = std::max(dev, std::fabs(ar - br) + std::fabs(ai + bi));
            }
            return dev <= tol * big;
        };
        const bool ok = s->dtype == TNQS_C64 ? herm(reinterpret_cast<const float*>(host), 1e-5) : herm(reinterpret_cast<const double*>(host), 1e-12);
        if (!ok) s->msg_hermitian = false;
    }
    HIPCHK(hipMemcpyAsync(b->p, host, (size_t)chi * chi * s->esz(), hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    s->msg[de] = b;
}
void state_get_message(State* s, int src, int dst, void* host, int chi) {
    int de = s->g->dedge(src, dst);
    if (de < 0) throw Err(TNQS_ERR_INVALID, "get_message: not an edge");
    if (chi != s->chi[de / 2]) throw Err(TNQS_ERR_INVALID, "get_message: dimension does not match the bond");
    HIPCHK(hipSetDevice(s->device));
    size_t bytes = (size_t)chi * chi * s->esz();
    if (!s->msg[de]) {      // default_message: identity
        const size_t st = s->real_io ? 1 : 2;
        std::memset(host, 0, (size_t)chi * chi * s->io_esz());
        for (int i = 0; i < chi; ++i) {
            if (s->dtype == TNQS_C64) reinterpret_cast<float*>(host)[st * (size_t)(i + (size_t)chi * i)] = 1.f;
            else reinterpret_cast<double*>(host)[st * (size_t)(i + (size_t)chi * i)] = 1.0;
        }
        return;
    }
    std::vector<char> tmp; void* hdst = host; if (s->real_io) { tmp.resize(bytes); hdst = tmp.data(); }
    HIPCHK(hipMemcpyAsync(hdst, s->msg[de]->p, bytes, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (s->real_io) narrow_real(s, tmp, host, (size_t)chi * chi);
}

}  // namespace tnqs
