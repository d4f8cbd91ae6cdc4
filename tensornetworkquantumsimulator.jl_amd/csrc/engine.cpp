// engine.cpp -- see engine.hpp.  Host orchestration only; every flop runs in the HIP kernels of kernels*.hip.
#include "engine.hpp"
#include <algorithm>
#include <unordered_map>
#include <functional>
#include <chrono>
#include <cstdio>
#include <array>
#include <climits>
#include <cmath>
#include <complex>
#include <cstring>
#include <numeric>
#include <mutex>
#include <set>
#include "kernels.hpp"
#include <cstdlib>
#include <type_traits>

namespace tnqs {

static size_t bp_ws_budget() { static size_t v = 0; if (!v) { const char* e = std::getenv("TNQS_BP_WS_MB"); v = (e ? (size_t)std::atoll(e) : (size_t)24576) << 20; } return v; }
static size_t jacobi_lds(size_t bytes) { static int g = -1; if (g < 0) { const char* e = std::getenv("TNQS_JACOBI_GLOBAL"); g = (e && e[0] == '1') ? 1 : 0; } return (g || bytes > 160 * 1024 - 256) ? 0 : bytes; }
static int mmax_of(const std::vector<JacobiItem>& ji) { int m = 1; for (auto& j : ji) m = std::max(m, std::max(j.m, j.n)); return m; }
static bool use_chol() { static int v = -1; if (v < 0) { const char* e = std::getenv("TNQS_NO_CHOL"); v = (e && e[0] == '1') ? 0 : 1; } return v == 1; }
static bool use_qr2() { static int v = -1; if (v < 0) { const char* e = std::getenv("TNQS_NO_QR2"); v = (e && e[0] == '1') ? 0 : 1; } return v == 1; }
static bool use_lowrank() { static int v = -1; if (v < 0) { const char* e = std::getenv("TNQS_NO_LOWRANK"); v = (e && e[0] == '1') ? 0 : 1; } return v == 1; }
static bool use_prefix() { static int v = -1; if (v < 0) { const char* e = std::getenv("TNQS_NO_PREFIX"); v = (e && e[0] == '1') ? 0 : 1; } return v == 1; }
static bool use_small_svd() { static int v = -1; if (v < 0) { const char* e = std::getenv("TNQS_NO_SMALLSVD"); v = (e && e[0] == '1') ? 0 : 1; } return v == 1; }
static bool use_apply64() { static int v = -1; if (v < 0) { const char* e = std::getenv("TNQS_NO_APPLY64"); v = (e && e[0] == '1') ? 0 : 1; } return v == 1; }
// optional host-side phase timing (TNQS_HOST_TIMING=1): printed when the process exits
struct HostTimer {
    static double acc[8]; static long cnt[8];
    int k; std::chrono::steady_clock::time_point t0; bool on;
    explicit HostTimer(int kk) : k(kk), t0(std::chrono::steady_clock::now()), on(true) {}
    void stop() { if (on) { acc[k] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); cnt[k]++; on = false; } }
    ~HostTimer() { stop(); }
};
double HostTimer::acc[8] = {0}; long HostTimer::cnt[8] = {0};
static struct HostTimerReport { ~HostTimerReport() { const char* e = std::getenv("TNQS_HOST_TIMING"); if (e && e[0] == '1') for (int k = 0; k < 8; ++k) if (HostTimer::cnt[k])
    std::fprintf(stderr, "[tnqs host timing] phase %d: %.2f ms total, %ld calls, %.1f us each\n", k, HostTimer::acc[k], HostTimer::cnt[k], 1e3 * HostTimer::acc[k] / HostTimer::cnt[k]); } } g_host_timer_report;
static bool use_dbl() { static int v = -1; if (v < 0) { const char* e = std::getenv("TNQS_NO_DOUBLE_GRAM"); v = (e && e[0] == '1') ? 0 : 1; } return v == 1; }
static bool use_mfma() { static int v = -1; if (v < 0) { const char* e = std::getenv("TNQS_NO_MFMA"); v = (e && e[0] == '1') ? 0 : 1; } return v == 1; }

void hipchk(hipError_t e, const char* what) {
    if (e != hipSuccess) throw Err(TNQS_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPCHK(x) hipchk((x), #x)
void materialize_scale(State* s, const std::vector<int>& verts);
void materialize_scale_all(State* s);

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
Pool::~Pool() { trim(); }
void Pool::trim() {
    for (auto& kv : free_) for (void* p : kv.second) (void)hipFree(p);
    free_.clear(); cached_ = 0;
}
void* Pool::alloc(size_t bytes, size_t* rounded) {
    size_t r = round_size(bytes);
    *rounded = r;
    auto it = free_.find(r);
    if (it != free_.end() && !it->second.empty()) {
        void* p = it->second.back(); it->second.pop_back(); cached_ -= r; live_ += r; return p;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, r);
    if (e != hipSuccess) { trim(); e = hipMalloc(&p, r); }
    hipchk(e, "hipMalloc");
    live_ += r;
    return p;
}
void Pool::release(void* p, size_t rounded) {
    live_ -= rounded; cached_ += rounded;
    free_[rounded].push_back(p);
}
static Buf dalloc(State* s, size_t bytes) {
    auto b = std::make_shared<DevBuf>();
    b->pool = s->pool; b->bytes = bytes;
    b->p = s->pool->alloc(bytes ? bytes : 1, &b->rounded);
    return b;
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

// ---------------------------------------------------------------------------------------------------------------
// state plumbing
// ---------------------------------------------------------------------------------------------------------------
struct HostArena {   // pinned staging for descriptor uploads, reset at host sync points
    char* base = nullptr; size_t cap = 0, off = 0;
};
static thread_local std::unordered_map<State*, HostArena> g_arenas;
// apply_gates has value semantics (apply_gates.jl:55): every call works on a copy of the handle, so a Trotter loop creates and destroys
// one State per layer.  A stream and a pinned staging arena cost milliseconds to create and to release; the ones of destroyed States are
// recycled through these small free lists instead (a State still owns its stream and arena exclusively while it lives).
static std::mutex g_recycle_mu;
static std::vector<HostArena> g_spare_arenas;                              // pinned, device independent
static std::vector<std::pair<int, hipStream_t>> g_spare_streams;           // (device, idle stream)
static const size_t kMaxSpares = 8;
static hipStream_t acquire_stream(int device) {
    {
        std::lock_guard<std::mutex> lk(g_recycle_mu);
        for (size_t i = 0; i < g_spare_streams.size(); ++i)
            if (g_spare_streams[i].first == device) { hipStream_t st = g_spare_streams[i].second; g_spare_streams.erase(g_spare_streams.begin() + i); return st; }
    }
    hipStream_t st = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    return st;
}

State::~State() {
    if (own_stream && stream) (void)hipStreamSynchronize(stream);           // nothing of this State is in flight past this point
    keepalive.clear(); site.clear(); msg.clear();
    auto it = g_arenas.find(this);
    HostArena ar{}; if (it != g_arenas.end()) { ar = it->second; g_arenas.erase(it); }
    hipStream_t st = (own_stream && stream) ? stream : nullptr;
    {
        std::lock_guard<std::mutex> lk(g_recycle_mu);
        if (ar.base && g_spare_arenas.size() < kMaxSpares) { ar.off = 0; g_spare_arenas.push_back(ar); ar.base = nullptr; }
        if (st && g_spare_streams.size() < kMaxSpares) { g_spare_streams.push_back({device, st}); st = nullptr; }
    }
    if (ar.base) (void)hipHostFree(ar.base);
    if (st) (void)hipStreamDestroy(st);
}

static void sync(State* s) {
    HIPCHK(hipStreamSynchronize(s->stream));
    s->keepalive.clear();
    auto it = g_arenas.find(s);
    if (it != g_arenas.end()) it->second.off = 0;
}

template <class Item> static const Item* upload(State* s, const std::vector<Item>& v) {
    if (v.empty()) return nullptr;
    size_t bytes = v.size() * sizeof(Item);
    HostArena& ar = g_arenas[s];
    if (!ar.base) {
        { std::lock_guard<std::mutex> lk(g_recycle_mu); if (!g_spare_arenas.empty()) { ar = g_spare_arenas.back(); g_spare_arenas.pop_back(); ar.off = 0; } }
        if (!ar.base) { ar.cap = size_t(32) << 20; HIPCHK(hipHostMalloc((void**)&ar.base, ar.cap, hipHostMallocDefault)); }
    }
    size_t aligned = (bytes + 255) & ~size_t(255);
    if (aligned > ar.cap) throw Err(TNQS_ERR_UNSUPPORTED, "descriptor batch too large");
    if (ar.off + aligned > ar.cap) sync(s);
    char* h = ar.base + ar.off; ar.off += aligned;
    std::memcpy(h, v.data(), bytes);
    Buf b = dalloc(s, bytes);
    HIPCHK(hipMemcpyAsync(b->p, h, bytes, hipMemcpyHostToDevice, s->stream));
    s->keepalive.push_back(b);
    return reinterpret_cast<const Item*>(b->p);
}

struct ProfScope {
    State* s; int cls; hipEvent_t a = nullptr, b = nullptr;
    ProfScope(State* st, int c, double bytes, double flops) : s(st), cls(c) {
        Prof& P = *s->prof;
        if (!P.on) return;
        P.cls[c].bytes += bytes; P.cls[c].flops += flops; P.cls[c].launches += 1;
        auto get = [&]() { hipEvent_t e; if (!P.ev_free.empty()) { e = P.ev_free.back(); P.ev_free.pop_back(); } else HIPCHK(hipEventCreate(&e)); return e; };
        a = get(); b = get();
        HIPCHK(hipEventRecord(a, s->stream));
    }
    ~ProfScope() {
        if (!a) return;
        (void)hipEventRecord(b, s->stream);
        s->prof->pending.push_back({cls, a, b});
    }
};
void prof_collect(State* s) {
    Prof& P = *s->prof;
    if (P.pending.empty()) return;
    HIPCHK(hipStreamSynchronize(s->stream));
    for (auto& p : P.pending) {
        float ms = 0; (void)hipEventElapsedTime(&ms, p.a, p.b);
        P.cls[p.cls].ms += ms;
        P.ev_free.push_back(p.a); P.ev_free.push_back(p.b);
    }
    P.pending.clear();
}

struct SD {       // dims of a site tensor in canonical layout
    int z = 0, d = 1; std::vector<int> chi; size_t n = 1;
    size_t pre(int j) const { size_t p = d; for (int i = 0; i < j; ++i) p *= chi[i]; return p; }
    size_t post(int j) const { size_t p = 1; for (int i = j + 1; i < z; ++i) p *= chi[i]; return p; }
};
static SD site_dims(const State* s, int v) {
    SD r; const Graph& g = *s->g;
    r.z = (int)g.nbr[v].size(); r.d = s->d[v]; r.n = r.d;
    for (int j = 0; j < r.z; ++j) { int c = s->chi[g.nbr_e[v][j]]; r.chi.push_back(c); r.n *= c; }
    return r;
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
    s->rank = o->rank; s->nranks = o->nranks; s->owner = o->owner; s->ag_fn = o->ag_fn; s->ag_ctx = o->ag_ctx;
    s->exch = o->exch; s->exch_bytes = o->exch_bytes; s->comm = o->comm;
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
static void narrow_real(const State* s, const std::vector<char>& cplx, void* host, size_t n) {
    if (s->dtype == TNQS_C64) { const float* q = reinterpret_cast<const float*>(cplx.data()); float* p = static_cast<float*>(host); for (size_t i = 0; i < n; ++i) p[i] = q[2 * i]; }
    else { const double* q = reinterpret_cast<const double*>(cplx.data()); double* p = static_cast<double*>(host); for (size_t i = 0; i < n; ++i) p[i] = q[2 * i]; }
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

void state_get_site(State* s, int v, void* host, int ndim, const int32_t* role) {
    const Graph& g = *s->g;
    if (v < 0 || v >= g.nv) throw Err(TNQS_ERR_INVALID, "get_site_tensor: bad vertex");
    if (!s->site[v]) throw Err(TNQS_ERR_INVALID, "get_site_tensor: vertex not owned by this rank");
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

// ---------------------------------------------------------------------------------------------------------------
// batched building blocks
// ---------------------------------------------------------------------------------------------------------------
static int pick_TR(size_t KK, size_t esz, int copies) {
    for (int tr : {64, 32, 16, 8, 4}) if (KK * tr * esz * copies <= 64 * 1024) return tr;
    throw Err(TNQS_ERR_UNSUPPORTED, "bond dimension too large for the fiber-tile kernels (d*chi*16*elemsize must fit 64 KiB of LDS)");
}
static void tile_params(size_t PA, size_t PB, int TR, int& TA, int& TB, int& nta, int& ntb) {
    TA = (int)std::min<size_t>(PA, TR); TB = std::max(1, TR / TA); TB = (int)std::min<size_t>(TB, PB);
    nta = (int)((PA + TA - 1) / TA); ntb = (int)((PB + TB - 1) / TB);
}

static void exchange(State* s, size_t bytes_per_rank);
static void check_exchange(const State* s, size_t bytes_per_rank);      // call BEFORE enqueuing anything that writes into s->exch
static size_t round256(size_t b);

// a chain = one site tensor pushed through several mode products (leg j with matrix X_j, chi_j x chi_j)
struct Chain {
    int v = -1; const void* src = nullptr; SD sd;
    const void* y = nullptr;                     // the untouched site tensor when src is a shared partial product of it (BP prefix sharing)
    std::vector<std::pair<int, const void*>> steps;
    const void* result = nullptr; Buf tmp[2];
};

static bool envflag(const char* name) { const char* e = std::getenv(name); return e && e[0] == '1'; }
static bool use_rowgemm() { static int v = -1; if (v < 0) v = (envflag("TNQS_NO_ROWGEMM") || envflag("TNQS_NO_CHI64")) ? 0 : 1; return v == 1; }
static bool use_gram64() { static int v = -1; if (v < 0) v = (envflag("TNQS_NO_GRAM64") || envflag("TNQS_NO_CHI64")) ? 0 : 1; return v == 1; }
static bool use_gram128() { static int v = -1; if (v < 0) v = (envflag("TNQS_NO_GRAM128") || envflag("TNQS_NO_CHI64")) ? 0 : 1; return v == 1; }
static bool use_chol128() { static int v = -1; if (v < 0) v = (envflag("TNQS_NO_CHOL128") || envflag("TNQS_NO_CHI64")) ? 0 : 1; return v == 1; }
static bool use_pair() { static int v = -1; if (v < 0) { const char* e = std::getenv("TNQS_NO_PAIR"); v = (e && e[0] == '1') ? 0 : 1; } return v == 1; }

template <class T> static void run_chains(State* s, std::vector<Chain>& chains, int cls, int cls_pair = -1) {
    if (cls_pair < 0) cls_pair = cls;
    const size_t esz = s->esz();
    std::vector<int> nt(chains.size(), 0);          // temporaries written so far (ping-pong index)
    std::vector<size_t> done(chains.size(), 0);     // steps consumed from the FRONT of c.steps after the pair stage
    for (auto& c : chains) c.result = c.src;
    // ---- stage 0: two legs per pass over the tensor -- 32-dimensional legs: mfma_pair_kernel (once), 16-dimensional legs:
    // mfma_pair16_kernel, repeated while a chain still has two of them (a degree-6 site absorbs its legs in 3 passes instead of 5) ---
    if (std::is_same<T, float>::value && use_mfma() && use_pair()) {
        std::vector<PairItem> items; int wgs = 0; double bytes = 0, flops = 0;
        double tot_slices = 0;
        std::vector<std::pair<size_t, std::pair<int, int>>> sel;     // chain index, (position of x, position of y) in c.steps
        for (size_t ci = 0; ci < chains.size(); ++ci) {
            Chain& c = chains[ci];
            if (c.steps.size() < 2) continue;
            // the two highest eligible legs (steps are in ascending leg order)
            int py = -1, px = -1;
            for (int q = (int)c.steps.size() - 1; q >= 0 && px < 0; --q) {
                int leg = c.steps[q].first;
                bool ok = c.sd.chi[leg] == 32 && leg >= 1 && (c.sd.pre(leg) % 16 == 0);
                if (!ok) continue;
                if (py < 0) py = q; else px = q;
            }
            if (px < 0) continue;
            sel.push_back({ci, {px, py}});
            tot_slices += (double)c.sd.n / (16.0 * 1024.0);
        }
        if (!sel.empty()) {
            int spw = (int)std::max(1.0, std::min(8.0, tot_slices / 2048.0));
            for (auto& se : sel) {
                Chain& c = chains[se.first];
                int x = c.steps[se.second.first].first, y = c.steps[se.second.second].first;
                PairItem it{};
                Buf& dst = c.tmp[nt[se.first] & 1];
                if (!dst) dst = dalloc(s, c.sd.n * esz);
                it.in = c.result; it.out = dst->p; it.Mx = c.steps[se.second.first].second; it.My = c.steps[se.second.second].second;
                if (!pair_geometry(c.sd.d, c.sd.z, c.sd.chi.data(), x, y, it.g)) throw Err(TNQS_ERR_HIP, "internal: pair geometry");
                int nslices = it.g.n0 * it.g.n1 * it.g.n2;
                it.spw = spw; it.slice_begin = wgs; wgs += (nslices + spw - 1) / spw;
                items.push_back(it);
                c.result = dst->p; nt[se.first]++;
                // drop the two consumed steps
                c.steps.erase(c.steps.begin() + se.second.second); c.steps.erase(c.steps.begin() + se.second.first);
                bytes += 2.0 * c.sd.n * esz; flops += 2 * 8.0 * c.sd.n * 32;
            }
            const PairItem* d = upload(s, items);
            ProfScope ps(s, cls_pair, bytes, flops);
            launch_mfma_pair(s->stream, d, (int)items.size(), wgs);
        }
        for (;;) {                                                  // 16-dimensional legs, two per round
            std::vector<Pair16Item> it16; std::vector<std::pair<size_t, std::pair<int, int>>> sel16; double slices16 = 0, by16 = 0, fl16 = 0;
            for (size_t ci = 0; ci < chains.size(); ++ci) {
                Chain& c = chains[ci];
                if (c.steps.size() < 2 || c.sd.n < (size_t)(1u << 14)) continue;     // small tensors stay on the single-leg kernel (launch bound)
                bool found = false;
                for (int qy = (int)c.steps.size() - 1; qy >= 1 && !found; --qy)
                    for (int qx = qy - 1; qx >= 0 && !found; --qx) {
                        Pair16Item it{};
                        if (!plane_geometry(c.sd.d, c.sd.z, c.sd.chi.data(), c.steps[qx].first, c.steps[qy].first, 16, it.g)) continue;
                        it.Mx = c.steps[qx].second; it.My = c.steps[qy].second;
                        it16.push_back(it); sel16.push_back({ci, {qx, qy}}); slices16 += (double)it.g.nslices(); found = true;
                    }
            }
            if (it16.empty()) break;
            // slices per workgroup: a multiple of 4 (8 waves = 4 slices x 2 halves), at least ~8 workgroups per CU overall
            int spw = 4; while (spw < 64 && slices16 / (2 * spw) >= 2048.0) spw *= 2;
            int wgs16 = 0;
            for (size_t q = 0; q < it16.size(); ++q) {
                Chain& c = chains[sel16[q].first]; Pair16Item& it = it16[q];
                Buf& dst = c.tmp[nt[sel16[q].first] & 1];
                if (!dst) dst = dalloc(s, c.sd.n * esz);
                it.in = c.result; it.out = dst->p; it.spw = spw; it.wg_begin = wgs16; wgs16 += (it.g.nslices() + spw - 1) / spw;
                c.result = dst->p; nt[sel16[q].first]++;
                c.steps.erase(c.steps.begin() + sel16[q].second.second); c.steps.erase(c.steps.begin() + sel16[q].second.first);
                by16 += 2.0 * c.sd.n * esz; fl16 += 2 * 8.0 * c.sd.n * 16;
            }
            const Pair16Item* d = upload(s, it16);
            ProfScope ps(s, cls_pair, by16, fl16);
            launch_mfma_pair16(s->stream, d, (int)it16.size(), wgs16);
        }
    }
    size_t maxsteps = 0;
    for (auto& c : chains) maxsteps = std::max(maxsteps, c.steps.size());
    for (size_t o = 0; o < maxsteps; ++o) {
        std::vector<FiberItem> items; int tiles = 0; size_t KKmax = 1; double bytes = 0, flops = 0;
        for (auto& c : chains) if (c.steps.size() > o) KKmax = std::max<size_t>(KKmax, c.sd.chi[c.steps[o].first]);
        int TR = pick_TR(KKmax, esz, 1);
        bool mf = false;
        if (std::is_same<T, float>::value && use_mfma() && KKmax >= 8) { int t = mfma_fiber_tile_rows((int)KKmax, (int)KKmax); if (t > 0) { TR = t; mf = true; } }
        int tpw = 1;
        if (mf) { double tot = 0; for (auto& c : chains) if (c.steps.size() > o) tot += (double)c.sd.n / c.sd.chi[c.steps[o].first] / TR; tpw = (int)std::max(1.0, std::min(TR == 32 ? 32.0 : 8.0, tot / 4096.0)); if (TR == 32 && tpw >= 4) tpw &= ~3; }
        std::vector<FiberItem> rg_items; double rg_tiles = 0, rg_bytes = 0, rg_flops = 0;      // chi = 64 legs: register-direct MFMA kernel
        for (size_t ci = 0; ci < chains.size(); ++ci) {
            Chain& c = chains[ci];
            if (c.steps.size() <= o) continue;
            int j = c.steps[o].first;
            FiberItem it{};
            Buf& dst = c.tmp[nt[ci] & 1];
            if (!dst) dst = dalloc(s, c.sd.n * esz);
            it.in = c.result; it.out = dst->p; it.X = c.steps[o].second;
            it.D = 1; it.PA = (int)c.sd.pre(j); it.K = c.sd.chi[j]; it.PB = (int)c.sd.post(j); it.Do = 1; it.No = it.K;
            if (std::is_same<T, float>::value && use_mfma() && use_rowgemm() && rowgemm_covers(it)) {
                rowgemm_tiles(it); it.want_norm = 0;
                rg_items.push_back(it); rg_tiles += (double)it.nta * it.ntb;
                c.result = dst->p; nt[ci]++;
                rg_bytes += 2.0 * c.sd.n * esz; rg_flops += 8.0 * c.sd.n * it.K;
                continue;
            }
            tile_params(it.PA, it.PB, TR, it.TA, it.TB, it.nta, it.ntb);
            it.tpw = mf ? tpw : 1;
            it.tile_begin = tiles; tiles += (it.nta * it.ntb + it.tpw - 1) / it.tpw; it.want_norm = 0;
            items.push_back(it);
            c.result = dst->p; nt[ci]++;
            bytes += 2.0 * c.sd.n * esz; flops += 8.0 * c.sd.n * it.K;
        }
        if (!rg_items.empty()) {
            int tpw = (int)std::max(4.0, std::min(64.0, rg_tiles / 2048.0)); tpw &= ~3; int wgs = 0;
            for (auto& it : rg_items) { it.tpw = tpw; it.tile_begin = wgs; wgs += (it.nta * it.ntb + tpw - 1) / tpw; }
            const FiberItem* d = upload(s, rg_items);
            ProfScope ps(s, cls, rg_bytes, rg_flops);
            launch_mfma_rowgemm(s->stream, d, (int)rg_items.size(), wgs, 1, nullptr);
        }
        if (items.empty()) continue;
        const FiberItem* d = upload(s, items);
        ProfScope ps(s, cls, bytes, flops);
        if (mf) launch_mfma_fiber_gemm(s->stream, d, (int)items.size(), tiles, (int)KKmax, (int)KKmax, nullptr);
        else launch_fiber_gemm<T>(s->stream, d, (int)items.size(), tiles, TR, (int)KKmax, nullptr);
    }
    (void)done;
}

// One-sided Jacobi SVD of a batch of matrices (A <- U Sigma in place; V accumulated only when the items carry one).  Three routes:
//   * the matrix fits the LDS (jacobi_lds_kernel);
//   * ComplexF32, no V wanted, too tall for the LDS but its n x n triangle fits (256 x 128 at chi = 64): Cholesky-QR preprocessing --
//     G = A^dagger A (f64) -> R = chol(G + delta I)^dagger -> Jacobi on R in LDS -> J = R^-1 (U_R S_R) (f64) -> A <- A J
//     (kernels_chi64.hip; the rotations that orthogonalise R's columns orthogonalise A's, delta only conditions R), followed by
//     polishing sweeps of the global-memory kernel on A J (relative orthogonality of the small columns);
//   * anything else: the global-memory kernel.
static bool use_tall_svd() { static int v = -1; if (v < 0) { const char* e = std::getenv("TNQS_NO_TALLSVD"); const char* f = std::getenv("TNQS_NO_CHI64"); v = ((e && e[0] == '1') || (f && f[0] == '1')) ? 0 : 1; } return v == 1; }
template <class T> static void svd_batch(State* s, const std::vector<JacobiItem>& all, bool with_v) {
    const size_t esz = s->esz();
    const size_t cap = 160 * 1024 - 256;
    std::vector<JacobiItem> fit, tall, rest;
    static const bool force_global = [] { const char* e = std::getenv("TNQS_JACOBI_GLOBAL"); return e && e[0] == '1'; }();
    for (auto& j : all) {
        if (j.n < 1 || j.m < 1) continue;
        if (!force_global && jacobi_lds_bytes(j.m, j.n, with_v, esz) <= cap && std::max(j.m, j.n) <= 256) fit.push_back(j);
        else if (!force_global && std::is_same<T, float>::value && !with_v && !j.V && use_mfma() && use_tall_svd() && j.m >= j.n && j.n <= 128 && j.n >= 2 &&
                 jacobi_lds_bytes(j.n, j.n, false, esz) <= cap) tall.push_back(j);
        else rest.push_back(j);
    }
    if (!fit.empty()) {
        size_t lds = 0; for (auto& j : fit) lds = std::max(lds, jacobi_lds_bytes(j.m, j.n, with_v, esz));
        const JacobiItem* d = upload(s, fit);
        launch_jacobi<T>(s->stream, d, (int)fit.size(), 60, lds, mmax_of(fit));
    }
    if (!rest.empty()) {
        const JacobiItem* d = upload(s, rest);
        launch_jacobi<T>(s->stream, d, (int)rest.size(), 60, 0, mmax_of(rest));
    }
    if (!tall.empty()) {
        const size_t nt = tall.size();
        size_t off = 0; std::vector<size_t> oG(nt), oL(nt), oW(nt), oR0(nt), oRr(nt), oJ(nt), oT(nt);
        for (size_t i = 0; i < nt; ++i) {
            const size_t nn = (size_t)tall[i].n * tall[i].n, mn = (size_t)tall[i].m * tall[i].n;
            oG[i] = off; off += round256(nn * 16); oL[i] = off; off += round256(nn * 16); oW[i] = off; off += round256(nn * 16);
            oR0[i] = off; off += round256(nn * 8); oRr[i] = off; off += round256(nn * 8); oJ[i] = off; off += round256(nn * 8); oT[i] = off; off += round256(mn * 8);
        }
        Buf arena = dalloc(s, off); s->keepalive.push_back(arena);
        Buf d_fail = dalloc(s, nt * sizeof(int)); s->keepalive.push_back(d_fail);
        HIPCHK(hipMemsetAsync(d_fail->p, 0, nt * sizeof(int), s->stream));
        char* ap = reinterpret_cast<char*>(arena->p);
        std::vector<TallSvdItem> ti, wi; std::vector<CholItem> ci; std::vector<JacobiItem> rj; std::vector<SmallGemmItem> gi; std::vector<CopyItem> cp;
        int nmax = 1, mmax = 1;
        for (size_t i = 0; i < nt; ++i) {
            const int m = tall[i].m, n = tall[i].n; nmax = std::max(nmax, n); mmax = std::max(mmax, m);
            ti.push_back(TallSvdItem{tall[i].A, ap + oG[i], ap + oL[i], ap + oR0[i], ap + oRr[i], m, n});
            // delta = 1e-14 of the largest diagonal entry: singular directions below 1e-7 sigma_max are f32 noise of the data anyway, and R keeps
            // a condition number <= 1e7 whatever the rank of A (no failure branch: a rank-deficient theta is the normal case early in an evolution)
            ci.push_back(CholItem{ap + oG[i], ap + oL[i], ap + oW[i], n, reinterpret_cast<int*>(d_fail->p) + i, 0.0, 1e-14});      // Winv = (L^-1)^dagger = R^-1
            rj.push_back(JacobiItem{ap + oRr[i], nullptr, n, n, tall[i].sweeps_out});
            wi.push_back(TallSvdItem{nullptr, nullptr, ap + oW[i], ap + oJ[i], ap + oRr[i], n, n});                                 // J = R^-1 (R J), f64
            gi.push_back(SmallGemmItem{tall[i].A, ap + oJ[i], ap + oT[i], m, n, n});
            cp.push_back(CopyItem{ap + oT[i], tall[i].A, (size_t)m * n * 8 / 16});
        }
        const TallSvdItem* dt = upload(s, ti); const CholItem* dc = upload(s, ci); const JacobiItem* dj = upload(s, rj);
        const TallSvdItem* dw = upload(s, wi); const SmallGemmItem* dg = upload(s, gi); const CopyItem* dcp = upload(s, cp);
        launch_tall_gram(s->stream, dt, (int)nt, nmax);
        launch_chol_packed(s->stream, dc, (int)nt, nmax);
        launch_tall_rt(s->stream, dt, (int)nt);
        size_t lds = 0; for (auto& j : rj) lds = std::max(lds, jacobi_lds_bytes(j.m, j.n, false, esz));
        launch_jacobi<T>(s->stream, dj, (int)nt, 60, lds, nmax);
        launch_tall_w(s->stream, dw, (int)nt, nmax);
        launch_small_cgemm(s->stream, dg, (int)nt, mmax, nmax);
        launch_copy_items(s->stream, dcp, (int)nt);
        // polish: J comes out of f32 arithmetic, so a column of A J with a small singular value carries rounding residue ALONG the large left
        // singular vectors (absolute size eps sigma_max -- large relative to the column itself), which the V recovery that follows
        // (theta0^dagger (U Sigma) Sigma^-2) would amplify by sigma_max / sigma_j.  One-sided Jacobi on A itself guarantees orthogonality
        // RELATIVE to the column norms; a few sweeps of the global-memory kernel on the already orthogonalised A J restore exactly that
        // (they find almost nothing to rotate: 1-2 sweeps instead of the 8-10 of a cold start).
        std::vector<JacobiItem> pol;
        for (auto& j : tall) pol.push_back(JacobiItem{j.A, nullptr, j.m, j.n, nullptr});
        const JacobiItem* dp = upload(s, pol);
        launch_jacobi<T>(s->stream, dp, (int)nt, 6, 0, mmax);
        s->stats.n_tall_svd += (int)nt;
    }
}

struct GramJob {      // out[i,j] = sum X[i,.] conj(Y[j,.]) over everything but the kept index (s and/or leg)
    const void* X; const void* Y; SD sd; int leg;  /* -1: keep the site index only */ bool keep_site;
    Buf partial; int nchunks = 0; int KK = 0;
    const void* M = nullptr;       // fused path: message absorbed on the first row leg inside the Gram kernel
};
// can the last absorption of a BP message be fused into the Gram?  (c64, d = 2, the first row leg r has chi_r = 32,
// kept leg <= 32, tiles of 64 fibers = (s, i_r) aligned)
static int fused_leg(const State* s, const SD& sd, int jo) {
    if (s->dtype != TNQS_C64 || !use_mfma() || sd.d != 2 || sd.z < 2) return -1;
    const char* e = std::getenv("TNQS_NO_FUSED_GRAM"); if (e && e[0] == '1') return -1;
    int r = (jo == 0) ? 1 : 0;
    if (sd.chi[r] != 32 || sd.chi[jo] > 32 || sd.chi[jo] < 8) return -1;
    return r;
}
template <class T, class Acc> static void run_grams(State* s, std::vector<GramJob>& jobs, int cls) {
    if (jobs.empty()) return;
    const size_t esz = s->esz();
    size_t KKmax = 1;
    for (auto& j : jobs) { j.KK = (j.keep_site ? j.sd.d : 1) * (j.leg >= 0 ? j.sd.chi[j.leg] : 1); KKmax = std::max<size_t>(KKmax, j.KK); }
    int TR = pick_TR(KKmax + 1, esz, 2);
    const bool fused = jobs[0].M != nullptr;
    const bool mf = fused || (std::is_same<T, float>::value && std::is_same<Acc, float>::value && use_mfma() && KKmax <= (use_gram64() ? 64 : 32) && KKmax >= 8);
    bool mf64 = std::is_same<T, float>::value && std::is_same<Acc, double>::value && use_mfma() && KKmax <= 64 && KKmax >= 16;
    bool mf128 = std::is_same<T, float>::value && std::is_same<Acc, double>::value && use_mfma() && use_gram128() && KKmax <= 128 && KKmax > 64;
    for (auto& j : jobs) { mf64 = mf64 && (j.X == j.Y); mf128 = mf128 && (j.X == j.Y); }
    if (mf || mf64 || mf128) TR = 64;
    const int target = 2048;
    int per_item = std::max(1, target / (int)jobs.size());
    std::vector<GramItem> items; int chunks = 0; double bytes = 0, flops = 0;
    for (auto& j : jobs) {
        GramItem it{};
        it.X = j.X; it.Y = j.Y; it.M = j.M;
        if (j.leg >= 0) {
            size_t pre = j.sd.pre(j.leg);
            if (j.keep_site) { it.D = j.sd.d; it.PA = (int)(pre / j.sd.d); } else { it.D = 1; it.PA = (int)pre; }
            it.K = j.sd.chi[j.leg]; it.PB = (int)j.sd.post(j.leg);
        } else { it.D = j.sd.d; it.PA = (int)(j.sd.n / j.sd.d); it.K = 1; it.PB = 1; }
        tile_params(it.PA, it.PB, TR, it.TA, it.TB, it.nta, it.ntb);
        int ntiles = it.nta * it.ntb;
        int nch = std::min(per_item, ntiles);
        it.tiles_per_chunk = (ntiles + nch - 1) / nch;
        it.nchunks = (ntiles + it.tiles_per_chunk - 1) / it.tiles_per_chunk;
        // 32 x 32 f32 MFMA kernels: one partial per wave; f64 64 x 64 MFMA kernel: one per tile parity; the chi = 64 kernels: one per chunk
        j.nchunks = (mf && (fused || KKmax <= 32)) ? 4 * it.nchunks : (mf64 ? 2 * it.nchunks : it.nchunks);
        j.partial = dalloc(s, (size_t)j.nchunks * j.KK * j.KK * 2 * sizeof(Acc));
        it.partial = j.partial->p; it.chunk_begin = chunks; chunks += it.nchunks;
        items.push_back(it);
        bytes += (j.X == j.Y ? 1.0 : 2.0) * j.sd.n * esz; flops += 8.0 * j.sd.n * j.KK * (j.M ? 2.0 : 1.0);
    }
    const GramItem* d = upload(s, items);
    ProfScope ps(s, cls, bytes, flops);
    if (fused) launch_mfma_gram32_fused(s->stream, d, (int)items.size(), chunks);
    else if (mf64) launch_mfma_gram64_f64(s->stream, d, (int)items.size(), chunks, (int)KKmax);
    else if (mf128) launch_mfma_gram128_f64(s->stream, d, (int)items.size(), chunks, (int)KKmax);
    else if (mf) { if (KKmax <= 32) launch_mfma_gram32(s->stream, d, (int)items.size(), chunks, (int)KKmax); else launch_mfma_gram64(s->stream, d, (int)items.size(), chunks, (int)KKmax); }
    else launch_gram<T, Acc>(s->stream, d, (int)items.size(), chunks, TR, (int)KKmax);
}

// ---------------------------------------------------------------------------------------------------------------
// BP update  (abstractbeliefpropagationcache.jl:223-259; Gauss-Seidel over edge_sequence, executed level by level)
// ---------------------------------------------------------------------------------------------------------------
struct BPPlan {
    std::vector<int> seq;                       // directed edge ids in sequence order
    std::vector<std::vector<int>> levels;       // positions in seq grouped by dependency level
    std::vector<int> pos_of;                    // de -> position in seq or -1
    std::vector<int> level_of;                  // position -> level
    bool in_place = false;                      // duplicates in the sequence: strictly sequential, single buffer
};

// Default sweep order (the reference's default is NamedGraphs' forest-cover sequence, not available here; any sequence gives the
// same fixed point, abstractbeliefpropagationcache.jl:204-218).  The edges are decomposed into LINEAR FORESTS (disjoint simple paths);
// inside a forest the messages are ordered so that every message is computed from the OLD values of the other messages of the same
// forest: along a path v0..vk the hops v_i -> v_{i+1} are listed last hop first, the hops v_{i+1} -> v_i first hop first (a message
// u -> w depends on the message entering u through its other path edge, which therefore must come LATER in the sequence).  Level
// scheduling then puts a whole forest into one level: 2 levels per sweep on a square lattice (rows, columns), and both outgoing
// messages of a site inside a forest share one pair product.  It is an ordinary sequential Gauss-Seidel order.
struct DSU { std::vector<int> p; explicit DSU(int n) : p(n) { std::iota(p.begin(), p.end(), 0); } int f(int x) { while (p[x] != x) x = p[x] = p[p[x]]; return x; }
             bool join(int a, int b) { a = f(a); b = f(b); if (a == b) return false; p[a] = b; return true; } };
// Forests: the reference's own default order (NamedGraphs forest_cover_edge_sequence: per component, post-order DFS edges towards the
// root, then their reverses in reverse order).  With it ONE sweep is exact on a tree -- which is what the reference's tree defaults
// (maxiter = 1, no tolerance; beliefpropagationcache.jl:39,110-113) rely on.  The linear-forest order below lists every message BEFORE
// the one it depends on (so that a forest is one level), i.e. information moves one hop per sweep: right for loopy graphs, where the
// fixed point is iterated anyway, wrong for the single sweep of a tree.
static std::vector<int> tree_sequence(const Graph& g) {
    std::vector<int> seq; std::vector<char> seen(g.nv, 0);
    for (int root = 0; root < g.nv; ++root) {
        if (seen[root] || g.nbr[root].empty()) continue;
        std::vector<std::pair<int, int>> post;                          // (child, parent)
        std::vector<std::pair<int, size_t>> stack{{root, 0}}; std::vector<int> par(1, -1);
        seen[root] = 1;
        while (!stack.empty()) {
            auto& top = stack.back(); const int x = top.first;
            bool pushed = false;
            while (top.second < g.nbr[x].size()) {
                const int y = g.nbr[x][top.second++];
                if (seen[y]) continue;
                seen[y] = 1; stack.push_back({y, 0}); par.push_back(x); pushed = true; break;
            }
            if (pushed) continue;
            if (par.back() >= 0) post.push_back({x, par.back()});
            stack.pop_back(); par.pop_back();
        }
        for (auto& e : post) seq.push_back(g.dedge(e.first, e.second));
        for (auto it = post.rbegin(); it != post.rend(); ++it) seq.push_back(g.dedge(it->second, it->first));
    }
    return seq;
}
static std::vector<int> default_sequence(const Graph& g) {
    if (g.is_tree) {
        std::vector<int> seq = tree_sequence(g);
        if ((int)seq.size() != 2 * g.ne) throw Err(TNQS_ERR_HIP, "internal: tree sequence does not cover every message");
        return seq;
    }
    std::vector<int> forest(g.ne, -1);
    int nf = 0;
    // 1. unions of two colour classes that contain no cycle are linear forests (straight lines on lattices): pair the colours up
    std::vector<std::vector<char>> ok(g.ncolors, std::vector<char>(g.ncolors, 0));
    for (int a = 0; a < g.ncolors; ++a) for (int b = a + 1; b < g.ncolors; ++b) {
        DSU d(g.nv); bool acyclic = true;
        for (int e = 0; e < g.ne && acyclic; ++e) if (g.ecolor[e] == a || g.ecolor[e] == b) acyclic = d.join(g.esrc[e], g.edst[e]);
        ok[a][b] = ok[b][a] = acyclic ? 1 : 0;
    }
    std::vector<int> mate(g.ncolors, -1), best;
    int best_pairs = -1;
    std::function<void(int, int)> rec = [&](int c, int pairs) {          // maximum matching of the colours (few colours: brute force)
        while (c < g.ncolors && mate[c] >= 0) ++c;
        if (c >= g.ncolors) { if (pairs > best_pairs) { best_pairs = pairs; best = mate; } return; }
        mate[c] = c; rec(c + 1, pairs); mate[c] = -1;                     // leave c single
        for (int b = c + 1; b < g.ncolors; ++b) if (mate[b] < 0 && ok[c][b]) { mate[c] = b; mate[b] = c; rec(c + 1, pairs + 1); mate[c] = mate[b] = -1; }
    };
    if (g.ncolors <= 10) rec(0, 0);
    // 2. greedy linear forests: an edge joins the first forest where both ends still have degree < 2 and no cycle closes
    std::vector<int> gforest(g.ne, -1); int gnf = 0;
    {
        std::vector<std::vector<int>> deg; std::vector<DSU> comp;
        for (int e = 0; e < g.ne; ++e) {
            int a = g.esrc[e], b = g.edst[e], f = 0;
            for (;; ++f) {
                if (f == gnf) { deg.emplace_back(g.nv, 0); comp.emplace_back(g.nv); ++gnf; }
                if (deg[f][a] < 2 && deg[f][b] < 2 && comp[f].f(a) != comp[f].f(b)) break;
            }
            comp[f].join(a, b); ++deg[f][a]; ++deg[f][b]; gforest[e] = f;
        }
    }
    // the decomposition with fewer forests (= fewer levels per sweep) wins; ties go to the colour pairs (straight lines on lattices)
    if (best_pairs > 0 && g.ncolors - best_pairs <= gnf) {
        std::vector<int> fof(g.ncolors, -1);
        for (int c = 0; c < g.ncolors; ++c) if (fof[c] < 0) { fof[c] = nf; if (best[c] != c && best[c] >= 0) fof[best[c]] = nf; ++nf; }
        for (int e = 0; e < g.ne; ++e) forest[e] = fof[g.ecolor[e]];
    } else { forest = gforest; nf = gnf; }
    std::vector<int> seq;
    for (int f = 0; f < nf; ++f) {
        std::vector<std::vector<int>> adj(g.nv);
        for (int e = 0; e < g.ne; ++e) if (forest[e] == f) { adj[g.esrc[e]].push_back(g.edst[e]); adj[g.edst[e]].push_back(g.esrc[e]); }
        std::vector<char> seen(g.nv, 0);
        for (int v = 0; v < g.nv; ++v) {
            if (adj[v].size() != 1 || seen[v]) continue;                   // start at a path end
            std::vector<int> path{v}; seen[v] = 1; int prev = -1, cur = v;
            for (;;) { int nxt = -1; for (int w : adj[cur]) if (w != prev) nxt = w; if (nxt < 0) break; prev = cur; cur = nxt; path.push_back(cur); seen[cur] = 1; }
            const int k = (int)path.size() - 1;
            for (int i = k - 1; i >= 0; --i) seq.push_back(g.dedge(path[i], path[i + 1]));
            for (int i = 0; i < k; ++i) seq.push_back(g.dedge(path[i + 1], path[i]));
        }
    }
    if ((int)seq.size() != 2 * g.ne) throw Err(TNQS_ERR_HIP, "internal: default sequence does not cover every message");
    return seq;
}

// the default order as (src, dst) vertex pairs, for tests that replay it on the oracle (include/tnqs_debug.h)
void dbg_default_sequence(const State* s, std::vector<int>& src, std::vector<int>& dst) {
    const Graph& g = *s->g;
    if (g.default_seq.empty() && g.ne > 0) g.default_seq = default_sequence(g);
    for (int de : g.default_seq) { const int e = de / 2; src.push_back((de & 1) ? g.edst[e] : g.esrc[e]); dst.push_back((de & 1) ? g.esrc[e] : g.edst[e]); }
}

static BPPlan make_plan(const State* s, const tnqs_bp_opts* o) {
    const Graph& g = *s->g;
    BPPlan p;
    if (o && o->n_sequence > 0) {
        for (int i = 0; i < o->n_sequence; ++i) {
            int de = g.dedge(o->seq_src[i], o->seq_dst[i]);
            if (de < 0) throw Err(TNQS_ERR_INVALID, "bp_update: edge_sequence contains a pair of non-adjacent vertices");
            p.seq.push_back(de);
        }
    } else { if (g.default_seq.empty() && g.ne > 0) g.default_seq = default_sequence(g); p.seq = g.default_seq; }
    p.pos_of.assign(2 * (size_t)g.ne, -1);
    for (size_t t = 0; t < p.seq.size(); ++t) { if (p.pos_of[p.seq[t]] >= 0) p.in_place = true; p.pos_of[p.seq[t]] = (int)t; }
    if (p.in_place) { for (size_t t = 0; t < p.seq.size(); ++t) { p.levels.push_back({(int)t}); p.level_of.push_back((int)t); } return p; }
    std::vector<int> level(p.seq.size(), 0); int nlev = 0;
    for (size_t t = 0; t < p.seq.size(); ++t) {
        int de = p.seq[t]; int e = de / 2; int src = (de & 1) ? g.edst[e] : g.esrc[e]; int dst = (de & 1) ? g.esrc[e] : g.edst[e];
        int lv = 0;
        for (size_t j = 0; j < g.nbr[src].size(); ++j) {
            int k = g.nbr[src][j]; if (k == dst) continue;
            int din = g.dedge(k, src); int pp = p.pos_of[din];
            if (pp >= 0 && pp < (int)t) lv = std::max(lv, level[pp] + 1);
        }
        level[t] = lv; nlev = std::max(nlev, lv + 1);
    }
    p.levels.resize(nlev);
    for (size_t t = 0; t < p.seq.size(); ++t) p.levels[level[t]].push_back((int)t);
    // the messages of a level are independent of each other: list them by source vertex, so that a workspace-bounded sub-batch (bp_update_t)
    // holds all messages of the sites it touches (they share the pair product and the double pair-Gram pass)
    auto src_of = [&](int t) { int de = p.seq[t]; int e = de / 2; return (de & 1) ? g.edst[e] : g.esrc[e]; };
    for (auto& lev : p.levels) std::stable_sort(lev.begin(), lev.end(), [&](int a, int b) { return src_of(a) < src_of(b); });
    p.level_of = level;
    return p;
}

static double default_tol(const State* s) { return s->dtype == TNQS_C64 ? 1e-5 : 1e-8; }   // beliefpropagationcache.jl:104-108

// ---- shared pair products ---------------------------------------------------------------------------------------
// A degree-4 site sends four messages per sweep, each needing the other three incoming messages absorbed.  Its legs are
// split into two pairs {A, B} by the level at which their outgoing message is computed; for an outgoing leg in A the pair
// product T_B = psi x m_b1 x m_b2 is shared with the other leg of A (the messages entering through B do not change between
// the two levels of A in the level-scheduled sequences), so a sweep costs 2 pair products + 4 (absorb + Gram) passes instead of
// 4 + 4.  Validity is not assumed but checked: an entry is reused only while the very same site / message buffers are current.
struct SharedT { Buf site, ma, mb, T; int la = -1, lb = -1; };
static bool use_tshare() { static int v = -1; if (v < 0) { const char* e = std::getenv("TNQS_NO_TSHARE"); v = (e && e[0] == '1') ? 0 : 1; } return v == 1; }

template <class T> static void bp_update_t(State* s, const tnqs_bp_opts* o, int* niter_out, double* diff_out) {
    const Graph& g = *s->g;
    HIPCHK(hipSetDevice(s->device));
    // the level schedule depends on the graph and the sequence only: the one of the default sequence is kept with the graph
    std::shared_ptr<const BPPlan> plan_p;
    if (o && o->n_sequence > 0) plan_p = std::make_shared<const BPPlan>(make_plan(s, o));
    else {
        if (!g.default_plan) g.default_plan = std::make_shared<const BPPlan>(make_plan(s, o));
        plan_p = std::static_pointer_cast<const BPPlan>(g.default_plan);
    }
    const BPPlan& plan = *plan_p;
    int maxiter = (o && o->maxiter > 0) ? o->maxiter : (g.is_tree ? 1 : 25);                  // :39,:103
    double tol;
    if (!o || std::isnan(o->tolerance)) tol = g.is_tree ? -1.0 : default_tol(s); else tol = o->tolerance;
    const bool compute_error = tol >= 0;
    const int normalize = o ? o->normalize : 1;
    if (!normalize) materialize_scale_all(s);      // un-normalised messages carry the absolute scale of the site tensors
    const size_t esz = s->esz();
    const size_t nseq = plan.seq.size();
    if (nseq == 0) { if (niter_out) *niter_out = 0; if (diff_out) *diff_out = 0; return; }
    Buf d_diffs = dalloc(s, nseq * sizeof(double));
    Buf d_sum = dalloc(s, sizeof(double));
    std::vector<Buf> cur = s->msg;
    int niter = maxiter; double avg = 0; bool converged = false;
    // shared pair products (see SharedT): partner[v][j] = the leg paired with j, -1 when the site is not covered
    std::vector<std::array<int, 4>> partner(g.nv, std::array<int, 4>{{-1, -1, -1, -1}});
    std::vector<std::array<SharedT, 2>> tshare;
    if (std::is_same<T, float>::value && use_mfma() && use_pair() && use_tshare()) {
        tshare.resize(g.nv);
        for (int v = 0; v < g.nv; ++v) {
            if (!s->owns(v) || g.nbr[v].size() != 4 || s->d[v] != 2) continue;
            bool ok = true; std::array<std::pair<int, int>, 4> ord;
            for (int j = 0; j < 4; ++j) {
                if (s->chi[g.nbr_e[v][j]] != 32) ok = false;
                int pp = plan.pos_of[g.dedge(v, g.nbr[v][j])];
                ord[j] = {pp >= 0 ? plan.level_of[pp] : INT_MAX, j};
            }
            if (!ok) continue;
            std::sort(ord.begin(), ord.end());
            partner[v][ord[0].second] = ord[1].second; partner[v][ord[1].second] = ord[0].second;
            partner[v][ord[2].second] = ord[3].second; partner[v][ord[3].second] = ord[2].second;
        }
    }
    for (int iter = 1; iter <= maxiter; ++iter) {
        std::vector<Buf> fresh(2 * (size_t)g.ne);
        for (auto& lev : plan.levels) {
            // sub-batches bounded by workspace bytes
            size_t start = 0;
            while (start < lev.size()) {
                HostTimer ht_prep(0);
                size_t budget = bp_ws_budget(), used = 0, end = start;
                while (end < lev.size()) {
                    int de = plan.seq[lev[end]]; int e = de / 2; int src = (de & 1) ? g.edst[e] : g.esrc[e];
                    size_t need = 2 * site_dims(s, src).n * esz;
                    if (end > start && used + need > budget) break;
                    used += need; ++end;
                }
                std::vector<Chain> chains; std::vector<int> tpos; std::vector<const void*> fmsg;
                std::vector<PairItem> sh_pair; std::vector<PairGramItem> sh_gram; std::vector<int> sh_chain;   // shared-T path
                std::vector<int> is_shared_chain;
                std::vector<PairGram2Item> sh_dbl; std::vector<std::pair<int, int>> sh_dbl_chain;              // both messages of a forest in one pass
                struct Pend { int idx, jo, r; };                                                                // first message of a (site, T) seen in this level
                std::unordered_map<long long, Pend> pend;
                double sh_pair_slices = 0, sh_gram_slices = 0, sh_dbl_slices = 0;
                // ---- shared partial products for the sites the plane kernels do not cover (any degree, any bond dimension): a site that sends
                // several messages in this level absorbs the messages on its OTHER legs once (T = psi x_{legs not going out here} m) and every
                // outgoing message continues from T.  With the default linear-forest order a site sends two messages per level, so a degree-6
                // site does 4 + 2 x 1 absorption passes per level instead of 2 x 5.  Reuse is decided by buffer identity per message (the
                // Gauss-Seidel rule may give two messages of a site different versions of an incoming message), never assumed.
                struct Prefix { Buf site; std::vector<std::pair<int, const void*>> legs; Buf prod; };
                std::unordered_map<int, Prefix> prefix;
                auto select_in = [&](int src, int j, int t) -> const Buf& {
                    int din = g.dedge(g.nbr[src][j], src); int pp = plan.pos_of[din];
                    return (plan.in_place || (pp >= 0 && pp < t)) ? (fresh[din] ? fresh[din] : cur[din]) : cur[din];
                };
                if (use_prefix() && !plan.in_place) {
                    std::unordered_map<int, std::vector<int>> outl;            // source site -> legs going out in this sub-batch
                    auto generic_site = [&](int src, int jo) { return tshare.empty() || site_dims(s, src).z != 4 || partner[src][jo] < 0; };
                    for (size_t q = start; q < end; ++q) {
                        int de = plan.seq[lev[q]]; int e = de / 2; int src = (de & 1) ? g.edst[e] : g.esrc[e]; int dst = (de & 1) ? g.esrc[e] : g.edst[e];
                        if (s->owns(src) && generic_site(src, g.leg(src, dst))) outl[src].push_back(g.leg(src, dst));
                    }
                    std::vector<Chain> pch; std::vector<int> psrc;
                    for (size_t q = start; q < end; ++q) {
                        int t = lev[q]; int de = plan.seq[t]; int e = de / 2; int src = (de & 1) ? g.edst[e] : g.esrc[e];
                        auto ol = outl.find(src);
                        if (ol == outl.end() || ol->second.size() < 2 || prefix.count(src)) continue;
                        Chain cp; cp.v = src; cp.src = s->site[src]->p; cp.sd = site_dims(s, src);
                        Prefix pf; pf.site = s->site[src];
                        for (int j = 0; j < cp.sd.z; ++j) {
                            if (std::find(ol->second.begin(), ol->second.end(), j) != ol->second.end()) continue;
                            const Buf& mb = select_in(src, j, t);
                            if (!mb) continue;
                            cp.steps.push_back({j, mb->p}); pf.legs.push_back({j, mb->p});
                        }
                        if (pf.legs.empty()) continue;
                        prefix[src] = pf; pch.push_back(std::move(cp)); psrc.push_back(src);
                    }
                    if (!pch.empty()) {
                        run_chains<T>(s, pch, TNQS_PROF_BP_MODEPROD, TNQS_PROF_BP_PAIR);
                        for (size_t i = 0; i < pch.size(); ++i) {
                            Prefix& pf = prefix[psrc[i]];
                            for (int k = 0; k < 2; ++k) if (pch[i].tmp[k] && pch[i].tmp[k]->p == pch[i].result) pf.prod = pch[i].tmp[k];
                        }
                    }
                }
                for (size_t q = start; q < end; ++q) {
                    int t = lev[q]; int de = plan.seq[t]; int e = de / 2;
                    int src = (de & 1) ? g.edst[e] : g.esrc[e]; int dst = (de & 1) ? g.esrc[e] : g.edst[e];
                    if (!s->owns(src)) continue;
                    Chain c; c.v = src; c.src = s->site[src]->p; c.sd = site_dims(s, src);
                    const int jo = g.leg(src, dst);
                    if (!tshare.empty() && c.sd.z == 4 && partner[src][jo] >= 0) {
                        const int r = partner[src][jo];
                        int pa = -1, pb = -1;
                        for (int j = 0; j < 4; ++j) if (j != jo && j != r) { if (pa < 0) pa = j; else pb = j; }
                        auto incoming = [&](int j) -> const Buf& {
                            int din = g.dedge(g.nbr[src][j], src); int pp = plan.pos_of[din];
                            return (plan.in_place || (pp >= 0 && pp < t)) ? (fresh[din] ? fresh[din] : cur[din]) : cur[din];
                        };
                        const Buf& ma = incoming(pa); const Buf& mb = incoming(pb); const Buf& mr = incoming(r);
                        PairGramItem gi{}; PairItem pi{};
                        if (ma && mb && mr && pair_geometry(c.sd.d, c.sd.z, c.sd.chi.data(), pa, pb, pi.g)
                            && pair_geometry(c.sd.d, c.sd.z, c.sd.chi.data(), r, jo, gi.g)) {
                            SharedT& sh = tshare[src][std::min(pa, pb) < std::min(r, jo) ? 0 : 1];      // slot of the pair {pa, pb}
                            if (!(sh.T && sh.site == s->site[src] && sh.ma == ma && sh.mb == mb && sh.la == pa && sh.lb == pb)) {
                                sh.site = s->site[src]; sh.ma = ma; sh.mb = mb; sh.la = pa; sh.lb = pb;
                                sh.T = dalloc(s, c.sd.n * esz);
                                pi.in = c.src; pi.out = sh.T->p; pi.Mx = ma->p; pi.My = mb->p;
                                sh_pair.push_back(pi); sh_pair_slices += (double)c.sd.n / 16384.0;
                            }
                            gi.X = sh.T->p; gi.Y = c.src; gi.M = mr->p;
                            const long long key = ((long long)src << 1) | (std::min(pa, pb) < std::min(r, jo) ? 0 : 1);
                            auto pit = use_dbl() ? pend.find(key) : pend.end();
                            if (pit != pend.end() && pit->second.jo == r && pit->second.r == jo && sh_gram[pit->second.idx].X == gi.X) {
                                // the partner message of the same forest is in this level too: one pass computes both
                                PairGramItem& first = sh_gram[pit->second.idx];       // plane (lx = r_first = jo, ly = jo_first = r)
                                PairGram2Item d2{}; d2.X = first.X; d2.Y = first.Y; d2.Mx = first.M; d2.My = gi.M; d2.g = first.g;
                                sh_dbl.push_back(d2); sh_dbl_chain.push_back({sh_chain[pit->second.idx], (int)chains.size()});
                                sh_dbl_slices += (double)c.sd.n / 8192.0;
                                first.X = nullptr;                                     // retired from the single list
                                sh_gram_slices -= (double)c.sd.n / 16384.0;
                                pend.erase(pit);
                            } else {
                                if (use_dbl()) pend[key] = Pend{(int)sh_gram.size(), jo, r};
                                sh_gram.push_back(gi); sh_gram_slices += (double)c.sd.n / 16384.0;
                                sh_chain.push_back((int)chains.size());
                            }
                            is_shared_chain.push_back((int)chains.size());
                            chains.push_back(std::move(c)); tpos.push_back(t); fmsg.push_back(nullptr);
                            continue;
                        }
                    }
                    const int fr = fused_leg(s, c.sd, jo);
                    const void* fm = nullptr;
                    std::vector<char> done(c.sd.z, 0);                   // legs already absorbed in the shared partial product
                    {
                        auto pf = prefix.find(src);
                        if (pf != prefix.end() && pf->second.prod && pf->second.site == s->site[src]) {
                            bool same = true;
                            for (auto& lm : pf->second.legs) { if (lm.first == jo) { same = false; break; } const Buf& mb = select_in(src, lm.first, t); if (!mb || mb->p != lm.second) { same = false; break; } }
                            if (same) { c.y = c.src; c.src = pf->second.prod->p; for (auto& lm : pf->second.legs) done[lm.first] = 1; }
                        }
                    }
                    for (int j = 0; j < c.sd.z; ++j) {
                        int k = g.nbr[src][j]; if (k == dst || done[j]) continue;
                        int din = g.dedge(k, src); int pp = plan.pos_of[din];
                        const Buf& mb = (plan.in_place || (pp >= 0 && pp < t)) ? (fresh[din] ? fresh[din] : cur[din]) : cur[din];
                        if (!mb) continue;                               // unset message = identity: nothing to absorb
                        if (j == fr) fm = mb->p;                         // absorbed inside the Gram kernel
                        else c.steps.push_back({j, mb->p});
                    }
                    chains.push_back(std::move(c)); tpos.push_back(t); fmsg.push_back(fm);
                }
                // ---- 16-dimensional planes: the two messages a site sends in this level, both continuing from the same shared product and
                // each absorbing exactly the other's outgoing leg, come from ONE pass over (T, psi) (mfma_pair_gram2x16_kernel) ------------
                std::vector<PairGram2x16Item> g16; std::vector<std::pair<int, int>> g16_chain;       // (chain of the message through ly, through lx)
                if (std::is_same<T, float>::value && use_mfma() && use_pair() && use_dbl()) {
                    std::unordered_map<int, std::vector<int>> by_src;
                    for (size_t ci = 0; ci < chains.size(); ++ci)
                        if (chains[ci].y && chains[ci].steps.size() == 1 && !fmsg[ci] && chains[ci].sd.n >= (size_t)(1u << 14)) by_src[chains[ci].v].push_back((int)ci);
                    for (auto& kv : by_src) {
                        if (kv.second.size() != 2) continue;
                        const int ci = kv.second[0], cj = kv.second[1];
                        Chain& a = chains[ci]; Chain& b = chains[cj];
                        auto out_leg = [&](int c) { int de = plan.seq[tpos[c]]; int e = de / 2; int dst = (de & 1) ? g.esrc[e] : g.edst[e]; return g.leg(chains[c].v, dst); };
                        const int ly = out_leg(ci), lx = out_leg(cj);
                        if (a.src != b.src || a.y != b.y || a.steps[0].first != lx || b.steps[0].first != ly) continue;
                        PairGram2x16Item it{};
                        if (!plane_geometry(a.sd.d, a.sd.z, a.sd.chi.data(), lx, ly, 16, it.g)) continue;
                        it.X = a.src; it.Y = a.y; it.Mx = a.steps[0].second; it.My = b.steps[0].second;
                        g16.push_back(it); g16_chain.push_back({ci, cj});
                        a.steps.clear(); b.steps.clear();
                        is_shared_chain.push_back(ci); is_shared_chain.push_back(cj);
                    }
                }
                ht_prep.stop();
                std::vector<char> is_shared(chains.size(), 0);
                for (int ci : is_shared_chain) is_shared[ci] = 1;
                HostTimer ht_launch(1);
                if (!sh_pair.empty()) {
                    int spw = (int)std::max(1.0, std::min(8.0, sh_pair_slices / 2048.0)); int wgs = 0;
                    for (auto& it : sh_pair) { it.spw = spw; it.slice_begin = wgs; wgs += (it.g.n0 * it.g.n1 * it.g.n2 + spw - 1) / spw; }
                    const PairItem* d = upload(s, sh_pair);
                    ProfScope ps(s, TNQS_PROF_BP_PAIR, 2.0 * sh_pair_slices * 16384.0 * esz, 2 * 8.0 * sh_pair_slices * 16384.0 * 32);
                    launch_mfma_pair(s->stream, d, (int)sh_pair.size(), wgs);
                }
                run_chains<T>(s, chains, TNQS_PROF_BP_MODEPROD, TNQS_PROF_BP_PAIR);
                std::vector<GramJob> jobs;
                for (size_t i = 0; i < chains.size(); ++i) {
                    int de = plan.seq[tpos[i]]; int e = de / 2; int dst = (de & 1) ? g.esrc[e] : g.edst[e];
                    GramJob j{}; j.X = chains[i].result; j.Y = chains[i].y ? chains[i].y : chains[i].src; j.sd = chains[i].sd; j.leg = g.leg(chains[i].v, dst); j.keep_site = false;
                    j.M = fmsg[i];
                    jobs.push_back(j);
                }
                if (!sh_dbl.empty()) {
                    // full slices per workgroup pair: the largest power of two that still gives >= 4 workgroups per CU; an item gets groups
                    // of 16 workgroups (8 pairs), so powers of two avoid idle workgroups for the usual 2^k slices per site
                    int spw = 16, wgs = 0;
                    for (; spw > 1; spw >>= 1) {
                        long tot = 0;
                        for (auto& it : sh_dbl) { int np = (it.g.n0 * it.g.n1 * it.g.n2 + spw - 1) / spw; tot += 16 * ((np + 7) / 8); }
                        if (tot >= 1024) break;
                    }
                    for (size_t q = 0; q < sh_dbl.size(); ++q) {
                        PairGram2Item& it = sh_dbl[q]; GramJob& jy = jobs[sh_dbl_chain[q].first]; GramJob& jx = jobs[sh_dbl_chain[q].second];
                        const int npairs = (it.g.n0 * it.g.n1 * it.g.n2 + spw - 1) / spw;     // workgroup pairs (one per half), in groups of 8 pairs
                        const int nwg = 16 * ((npairs + 7) / 8);
                        it.spw = spw; it.wg_begin = wgs; wgs += nwg;
                        jy.nchunks = jx.nchunks = nwg; jy.KK = jx.KK = 32;             // one partial per workgroup
                        jy.partial = dalloc(s, (size_t)jy.nchunks * 1024 * esz); jx.partial = dalloc(s, (size_t)jx.nchunks * 1024 * esz);
                        it.partial_y = jy.partial->p; it.partial_x = jx.partial->p;
                    }
                    const PairGram2Item* d = upload(s, sh_dbl);
                    ProfScope ps(s, TNQS_PROF_BP_PAIRGRAM, 2.0 * sh_dbl_slices * 8192.0 * esz, 4 * 8.0 * sh_dbl_slices * 8192.0 * 32);
                    launch_mfma_pair_gram2(s->stream, d, (int)sh_dbl.size(), wgs);
                }
                if (!g16.empty()) {
                    double tot = 0; for (auto& it : g16) tot += it.g.nslices();
                    int spw = 2; while (spw < 128 && tot / (2 * spw) >= 2048.0) spw *= 2;       // a multiple of 2: 4 waves = 2 slices x 2 halves
                    int wgs = 0; double by = 0, fl = 0;
                    for (size_t q = 0; q < g16.size(); ++q) {
                        PairGram2x16Item& it = g16[q]; GramJob& jy = jobs[g16_chain[q].first]; GramJob& jx = jobs[g16_chain[q].second];
                        const int nwg = (it.g.nslices() + spw - 1) / spw;
                        it.spw = spw; it.wg_begin = wgs; wgs += nwg;
                        jy.nchunks = jx.nchunks = nwg; jy.KK = jx.KK = 16;
                        jy.partial = dalloc(s, (size_t)nwg * 256 * esz); jx.partial = dalloc(s, (size_t)nwg * 256 * esz);
                        it.partial_y = jy.partial->p; it.partial_x = jx.partial->p;
                        by += 2.0 * jy.sd.n * esz; fl += 4 * 8.0 * jy.sd.n * 16;
                    }
                    const PairGram2x16Item* d = upload(s, g16);
                    ProfScope ps(s, TNQS_PROF_BP_PAIRGRAM, by, fl);
                    launch_mfma_pair_gram2x16(s->stream, d, (int)g16.size(), wgs);
                }
                {   // singles: drop the entries that were merged into a double item
                    std::vector<PairGramItem> keep; std::vector<int> keepc;
                    for (size_t q = 0; q < sh_gram.size(); ++q) if (sh_gram[q].X) { keep.push_back(sh_gram[q]); keepc.push_back(sh_chain[q]); }
                    sh_gram.swap(keep); sh_chain.swap(keepc);
                }
                if (!sh_gram.empty()) {
                    int spw = (int)std::max(4.0, std::min(16.0, sh_gram_slices / 2048.0)); int wgs = 0;
                    for (size_t q = 0; q < sh_gram.size(); ++q) {
                        PairGramItem& it = sh_gram[q]; GramJob& j = jobs[sh_chain[q]];
                        int nwg = (it.g.n0 * it.g.n1 * it.g.n2 + spw - 1) / spw;
                        it.spw = spw; it.wg_begin = wgs; wgs += nwg;
                        j.nchunks = nwg; j.KK = 32; j.partial = dalloc(s, (size_t)j.nchunks * 1024 * esz);
                        it.partial = j.partial->p;
                    }
                    const PairGramItem* d = upload(s, sh_gram);
                    ProfScope ps(s, TNQS_PROF_BP_PAIRGRAM, 2.0 * sh_gram_slices * 16384.0 * esz, 2 * 8.0 * sh_gram_slices * 16384.0 * 32);
                    launch_mfma_pair_gram(s->stream, d, (int)sh_gram.size(), wgs);
                }
                {   // the fused and the plain Gram are different kernels: run them as two batches, keep the job order
                    std::vector<GramJob> jf, jp; std::vector<size_t> idf, idp;
                    for (size_t i = 0; i < jobs.size(); ++i) { if (is_shared[i]) continue; if (jobs[i].M) { jf.push_back(jobs[i]); idf.push_back(i); } else { jp.push_back(jobs[i]); idp.push_back(i); } }
                    run_grams<T, T>(s, jf, TNQS_PROF_BP_FUSED);
                    run_grams<T, T>(s, jp, TNQS_PROF_BP_GRAM);
                    for (size_t q = 0; q < jf.size(); ++q) jobs[idf[q]] = jf[q];
                    for (size_t q = 0; q < jp.size(); ++q) jobs[idp[q]] = jp[q];
                }
                ht_launch.stop();
                HostTimer ht_fin(2);
                std::vector<MsgFinalItem> fin;
                if (s->nranks <= 1) {
                    for (size_t i = 0; i < jobs.size(); ++i) {
                        int t = tpos[i]; int de = plan.seq[t]; int c = s->chi[de / 2];
                        Buf nb = dalloc(s, (size_t)c * c * esz);
                        MsgFinalItem f{}; f.partial = jobs[i].partial->p; f.nchunks = jobs[i].nchunks; f.chi = c;
                        const Buf& oldb = plan.in_place && fresh[de] ? fresh[de] : cur[de];
                        f.old_msg = oldb ? oldb->p : nullptr; f.new_msg = nb->p;
                        f.diff_out = reinterpret_cast<double*>(d_diffs->p) + t; f.normalize = normalize;
                        fin.push_back(f);
                        fresh[de] = nb;
                    }
                } else {
                    // sharded: owners reduce their raw messages into the exchange buffer, all-gather, then EVERY rank
                    // normalises / diffs every message of the sub-batch (messages are replicated)
                    std::vector<size_t> slot(end - start, 0); std::vector<size_t> rank_bytes(s->nranks, 0);
                    for (size_t q = start; q < end; ++q) {
                        int de = plan.seq[lev[q]]; int e = de / 2; int src = (de & 1) ? g.edst[e] : g.esrc[e];
                        int r = s->owner[src]; slot[q - start] = rank_bytes[r];
                        rank_bytes[r] += round256((size_t)s->chi[e] * s->chi[e] * esz);
                    }
                    size_t stride = 0; for (size_t b : rank_bytes) stride = std::max(stride, b);
                    check_exchange(s, stride);
                    char* base = reinterpret_cast<char*>(s->exch);
                    std::vector<ReduceItem> ri; int elems = 0; size_t oi = 0;
                    for (size_t q = start; q < end; ++q) {
                        int de = plan.seq[lev[q]]; int e = de / 2; int src = (de & 1) ? g.edst[e] : g.esrc[e];
                        if (!s->owns(src)) continue;
                        int n2 = s->chi[e] * s->chi[e];
                        ri.push_back(ReduceItem{jobs[oi].partial->p, base + (size_t)s->rank * stride + slot[q - start], n2, jobs[oi].nchunks, 0, elems});
                        elems += n2; ++oi;
                    }
                    if (!ri.empty()) { const ReduceItem* dr = upload(s, ri); launch_reduce<T, T>(s->stream, dr, (int)ri.size(), elems); }
                    exchange(s, stride);
                    for (size_t q = start; q < end; ++q) {
                        int t = lev[q]; int de = plan.seq[t]; int e = de / 2; int src = (de & 1) ? g.edst[e] : g.esrc[e];
                        int c = s->chi[e];
                        Buf nb = dalloc(s, (size_t)c * c * esz);
                        MsgFinalItem f{}; f.partial = base + (size_t)s->owner[src] * stride + slot[q - start]; f.nchunks = 1; f.chi = c;
                        const Buf& oldb = plan.in_place && fresh[de] ? fresh[de] : cur[de];
                        f.old_msg = oldb ? oldb->p : nullptr; f.new_msg = nb->p;
                        f.diff_out = reinterpret_cast<double*>(d_diffs->p) + t; f.normalize = normalize;
                        fin.push_back(f);
                        fresh[de] = nb;
                    }
                }
                {
                    const MsgFinalItem* d = upload(s, fin);
                    ProfScope ps(s, TNQS_PROF_SMALL, 0, 0);
                    launch_msg_finalize<T>(s->stream, d, (int)fin.size());
                }
                ht_fin.stop();
                // (sharded) the next sub-batch writes the exchange buffer again: ordered after this finalize by the stream; the host-side
                // all-gather callback is always preceded by a stream synchronisation inside exchange()
                start = end;
            }
        }
        for (size_t t = 0; t < nseq; ++t) if (fresh[plan.seq[t]]) cur[plan.seq[t]] = fresh[plan.seq[t]];
        s->stats.n_bp_sweeps += 1;
        if (compute_error) {
            launch_sum_doubles(s->stream, reinterpret_cast<const double*>(d_diffs->p), (int)nseq, reinterpret_cast<double*>(d_sum->p));
            double tot = 0;
            HIPCHK(hipMemcpyAsync(&tot, d_sum->p, sizeof(double), hipMemcpyDeviceToHost, s->stream));
            sync(s);
            avg = tot / (double)nseq;
            if (avg <= tol) { converged = true; niter = iter; break; }
        }
    }
    sync(s);
    s->msg = cur;
    s->stats.n_bp_updates += 1;
    if (compute_error && !converged) s->stats.bp_not_converged += 1;
    s->stats.last_bp_diff = avg;
    if (niter_out) *niter_out = niter;
    if (diff_out) *diff_out = compute_error ? avg : -1.0;
}

void bp_update(State* s, const tnqs_bp_opts* o, int* niter, double* diff) {
    if (s->dtype == TNQS_C64) bp_update_t<float>(s, o, niter, diff); else bp_update_t<double>(s, o, niter, diff);
}

// ---------------------------------------------------------------------------------------------------------------
// gates
// ---------------------------------------------------------------------------------------------------------------
struct Gate1 { int v; const double* mat; };
struct Gate2 { int v1, v2; const double* mat; int index; };

static bool eager_scale() { static int v = -1; if (v < 0) { const char* e = std::getenv("TNQS_EAGER_SCALE"); v = (e && e[0] == '1') ? 1 : 0; } return v == 1; }
// apply the pending scale factors of `verts` (out of place: site buffers may be shared with copies of the handle)
template <class T> static void materialize_scale_t(State* s, const std::vector<int>& verts) {
    std::vector<ScaleItem> sc; std::vector<Buf> outs; std::vector<int> vs;
    for (int v : verts) {
        if (v < 0 || v >= (int)s->site.size() || !s->site[v] || !s->sscale[v]) continue;
        Buf out = dalloc(s, s->site[v]->bytes);
        ScaleItem it{}; it.src = s->site[v]->p; it.dst = out->p; it.n = s->site[v]->bytes / s->esz(); it.factor = reinterpret_cast<const double*>(s->sscale[v]->p);
        sc.push_back(it); outs.push_back(out); vs.push_back(v);
    }
    if (sc.empty()) return;
    HIPCHK(hipSetDevice(s->device));
    const ScaleItem* d = upload(s, sc);
    { ProfScope ps(s, TNQS_PROF_SMALL, 0, 0); launch_scale<T>(s->stream, d, (int)sc.size()); }
    for (size_t i = 0; i < vs.size(); ++i) { s->keepalive.push_back(s->site[vs[i]]); s->keepalive.push_back(s->sscale[vs[i]]); s->site[vs[i]] = outs[i]; s->sscale[vs[i]] = nullptr; }
}
void materialize_scale(State* s, const std::vector<int>& verts) {
    if (s->dtype == TNQS_C64) materialize_scale_t<float>(s, verts); else materialize_scale_t<double>(s, verts);
}
void materialize_scale_all(State* s) {
    std::vector<int> all(s->site.size()); std::iota(all.begin(), all.end(), 0);
    materialize_scale(s, all);
}

// the new site tensors replace the old ones; with `normalize` their norm (from the producing kernel's partial sums) becomes
// the pending scale factor 1/||psi|| instead of a scaling pass over the tensor (simple_update.jl:66-72 normalises eagerly;
// every later step of the path is invariant under a real rescaling of a site tensor, see engine.hpp State::sscale)
template <class T> static void norm_and_replace(State* s, std::vector<int>& verts, std::vector<Buf>& outs,
                                                std::vector<size_t>& nelem, Buf norm_partials,
                                                std::vector<int>& tile_begin, std::vector<int>& ntiles, bool normalize) {
    (void)nelem;
    if (normalize) {
        std::vector<NormFactorItem> nf;
        Buf fac = dalloc(s, verts.size() * 256);           // one factor per site, 256-byte slots (aliased Bufs below)
        for (size_t i = 0; i < verts.size(); ++i) {
            NormFactorItem it{}; it.norm_partials = reinterpret_cast<const double*>(norm_partials->p) + tile_begin[i]; it.npart = ntiles[i];
            it.factor = reinterpret_cast<double*>(reinterpret_cast<char*>(fac->p) + 256 * i);
            nf.push_back(it);
        }
        const NormFactorItem* d = upload(s, nf);
        { ProfScope ps(s, TNQS_PROF_SMALL, 0, 0); launch_norm_factor(s->stream, d, (int)nf.size()); }
        for (size_t i = 0; i < verts.size(); ++i) { s->site[verts[i]] = outs[i]; s->sscale[verts[i]] = sub_buffer(fac, 256 * i, 8); }
        if (eager_scale()) materialize_scale_t<T>(s, verts);
    } else {
        for (size_t i = 0; i < verts.size(); ++i) s->site[verts[i]] = outs[i];       // a pending factor of the input carries over (linear map)
    }
}

template <class T> static void apply_one_site_batch(State* s, const std::vector<Gate1>& gates, bool normalize) {
    if (gates.empty()) return;
    const size_t esz = s->esz();
    if (std::is_same<T, float>::value) {
        bool all2 = true; for (auto& g1 : gates) all2 = all2 && s->d[g1.v] == 2;
        if (all2) {         // streaming 2x2 kernel (HBM-bound: read + write each site tensor once)
            const int NBX = 64;
            std::vector<Site1Item> items; std::vector<int> verts, tb, nt; std::vector<Buf> outs; std::vector<size_t> ne; double bytes = 0, flops = 0;
            for (auto& g1 : gates) {
                if (!s->owns(g1.v)) continue;
                SD sd = site_dims(s, g1.v);
                Site1Item it{}; Buf out = dalloc(s, sd.n * esz);
                it.in = s->site[g1.v]->p; it.out = out->p; it.npairs = sd.n / 2;
                // column-major G[s' + 2 s]: g00 = mat[0], g10 = mat[1], g01 = mat[2], g11 = mat[3]
                const double* m = g1.mat;
                it.g[0] = (float)m[0]; it.g[1] = (float)m[1]; it.g[2] = (float)m[4]; it.g[3] = (float)m[5];
                it.g[4] = (float)m[2]; it.g[5] = (float)m[3]; it.g[6] = (float)m[6]; it.g[7] = (float)m[7];
                verts.push_back(g1.v); outs.push_back(out); ne.push_back(sd.n); tb.push_back((int)items.size() * NBX); nt.push_back(NBX);
                items.push_back(it);
                bytes += 2.0 * sd.n * esz; flops += 8.0 * sd.n * 2;
            }
            if (items.empty()) return;
            Buf np = dalloc(s, items.size() * NBX * sizeof(double));
            const Site1Item* d = upload(s, items);
            { ProfScope ps(s, TNQS_PROF_GATE_APPLY, bytes, flops);
              launch_site1_c64(s->stream, d, (int)items.size(), NBX, normalize ? reinterpret_cast<double*>(np->p) : nullptr); }
            norm_and_replace<T>(s, verts, outs, ne, np, tb, nt, normalize);
            return;
        }
    }
    std::vector<FiberItem> items; std::vector<int> verts, tb, nt; std::vector<Buf> outs; std::vector<size_t> ne;
    int tiles = 0; size_t KKmax = 1; double bytes = 0, flops = 0;
    for (auto& g1 : gates) KKmax = std::max<size_t>(KKmax, s->d[g1.v]);
    const int TR = pick_TR(KKmax, esz, 1);
    std::vector<T> hx;       // X[kk + d*nn] = G[nn, kk]  (out[s'] = sum_s G[s', s] psi[s], simple_update.jl:27)
    std::vector<size_t> xoff;
    for (auto& g1 : gates) {
        int d = s->d[g1.v]; xoff.push_back(hx.size());
        for (int nn = 0; nn < d; ++nn) for (int kk = 0; kk < d; ++kk) { hx.push_back((T)g1.mat[2 * (nn + d * kk)]); hx.push_back((T)g1.mat[2 * (nn + d * kk) + 1]); }
    }
    // note: column-major X means index kk + d*nn; the loop above emits nn-major order, i.e. X[kk + d*nn] at position nn*d + kk
    const char* dxp;
    {
        std::vector<char> raw(reinterpret_cast<char*>(hx.data()), reinterpret_cast<char*>(hx.data()) + hx.size() * sizeof(T));
        dxp = upload(s, raw);
    }
    size_t gi = 0;
    for (auto& g1 : gates) {
        if (!s->owns(g1.v)) { ++gi; continue; }
        SD sd = site_dims(s, g1.v);
        FiberItem it{}; Buf out = dalloc(s, sd.n * esz);
        it.in = s->site[g1.v]->p; it.out = out->p; it.X = dxp + xoff[gi] * sizeof(T);
        it.D = sd.d; it.PA = (int)(sd.n / sd.d); it.K = 1; it.PB = 1; it.Do = sd.d; it.No = 1;
        tile_params(it.PA, it.PB, TR, it.TA, it.TB, it.nta, it.ntb);
        it.tpw = 1; it.tile_begin = tiles; it.want_norm = normalize ? 1 : 0;
        verts.push_back(g1.v); outs.push_back(out); ne.push_back(sd.n); tb.push_back(tiles); nt.push_back(it.nta * it.ntb);
        tiles += it.nta * it.ntb; items.push_back(it);
        bytes += 2.0 * sd.n * esz; flops += 8.0 * sd.n * sd.d;
        ++gi;
    }
    if (items.empty()) return;
    Buf np = dalloc(s, std::max(1, tiles) * sizeof(double));
    const FiberItem* d = upload(s, items);
    { ProfScope ps(s, TNQS_PROF_GATE_APPLY, bytes, flops);
      launch_fiber_gemm<T>(s->stream, d, (int)items.size(), tiles, TR, (int)KKmax, reinterpret_cast<double*>(np->p)); }
    norm_and_replace<T>(s, verts, outs, ne, np, tb, nt, normalize);
}

// all-gather of equal-sized per-rank blocks laid out back to back in the host-provided exchange buffer
void check_exchange(const State* s, size_t bytes_per_rank) {
    if (s->nranks <= 1) return;
    if (bytes_per_rank * (size_t)s->nranks > s->exch_bytes)
        throw Err(TNQS_ERR_COMM, "exchange buffer too small for this batch: " + std::to_string(bytes_per_rank * (size_t)s->nranks) + " bytes needed, " +
                                     std::to_string(s->exch_bytes) + " available (raise the buffer size passed to tnqs_set_sharding)");
}
void exchange(State* s, size_t bytes_per_rank) {
    if (s->nranks <= 1) return;
    check_exchange(s, bytes_per_rank);
    if (s->comm) { rccl_allgather(s, bytes_per_rank); return; }       // RCCL: enqueued on the handle's stream, nothing to wait for here
    if (!s->ag_fn) throw Err(TNQS_ERR_COMM, "sharded handle without a transport (tnqs_set_sharding_rccl or tnqs_set_sharding)");
    HIPCHK(hipStreamSynchronize(s->stream));
    int rc = s->ag_fn(s->ag_ctx, s->exch, (int64_t)bytes_per_rank, s->nranks);
    if (rc != 0) throw Err(TNQS_ERR_COMM, "all-gather callback failed");
}
size_t round256(size_t b) { return (b + 255) & ~size_t(255); }

template <class T> static void apply_two_site_batch(State* s, const std::vector<Gate2>& gates, const tnqs_apply_opts& ao, double* errs) {
    if (gates.empty()) return;
    const Graph& g = *s->g;
    const size_t esz = s->esz();
    const bool sharded = s->nranks > 1;
    const double sqrt_cutoff = ao.sqrt_cutoff >= 0 ? ao.sqrt_cutoff : 10.0 * (s->dtype == TNQS_C64 ? 1.1920928955078125e-07 : 2.220446049250313e-16);
    const int ng = (int)gates.size();
    if (!ao.normalize_tensors) {       // without the final normalisation the result scales with the inputs: apply pending factors first
        std::vector<int> vs; for (auto& g2 : gates) { vs.push_back(g2.v1); vs.push_back(g2.v2); }
        materialize_scale(s, vs);
    }
    struct SiteJob { int v, other, bleg; bool owned; SD sd; std::vector<int> env_idx; std::vector<int> env_leg; };
    std::vector<SiteJob> sj(2 * (size_t)ng);
    std::vector<char> part(ng, 0);                  // this rank runs the small algebra of the gate
    // ---- 1. environments: sqrt(M) and projector for every incoming message of an owned site (utils.jl:18-27) ------
    struct EnvRec { int de; int n; void *H, *V, *msq, *prj; };      // views into one arena (env_arena): thousands of 16 KiB pool allocations per batch
                                                                    // were a third of the host time between a BP update and the first kernel of a batch
    std::vector<EnvRec> envs;
    for (int gi = 0; gi < ng; ++gi) {
        for (int side = 0; side < 2; ++side) {
            SiteJob& j = sj[2 * gi + side];
            j.v = side == 0 ? gates[gi].v1 : gates[gi].v2; j.other = side == 0 ? gates[gi].v2 : gates[gi].v1;
            j.sd = site_dims(s, j.v); j.bleg = g.leg(j.v, j.other); j.owned = s->owns(j.v);
            if (j.owned) part[gi] = 1;
            if (!j.owned) continue;
            for (int l = 0; l < j.sd.z; ++l) {
                if (l == j.bleg) continue;
                int de = g.dedge(g.nbr[j.v][l], j.v);
                if (!s->msg[de]) continue;                  // identity message: sqrt = I, nothing to absorb
                EnvRec r; r.de = de; r.n = j.sd.chi[l];
                j.env_idx.push_back((int)envs.size()); j.env_leg.push_back(l);
                envs.push_back(r);
            }
        }
    }
    std::vector<int> h_flags(2 * envs.size() + 2, 0);
    Buf d_flags = dalloc(s, h_flags.size() * sizeof(int));
    Buf env_arena;
    {
        std::vector<EnvItem> ei; std::vector<JacobiItem> ji; std::vector<EnvFinishItem> fi;
        size_t env_bytes = 0;
        for (auto& r : envs) { const size_t nn = (size_t)r.n * r.n; env_bytes += 2 * round256(nn * 16) + 2 * round256(nn * esz); }
        env_arena = dalloc(s, std::max<size_t>(256, env_bytes));
        char* ap = reinterpret_cast<char*>(env_arena->p);
        ei.reserve(envs.size()); ji.reserve(envs.size()); fi.reserve(envs.size());
        for (size_t i = 0; i < envs.size(); ++i) {
            EnvRec& r = envs[i]; size_t nn = (size_t)r.n * r.n;
            r.H = ap; ap += round256(nn * 16); r.V = ap; ap += round256(nn * 16); r.msq = ap; ap += round256(nn * esz); r.prj = ap; ap += round256(nn * esz);
            ei.push_back(EnvItem{s->msg[r.de]->p, r.H, r.V, r.n});
            ji.push_back(JacobiItem{r.H, r.V, r.n, r.n, nullptr});
            fi.push_back(EnvFinishItem{r.H, r.V, r.msq, r.prj, r.n, sqrt_cutoff, reinterpret_cast<int*>(d_flags->p) + 2 * i});
        }
        if (!envs.empty()) {
            const EnvItem* de = upload(s, ei); const JacobiItem* dj = upload(s, ji); const EnvFinishItem* df = upload(s, fi);
            { ProfScope ps(s, TNQS_PROF_SMALL, 0, 0); launch_env_prepare<T>(s->stream, de, (int)ei.size()); }
            size_t lds = 0; for (auto& r : envs) lds = std::max(lds, jacobi_lds_bytes(r.n, r.n, true, 16));
            { ProfScope ps(s, TNQS_PROF_JACOBI, 0, 0); launch_jacobi<double>(s->stream, dj, (int)ji.size(), 60, jacobi_lds(lds), mmax_of(ji)); }
            { ProfScope ps(s, TNQS_PROF_SMALL, 0, 0); launch_env_finish<T>(s->stream, df, (int)fi.size()); }
        }
    }
    // ---- 2. gauge: psi~ = psi x_outer M^{1/2}  (simple_update.jl:43-44), owned sites only -----------------------------
    std::vector<int> own_idx;                        // indices into sj of the owned sites
    for (size_t i = 0; i < sj.size(); ++i) if (sj[i].owned) own_idx.push_back((int)i);
    std::vector<Chain> chains(own_idx.size());
    for (size_t q = 0; q < own_idx.size(); ++q) {
        const SiteJob& j = sj[own_idx[q]];
        Chain& c = chains[q]; c.v = j.v; c.src = s->site[j.v]->p; c.sd = j.sd;
        for (size_t e = 0; e < j.env_idx.size(); ++e) c.steps.push_back({j.env_leg[e], envs[j.env_idx[e]].msq});
    }
    run_chains<T>(s, chains, TNQS_PROF_GATE_MODEPROD);
    // ---- 3. G = psi~^dagger psi~ over the outer legs, f64 accumulation (replaces the thin QR, simple_update.jl:45-48) --
    std::vector<GramJob> jobs;
    for (size_t q = 0; q < own_idx.size(); ++q) {
        const SiteJob& sjq = sj[own_idx[q]];
        GramJob j{}; j.X = chains[q].result; j.Y = chains[q].result; j.sd = sjq.sd; j.leg = sjq.bleg; j.keep_site = true;
        jobs.push_back(j);
    }
    run_grams<T, double>(s, jobs, TNQS_PROF_GATE_GRAM);
    std::vector<Buf> GA(sj.size()), GV(sj.size());
    auto nof = [&](size_t i) { return sj[i].sd.d * sj[i].sd.chi[sj[i].bleg]; };
    // G slots: in the sharded case every rank needs G1 and G2 of the gates it takes part in -> all-gather all of them (the same layout
    // serves the Gram matrices of the second factorisation pass further down)
    std::vector<size_t> slot(sj.size(), 0); size_t stride = 0;
    {
        std::vector<size_t> rank_bytes(s->nranks, 0);
        if (sharded) {
            for (size_t i = 0; i < sj.size(); ++i) { int r = s->owner[sj[i].v]; slot[i] = rank_bytes[r]; rank_bytes[r] += round256((size_t)nof(i) * nof(i) * 16); }
            for (size_t b : rank_bytes) stride = std::max(stride, b);
            check_exchange(s, stride);
        }
        std::vector<ReduceItem> ri; int elems = 0;
        for (size_t q = 0; q < own_idx.size(); ++q) {
            size_t i = own_idx[q]; int n = jobs[q].KK; size_t nn = (size_t)n * n;
            GA[i] = dalloc(s, nn * 16);
            void* dst = sharded ? (void*)(reinterpret_cast<char*>(s->exch) + (size_t)s->rank * stride + slot[i]) : GA[i]->p;
            ri.push_back(ReduceItem{jobs[q].partial->p, dst, (int)nn, jobs[q].nchunks, 1, elems}); elems += (int)nn;
        }
        const ReduceItem* dr = upload(s, ri);
        { ProfScope ps(s, TNQS_PROF_SMALL, 0, 0); launch_reduce<double, double>(s->stream, dr, (int)ri.size(), elems); }
        if (sharded) {
            exchange(s, stride);
            // one private copy of the gathered block (the exchange buffer is reused by the record exchange of this batch); the G of
            // every site this rank needs is a view into it
            Buf G_keep = dalloc(s, std::max<size_t>(256, stride * (size_t)s->nranks));
            HIPCHK(hipMemcpyAsync(G_keep->p, s->exch, stride * (size_t)s->nranks, hipMemcpyDeviceToDevice, s->stream));
            for (size_t i = 0; i < sj.size(); ++i) {
                if (!part[i / 2]) continue;
                size_t nn = (size_t)nof(i) * nof(i);
                GA[i] = sub_buffer(G_keep, (size_t)s->owner[sj[i].v] * stride + slot[i], nn * 16);
            }
        }
    }
    // R factor of psi~ = Q R from G = R^dagger R: Cholesky (R = L^dagger) where G has full rank by construction (at least as
    // many fibers as columns); the f64 Jacobi eigen factorisation R = Lambda^1/2 W^dagger otherwise, and for the whole batch when
    // a Cholesky pivot collapses (numerically rank-deficient G; the eigen path drops the null space, rank_tau in kernels.hpp)
    std::vector<Buf> GW(sj.size()); std::vector<char> is_chol(sj.size(), 0), is_small(sj.size(), 0);
    // ComplexF64, single rank: ill-conditioned sites get a second factorisation pass below, which sorts out what is signal and what is
    // noise among the smallest directions -- so the first pass keeps everything above the f64 noise floor instead of rank_tau
    const bool qr2 = !std::is_same<T, float>::value && use_qr2();
    auto tau_of = [&](int n) { return qr2 ? 1e-15 : rank_tau(std::is_same<T, float>::value, n); };
    // sites with fewer fibers than columns are factorised by their owner without a Gram matrix (small-SVD route) and never refined; the
    // criterion must not depend on ownership, every rank taking part in a gate has to reach the same decision
    auto small_shape = [&](size_t i) { const int n = nof(i); return sj[i].sd.n / (size_t)n < (size_t)n && n <= 256 && use_small_svd(); };
    std::vector<const void*> gauged_of(sj.size(), nullptr);      // psi~ of the owned sites
    for (size_t q = 0; q < own_idx.size(); ++q) gauged_of[own_idx[q]] = chains[q].result;
    Buf d_cholfail = dalloc(s, std::max<size_t>(1, sj.size()) * sizeof(int));      // one flag per site: only the sites whose pivot collapsed are redone
    std::vector<int> h_cholfail(sj.size(), 0);
    auto factor_G = [&](bool allow_chol, bool fallback = false) {
        std::vector<JacobiItem> ji, sji; std::vector<EnvItem> idn; std::vector<CholItem> ci; std::vector<SmallSvdItem> si; int cmax = 1;
        if (!fallback) HIPCHK(hipMemsetAsync(d_cholfail->p, 0, std::max<size_t>(1, sj.size()) * sizeof(int), s->stream));
        for (size_t i = 0; i < sj.size(); ++i) {
            if (!part[i / 2]) continue;
            if (fallback && !(is_chol[i] && h_cholfail[i])) continue;      // fallback pass: only the Cholesky sites whose pivot collapsed (the eigen sites are factorised, GA rotated in place)
            int n = nof(i);
            if (!GV[i]) GV[i] = dalloc(s, (size_t)n * n * 16);
            const size_t Nout = sj[i].sd.n / (size_t)n;
            const bool ch = allow_chol && n <= (use_chol128() ? 128 : 96) && Nout >= (size_t)n;
            is_chol[i] = ch ? 1 : 0;
            if (!ch && sj[i].owned && Nout < (size_t)n && n <= 256 && use_small_svd()) {
                // fewer fibers than columns: R = Sigma U^dagger straight from the SVD of the n x N matricised psi~ (no rank-deficient G)
                GW[i] = GV[i]; is_small[i] = 1;
                Buf M = dalloc(s, (size_t)n * Nout * 16); s->keepalive.push_back(M);
                const SD& sd = sj[i].sd; const int b = sj[i].bleg;
                si.push_back(SmallSvdItem{gauged_of[i], M->p, GA[i]->p, GV[i]->p, sd.d, (int)(sd.pre(b) / sd.d), sd.chi[b], (int)sd.post(b)});
                sji.push_back(JacobiItem{M->p, nullptr, n, (int)Nout, nullptr});
                continue;
            }
            if (ch) {
                GW[i] = dalloc(s, (size_t)n * n * 16);
                ci.push_back(CholItem{GA[i]->p, GV[i]->p, GW[i]->p, n, reinterpret_cast<int*>(d_cholfail->p) + i, tau_of(n)}); cmax = std::max(cmax, n);
            } else {
                GW[i] = GV[i];
                idn.push_back(EnvItem{nullptr, GV[i]->p, GV[i]->p, n});      // msg == null: H := I, V := I (same buffer)
                ji.push_back(JacobiItem{GA[i]->p, GV[i]->p, n, n, nullptr});
            }
        }
        if (!si.empty()) {
            const SmallSvdItem* ds = upload(s, si); const JacobiItem* dj = upload(s, sji);
            ProfScope ps(s, TNQS_PROF_JACOBI, 0, 0);
            launch_small_svd_prepare<T>(s->stream, ds, (int)si.size());
            size_t lds = 0; for (auto& j : sji) lds = std::max(lds, jacobi_lds_bytes(j.m, j.n, false, 16));
            launch_jacobi<double>(s->stream, dj, (int)sji.size(), 60, jacobi_lds(lds), mmax_of(sji));
            launch_small_svd_finish(s->stream, ds, (int)si.size());
        }
        if (!ci.empty()) {      // n <= 96: square LDS array; 96 < n <= 128 (chi = 64 sites): packed triangle
            const CholItem* dc = upload(s, ci); ProfScope ps(s, TNQS_PROF_JACOBI, 0, 0);
            if (cmax <= 96) launch_chol(s->stream, dc, (int)ci.size(), cmax); else launch_chol_packed(s->stream, dc, (int)ci.size(), cmax);
        }
        if (!ji.empty()) {
            const EnvItem* di = upload(s, idn);
            { ProfScope ps(s, TNQS_PROF_SMALL, 0, 0); launch_env_prepare<T>(s->stream, di, (int)idn.size()); }
            const JacobiItem* dj = upload(s, ji);
            size_t lds = 0; for (auto& j : ji) lds = std::max(lds, jacobi_lds_bytes(j.n, j.n, true, 16));
            { ProfScope ps(s, TNQS_PROF_JACOBI, 0, 0); launch_jacobi<double>(s->stream, dj, (int)ji.size(), 60, jacobi_lds(lds), mmax_of(ji)); }
        }
    };
    factor_G(use_chol());
    // ---- 4. theta = gate . (R1 R2), SVD, truncation, X1 / X2  (simple_update.jl:51-59) -----------------------------
    struct GateWS { Buf lam1, lam2, idx1, idx2, theta, thetaV, theta0, X1, X2, S, lowA, lowB, lowG, lowL, lowW; int n1, n2, chi, cap; };
    std::vector<GateWS> ws(ng);
    std::vector<int> pg;                              // gates this rank takes part in
    for (int gi = 0; gi < ng; ++gi) if (part[gi]) pg.push_back(gi);
    std::vector<GateItem> gitems(pg.size());
    int cap_max = 1; size_t x2_max = 0;
    for (int gi = 0; gi < ng; ++gi) {
        GateWS& w = ws[gi];
        const SiteJob& a = sj[2 * gi]; const SiteJob& b = sj[2 * gi + 1];
        int chi = a.sd.chi[a.bleg];
        w.n1 = a.sd.d * chi; w.n2 = b.sd.d * chi; w.chi = chi;
        int Mr = w.n1 * a.sd.d, Nc = w.n2 * b.sd.d;
        if (Mr > 256 || Nc > 256) throw Err(TNQS_ERR_UNSUPPORTED, "two-site gate: d^2*chi > 256 is not supported by the Jacobi SVD kernel yet");
        int cap = std::min(Mr, Nc); if (ao.maxdim > 0) cap = std::min(cap, ao.maxdim);
        w.cap = cap; cap_max = std::max(cap_max, cap);
        x2_max = std::max(x2_max, (size_t)w.n2 * b.sd.d * cap * esz);
    }
    // SVD of theta: the right factor is never accumulated from the rotations (in f32 its orthogonality degrades with the
    // rotation count, ~1e-5 at 150 columns) but recovered from an unrotated copy: V = theta0^dagger (U S) S^-2
    const bool theta0_used = true;
    {
        std::vector<char> raw;
        std::vector<size_t> off(pg.size()), offA(pg.size(), 0), offB(pg.size(), 0); std::vector<int> kappa(pg.size(), 0);
        const bool lowrank_on = std::is_same<T, float>::value && use_lowrank();
        for (size_t q = 0; q < pg.size(); ++q) {
            int gi = pg[q];
            const int d1 = s->d[gates[gi].v1], d2 = s->d[gates[gi].v2];
            int dd = d1 * d2;
            off[q] = raw.size();
            const char* p = reinterpret_cast<const char*>(gates[gi].mat);
            raw.insert(raw.end(), p, p + (size_t)dd * dd * 16);
            if (!lowrank_on) continue;
            // the gate as an operator sum g = sum_k a_k (x) b_k: O[(s1',s1),(s2',s2)] = g[(s1' s2'),(s1 s2)] factorised by elimination with
            // complete pivoting (exact rank factorisation; kappa = operator Schmidt rank: 2 for Rzz / Rxx / CNOT / CPHASE, 4 for SWAP)
            const int na = d1 * d1, nb = d2 * d2;
            std::vector<std::complex<double>> O((size_t)na * nb), fa, fb;
            const std::complex<double>* gm = reinterpret_cast<const std::complex<double>*>(gates[gi].mat);
            double amax = 0;
            for (int s1p = 0; s1p < d1; ++s1p) for (int s1 = 0; s1 < d1; ++s1) for (int s2p = 0; s2p < d2; ++s2p) for (int s2 = 0; s2 < d2; ++s2) {
                auto v = gm[(s1p * d2 + s2p) + (size_t)dd * (s1 * d2 + s2)];
                O[(s1p + d1 * s1) + (size_t)na * (s2p + d2 * s2)] = v; amax = std::max(amax, std::abs(v));
            }
            int kp = 0;
            for (; kp < std::min(na, nb); ++kp) {
                int pi = 0, pj = 0; double best = 0;
                for (int j = 0; j < nb; ++j) for (int i = 0; i < na; ++i) { double a = std::abs(O[i + (size_t)na * j]); if (a > best) { best = a; pi = i; pj = j; } }
                if (!(best > 1e-13 * amax)) break;
                const std::complex<double> piv = O[pi + (size_t)na * pj];
                std::vector<std::complex<double>> col(na), row(nb);
                for (int i = 0; i < na; ++i) col[i] = O[i + (size_t)na * pj];
                for (int j = 0; j < nb; ++j) row[j] = O[pi + (size_t)na * j] / piv;
                for (int j = 0; j < nb; ++j) for (int i = 0; i < na; ++i) O[i + (size_t)na * j] -= col[i] * row[j];
                fa.insert(fa.end(), col.begin(), col.end()); fb.insert(fb.end(), row.begin(), row.end());
            }
            kappa[q] = kp;
            offA[q] = raw.size(); raw.insert(raw.end(), reinterpret_cast<const char*>(fa.data()), reinterpret_cast<const char*>(fa.data()) + fa.size() * 16);
            offB[q] = raw.size(); raw.insert(raw.end(), reinterpret_cast<const char*>(fb.data()), reinterpret_cast<const char*>(fb.data()) + fb.size() * 16);
        }
        const char* d_gm = pg.empty() ? nullptr : upload(s, raw);
        for (size_t q = 0; q < pg.size(); ++q) {
            int gi = pg[q];
            GateWS& w = ws[gi]; GateItem& it = gitems[q];
            const SiteJob& a = sj[2 * gi]; const SiteJob& b = sj[2 * gi + 1];
            int Mr = w.n1 * a.sd.d, Nc = w.n2 * b.sd.d, cap = w.cap;
            w.lam1 = dalloc(s, w.n1 * 8); w.lam2 = dalloc(s, w.n2 * 8); w.idx1 = dalloc(s, w.n1 * 4); w.idx2 = dalloc(s, w.n2 * 4);
            w.theta = dalloc(s, (size_t)Mr * Nc * esz); w.thetaV = dalloc(s, (size_t)std::max(Mr, Nc) * std::max(Mr, Nc) * esz);
            if (theta0_used) w.theta0 = dalloc(s, (size_t)Mr * Nc * esz);
            w.X1 = dalloc(s, (size_t)w.n1 * a.sd.d * cap * esz); w.X2 = dalloc(s, (size_t)w.n2 * b.sd.d * cap * esz);
            w.S = dalloc(s, cap * 8);
            it.GA1 = GA[2 * gi]->p; it.GV1 = GV[2 * gi]->p; it.GA2 = GA[2 * gi + 1]->p; it.GV2 = GV[2 * gi + 1]->p;
            it.GW1 = GW[2 * gi]->p; it.GW2 = GW[2 * gi + 1]->p; it.chol1 = is_chol[2 * gi]; it.chol2 = is_chol[2 * gi + 1];
            it.n1 = w.n1; it.n2 = w.n2; it.d1 = a.sd.d; it.d2 = b.sd.d; it.chi = w.chi;
            it.gate = reinterpret_cast<const double*>(d_gm + off[q]);
            it.kappa = 0; it.opA = it.opB = nullptr; it.lowA = it.lowB = it.lowG = nullptr; it.lowL = nullptr; it.lowfail = nullptr;
            {   // low-rank route of the theta SVD (GateItem): only where it can apply -- K = kappa chi below the theta columns and chol_kernel's size
                const int K = kappa[q] * w.chi;
                if (lowrank_on && kappa[q] > 0 && K < Nc && K <= 128 && cap <= K && Mr >= Nc) {
                    w.lowA = dalloc(s, (size_t)Mr * K * 16); w.lowB = dalloc(s, (size_t)Nc * K * 16); w.lowG = dalloc(s, (size_t)K * K * 16);
                    w.lowL = dalloc(s, (size_t)K * K * 16); w.lowW = dalloc(s, (size_t)K * K * 16);
                    it.kappa = kappa[q]; it.opA = reinterpret_cast<const double*>(d_gm + offA[q]); it.opB = reinterpret_cast<const double*>(d_gm + offB[q]);
                    it.lowA = w.lowA->p; it.lowB = w.lowB->p; it.lowG = w.lowG->p; it.lowL = w.lowL->p;
                }
            }
            it.lam1 = (double*)w.lam1->p; it.lam2 = (double*)w.lam2->p; it.idx1 = (int*)w.idx1->p; it.idx2 = (int*)w.idx2->p;
            it.theta = w.theta->p; it.thetaV = w.thetaV->p; it.theta0 = w.theta0 ? w.theta0->p : nullptr; it.X1 = w.X1->p; it.X2 = w.X2->p; it.S = (double*)w.S->p;
            it.maxdim = ao.maxdim; it.cutoff = ao.cutoff; it.normalize = ao.normalize_tensors; it.chi_cap = cap;
            // second-pass mode: the eigen route of the first pass is shifted (negative tau, gate_eigs) -- it must not drop a direction the
            // second pass could still resolve
            // (the small-SVD sites are factorised without a Gram matrix and are never refined: ordinary threshold)
            auto site_tau = [&](size_t i, int n) { return (qr2 && !small_shape(i)) ? -rank_tau(false, n) : rank_tau(std::is_same<T, float>::value, n); };
            it.tau1 = site_tau(2 * (size_t)gi, w.n1); it.tau2 = site_tau(2 * (size_t)gi + 1, w.n2); it.rk1 = nullptr; it.rk2 = nullptr;
        }
    }
    const int npg = (int)pg.size();
    // per-gate (r1, r2, chi', status, sweeps, wide, -, -) and truncation error live in two contiguous arrays: one D2H each
    Buf d_info_all = dalloc(s, std::max<size_t>(1, (size_t)npg * 32));
    Buf d_terr_all = dalloc(s, std::max<size_t>(1, (size_t)npg * 8));
    HIPCHK(hipMemsetAsync(d_info_all->p, 0, std::max<size_t>(1, (size_t)npg * 32), s->stream));
    for (int q = 0; q < npg; ++q) { gitems[q].info = reinterpret_cast<int*>(d_info_all->p) + 8 * q; gitems[q].truncerr = reinterpret_cast<double*>(d_terr_all->p) + q; }
    // low-rank route: one failure flag per gate for the Cholesky factorisation of B^dagger B
    Buf d_lowfail = dalloc(s, std::max<size_t>(1, (size_t)npg * sizeof(int)));
    Buf d_texp = dalloc(s, std::max<size_t>(1, (size_t)npg * sizeof(int)));
    for (int q = 0; q < npg; ++q) { gitems[q].lowfail = reinterpret_cast<const int*>(d_lowfail->p) + q; gitems[q].texp = reinterpret_cast<int*>(d_texp->p) + q; }
    const GateItem* d_gitems = upload(s, gitems);
    auto run_theta = [&]() {
        HIPCHK(hipMemsetAsync(d_info_all->p, 0, std::max<size_t>(1, (size_t)npg * 32), s->stream));
        HIPCHK(hipMemsetAsync(d_lowfail->p, 0, std::max<size_t>(1, (size_t)npg * sizeof(int)), s->stream));
        ProfScope ps(s, TNQS_PROF_SMALL, 0, 0);
        launch_gate_theta<T>(s->stream, d_gitems, npg);
        std::vector<CholItem> lc; int kmax = 1;
        for (int q = 0; q < npg; ++q) {
            if (!gitems[q].lowG) continue;
            const int K = gitems[q].kappa * gitems[q].chi;
            lc.push_back(CholItem{gitems[q].lowG, const_cast<void*>(gitems[q].lowL), K <= 96 ? ws[pg[q]].lowW->p : nullptr, K, reinterpret_cast<int*>(d_lowfail->p) + q, rank_tau(true, K)});
            kmax = std::max(kmax, K);
        }
        if (!lc.empty()) {
            const CholItem* dc = upload(s, lc); launch_lowrank_g(s->stream, d_gitems, npg);
            if (kmax <= 96) launch_chol(s->stream, dc, (int)lc.size(), kmax); else launch_chol_packed(s->stream, dc, (int)lc.size(), kmax);      // only L is used here
            launch_lowrank_m(s->stream, d_gitems, npg);
        }
        launch_theta_scale<T>(s->stream, d_gitems, npg);       // theta (or M) and theta0 to O(1), exponent kept per gate for gate_finish
    };
    run_theta();
    std::vector<int> info(8 * (size_t)ng, 0); std::vector<double> terr(ng, 0.0);
    {
        // theta dims depend on the ranks found on the device: read them back (also where message-eigenvalue errors surface)
        std::vector<int> hinfo(8 * (size_t)std::max(1, npg));
        int chol_failed = 0;
        if (npg) HIPCHK(hipMemcpyAsync(hinfo.data(), d_info_all->p, (size_t)npg * 32, hipMemcpyDeviceToHost, s->stream));
        if (!envs.empty()) HIPCHK(hipMemcpyAsync(h_flags.data(), d_flags->p, 2 * envs.size() * sizeof(int), hipMemcpyDeviceToHost, s->stream));
        if (!sj.empty()) HIPCHK(hipMemcpyAsync(h_cholfail.data(), d_cholfail->p, sj.size() * sizeof(int), hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        for (size_t i = 0; i < sj.size(); ++i) chol_failed += (part[i / 2] && is_chol[i] && h_cholfail[i]) ? 1 : 0;
        if (chol_failed) {              // numerically rank-deficient Gram matrix somewhere in the batch: redo with the eigen path
            factor_G(false, true);
            for (int q = 0; q < npg; ++q) { int gi = pg[q]; GateItem& it = gitems[q]; it.GW1 = GW[2 * gi]->p; it.GW2 = GW[2 * gi + 1]->p; it.chol1 = is_chol[2 * gi]; it.chol2 = is_chol[2 * gi + 1]; }
            d_gitems = upload(s, gitems);
            run_theta();
            if (npg) HIPCHK(hipMemcpyAsync(hinfo.data(), d_info_all->p, (size_t)npg * 32, hipMemcpyDeviceToHost, s->stream));
            HIPCHK(hipStreamSynchronize(s->stream));
            s->stats.n_chol_fallbacks += 1;
        }
        if (qr2) {
            // ---- second factorisation pass (CholeskyQR2) of the sites gate_theta flagged as ill-conditioned: a Gram matrix resolves the
            // singular directions of psi~ only down to sigma_rel ~ 1e-7, the reference's QR to eps.  Q1 = psi~ R1^+ is formed explicitly;
            // its Gram matrix is close to the identity on everything the first pass resolved and shows the true weight of what it did not,
            // so R = R2 R1 is as accurate as a Householder R.  
            // Sharded: the owner of a site forms Q1 and its Gram matrix, one more all-gather (same slots as the first Gram exchange, issued
            // by every rank whether or not it has a flagged site -- it is a collective) hands it to the partner rank, and both compose the
            // same factor from the same inputs.
            std::vector<size_t> rs; std::vector<int> rq;
            for (int q = 0; q < npg; ++q) for (int side = 0; side < 2; ++side) {
                const size_t i = 2 * (size_t)pg[q] + side;
                static const bool all = [] { const char* v = std::getenv("TNQS_QR2_ALL"); return v && v[0] == '1'; }();      // debug: refine every site
                if ((all || ((hinfo[8 * q + 6] >> side) & 1)) && !small_shape(i)) { rs.push_back(i); rq.push_back(q); }
            }
            if (sharded || !rs.empty()) {
                const size_t m = rs.size();
                std::vector<Buf> X1(m), Q1(m), G2(m), V2(m), GVn(m), GWn(m); Buf d_rk = dalloc(s, std::max<size_t>(1, m) * sizeof(int));
                std::vector<Qr2RinvItem> ri; std::vector<FiberItem> fi; std::vector<GramJob> gj; std::vector<size_t> own_k; size_t KKmax = 1; int tiles = 0;
                for (size_t k = 0; k < m; ++k) {
                    const size_t i = rs[k]; const int q = rq[k]; const bool second = (i & 1) != 0; const int n = nof(i); const size_t nn = (size_t)n * n;
                    X1[k] = dalloc(s, nn * 16); V2[k] = dalloc(s, nn * 16); GVn[k] = dalloc(s, nn * 16); GWn[k] = dalloc(s, nn * 16);
                    ri.push_back(Qr2RinvItem{GW[i]->p, second ? gitems[q].lam2 : gitems[q].lam1, second ? gitems[q].idx2 : gitems[q].idx1, gitems[q].info + (second ? 1 : 0), n, X1[k]->p});
                    if (sj[i].owned) { own_k.push_back(k); KKmax = std::max<size_t>(KKmax, (size_t)n); Q1[k] = dalloc(s, sj[i].sd.n * esz); }
                }
                const int TR = pick_TR(KKmax, esz, 1);
                for (size_t k : own_k) {
                    const size_t i = rs[k]; const SiteJob& j = sj[i]; const int chi = j.sd.chi[j.bleg];
                    FiberItem it{}; it.in = gauged_of[i]; it.out = Q1[k]->p; it.X = X1[k]->p;
                    it.D = j.sd.d; it.PA = (int)(j.sd.pre(j.bleg) / j.sd.d); it.K = chi; it.PB = (int)j.sd.post(j.bleg); it.Do = j.sd.d; it.No = chi;
                    tile_params(it.PA, it.PB, TR, it.TA, it.TB, it.nta, it.ntb); it.tpw = 1; it.tile_begin = tiles; it.want_norm = 0;
                    tiles += it.nta * it.ntb; fi.push_back(it);
                    GramJob g2{}; g2.X = Q1[k]->p; g2.Y = Q1[k]->p; g2.sd = j.sd; g2.leg = j.bleg; g2.keep_site = true; gj.push_back(g2);
                }
                if (m) { const Qr2RinvItem* d = upload(s, ri); ProfScope ps(s, TNQS_PROF_SMALL, 0, 0); launch_qr2_rinv(s->stream, d, (int)m); }
                if (!fi.empty()) {
                    Buf np = dalloc(s, std::max(1, tiles) * sizeof(double)); const FiberItem* d = upload(s, fi);
                    { ProfScope ps(s, TNQS_PROF_GATE_APPLY, 0, 0); launch_fiber_gemm<T>(s->stream, d, (int)fi.size(), tiles, TR, (int)KKmax, reinterpret_cast<double*>(np->p)); }
                    s->keepalive.push_back(np);
                    run_grams<T, double>(s, gj, TNQS_PROF_GATE_GRAM);
                    std::vector<ReduceItem> rd; int elems = 0;
                    for (size_t t = 0; t < own_k.size(); ++t) {
                        const size_t k = own_k[t], i = rs[k]; const int nn = gj[t].KK * gj[t].KK;
                        void* dst;
                        if (sharded) dst = reinterpret_cast<char*>(s->exch) + (size_t)s->rank * stride + slot[i];
                        else { G2[k] = dalloc(s, (size_t)nn * 16); dst = G2[k]->p; }
                        rd.push_back(ReduceItem{gj[t].partial->p, dst, nn, gj[t].nchunks, 1, elems}); elems += nn;
                    }
                    const ReduceItem* d2 = upload(s, rd); ProfScope ps(s, TNQS_PROF_SMALL, 0, 0); launch_reduce<double, double>(s->stream, d2, (int)rd.size(), elems);
                }
                if (sharded) {
                    exchange(s, stride);
                    Buf G2_keep = dalloc(s, std::max<size_t>(256, stride * (size_t)s->nranks));
                    HIPCHK(hipMemcpyAsync(G2_keep->p, s->exch, stride * (size_t)s->nranks, hipMemcpyDeviceToDevice, s->stream));
                    for (size_t k = 0; k < m; ++k) { const size_t i = rs[k]; const size_t nn = (size_t)nof(i) * nof(i); G2[k] = sub_buffer(G2_keep, (size_t)s->owner[sj[i].v] * stride + slot[i], nn * 16); }
                }
                if (m) {
                    std::vector<EnvItem> idn; std::vector<JacobiItem> ji; size_t lds = 0;
                    for (size_t k = 0; k < m; ++k) { const int n = nof(rs[k]); idn.push_back(EnvItem{nullptr, V2[k]->p, V2[k]->p, n}); ji.push_back(JacobiItem{G2[k]->p, V2[k]->p, n, n, nullptr}); lds = std::max(lds, jacobi_lds_bytes(n, n, true, 16)); }
                    const EnvItem* di = upload(s, idn); const JacobiItem* dj = upload(s, ji);
                    { ProfScope ps(s, TNQS_PROF_SMALL, 0, 0); launch_env_prepare<T>(s->stream, di, (int)m); }
                    { ProfScope ps(s, TNQS_PROF_JACOBI, 0, 0); launch_jacobi<double>(s->stream, dj, (int)m, 60, jacobi_lds(lds), mmax_of(ji)); }
                    std::vector<Qr2ComposeItem> ci;
                    for (size_t k = 0; k < m; ++k) {
                        const size_t i = rs[k]; const int q = rq[k]; const bool second = (i & 1) != 0; const int n = nof(i);
                        ci.push_back(Qr2ComposeItem{G2[k]->p, V2[k]->p, X1[k]->p, GV[i]->p, second ? gitems[q].lam2 : gitems[q].lam1, second ? gitems[q].idx2 : gitems[q].idx1,
                                                    gitems[q].info + (second ? 1 : 0), n, rank_tau(false, n), GVn[k]->p, GWn[k]->p, reinterpret_cast<int*>(d_rk->p) + k});
                    }
                    { const Qr2ComposeItem* d = upload(s, ci); ProfScope ps(s, TNQS_PROF_SMALL, 0, 0); launch_qr2_compose(s->stream, d, (int)m); }
                    for (size_t k = 0; k < m; ++k) {
                        const size_t i = rs[k]; GateItem& it = gitems[rq[k]];
                        GV[i] = GVn[k]; GW[i] = GWn[k]; s->keepalive.push_back(X1[k]); if (Q1[k]) s->keepalive.push_back(Q1[k]); s->keepalive.push_back(G2[k]); s->keepalive.push_back(V2[k]);
                        if (i & 1) { it.GV2 = GV[i]->p; it.GW2 = GW[i]->p; it.chol2 = 2; it.rk2 = reinterpret_cast<int*>(d_rk->p) + k; }
                        else { it.GV1 = GV[i]->p; it.GW1 = GW[i]->p; it.chol1 = 2; it.rk1 = reinterpret_cast<int*>(d_rk->p) + k; }
                    }
                    s->keepalive.push_back(d_rk);
                    d_gitems = upload(s, gitems);
                    // gate_theta reads the first-pass (lambda, idx, r) of the untouched partner site again and overwrites them with the same values
                    run_theta();
                    if (npg) HIPCHK(hipMemcpyAsync(hinfo.data(), d_info_all->p, (size_t)npg * 32, hipMemcpyDeviceToHost, s->stream));
                    HIPCHK(hipStreamSynchronize(s->stream));
                    for (size_t k = 0; k < m; ++k) s->stats.n_qr2_sites += sj[rs[k]].owned ? 1 : 0;
                }
            }
        }
        for (size_t i = 0; i < envs.size(); ++i)
            if (h_flags[2 * i + 1]) throw Err(TNQS_ERR_NUMERIC, "simple_update: incoming message has a negative eigenvalue above sqrt_cutoff (DomainError in the reference, src/utils.jl:21)");
        std::vector<JacobiItem> ji; std::vector<int> ncfull;
        for (int q = 0; q < npg; ++q) {
            int gi = pg[q];
            int r1 = hinfo[8 * q], r2 = hinfo[8 * q + 1];
            int Mr = r1 * gitems[q].d1, Nc = r2 * gitems[q].d2;
            if (Mr < Nc) std::swap(Mr, Nc);        // wide theta is stored as its adjoint (gate_theta_kernel)
            const int ncolJ = hinfo[8 * q + 7] > 0 ? hinfo[8 * q + 7] : Nc;      // low-rank route: the SVD runs on M (Mr x K), same U and Sigma
            ji.push_back(JacobiItem{ws[gi].theta->p, ws[gi].thetaV->p, Mr, ncolJ, gitems[q].info + 4});
            ncfull.push_back(Nc); s->stats.n_lowrank_svd += (ncolJ < Nc) ? 1 : 0;
        }
        // LDS residency: A and V if both fit; A only (V recovered from the unrotated copy) if only A fits; else global memory
        size_t lds_av = 0, lds_a = 0;
        for (auto& j : ji) { lds_av = std::max(lds_av, jacobi_lds_bytes(j.m, j.n, true, esz)); lds_a = std::max(lds_a, jacobi_lds_bytes(j.m, j.n, false, esz)); }
        const bool novee = theta0_used;
        if (novee) for (auto& j : ji) j.V = nullptr;
        (void)lds_av; (void)lds_a;
        { ProfScope ps(s, TNQS_PROF_JACOBI, 0, 0); svd_batch<T>(s, ji, !novee); }
        if (novee) {
            std::vector<RecoverItem> rv;
            for (int q = 0; q < npg; ++q) rv.push_back(RecoverItem{ws[pg[q]].theta0->p, ws[pg[q]].theta->p, ws[pg[q]].thetaV->p, ji[q].m, ncfull[q], ji[q].n});
            const RecoverItem* dr = upload(s, rv);
            { ProfScope ps(s, TNQS_PROF_JACOBI, 0, 0); { int nmax = 1; for (int nc : ncfull) nmax = std::max(nmax, nc); if (std::is_same<T, float>::value && use_mfma()) launch_recover_v_mfma(s->stream, dr, npg, nmax); else launch_recover_v<T>(s->stream, dr, npg, nmax); } }
        }
        { ProfScope ps(s, TNQS_PROF_SMALL, 0, 0); launch_gate_finish<T>(s->stream, d_gitems, npg); }
        std::vector<double> hterr(std::max(1, npg));
        if (npg) {
            HIPCHK(hipMemcpyAsync(hinfo.data(), d_info_all->p, (size_t)npg * 32, hipMemcpyDeviceToHost, s->stream));
            HIPCHK(hipMemcpyAsync(hterr.data(), d_terr_all->p, (size_t)npg * 8, hipMemcpyDeviceToHost, s->stream));
        }
        HIPCHK(hipStreamSynchronize(s->stream));
        for (int q = 0; q < npg; ++q) { for (int k = 0; k < 8; ++k) info[8 * pg[q] + k] = hinfo[8 * q + k]; terr[pg[q]] = hterr[q]; }
    }
    // ---- 4b. sharded: the owner of the first vertex publishes (chi', status, truncerr, S, X2) of each gate ----------------
    std::vector<const double*> Sptr(ng, nullptr);
    Buf S_keep;
    if (sharded) {
        const size_t slot_bytes = round256(32 + (size_t)cap_max * 8 + x2_max);
        std::vector<size_t> slot(ng, 0); std::vector<size_t> rank_bytes(s->nranks, 0);
        for (int gi = 0; gi < ng; ++gi) { int r = s->owner[gates[gi].v1]; slot[gi] = rank_bytes[r]; rank_bytes[r] += slot_bytes; }
        size_t stride = 0; for (size_t b : rank_bytes) stride = std::max(stride, b);
        check_exchange(s, stride);
        char* base = reinterpret_cast<char*>(s->exch);
        {   // pack the records of the gates whose first vertex is ours (one launch)
            std::vector<RecordPackItem> rp;
            std::vector<int> qof(ng, -1); for (int q = 0; q < npg; ++q) qof[pg[q]] = q;
            for (int gi = 0; gi < ng; ++gi) {
                if (s->owner[gates[gi].v1] != s->rank) continue;
                const SiteJob& b = sj[2 * gi + 1]; const int q = qof[gi];
                rp.push_back(RecordPackItem{base + (size_t)s->rank * stride + slot[gi], gitems[q].info, gitems[q].truncerr, reinterpret_cast<const double*>(ws[gi].S->p),
                                            ws[gi].cap, ws[gi].X2->p, (long long)((size_t)ws[gi].n2 * b.sd.d * ws[gi].cap * esz / 8), (long long)(32 + (size_t)cap_max * 8)});
            }
            if (!rp.empty()) { const RecordPackItem* d = upload(s, rp); launch_record_pack(s->stream, d, (int)rp.size()); }
        }
        exchange(s, stride);
        // keep a private copy of the gathered block: the exchange buffer is reused by the next batch
        S_keep = dalloc(s, std::max<size_t>(256, stride * (size_t)s->nranks));
        HIPCHK(hipMemcpyAsync(S_keep->p, base, stride * (size_t)s->nranks, hipMemcpyDeviceToDevice, s->stream));
        std::vector<double> allhdr(4 * (size_t)std::max(1, ng));
        {   // all headers in one gather + one D2H
            std::vector<const void*> srcs(ng);
            for (int gi = 0; gi < ng; ++gi) srcs[gi] = reinterpret_cast<char*>(S_keep->p) + (size_t)s->owner[gates[gi].v1] * stride + slot[gi];
            Buf d_hdr = dalloc(s, (size_t)std::max(1, ng) * 32);
            const void* const* d_srcs = upload(s, srcs);
            launch_header_gather(s->stream, d_srcs, ng, reinterpret_cast<double*>(d_hdr->p));
            if (ng) HIPCHK(hipMemcpyAsync(allhdr.data(), d_hdr->p, (size_t)ng * 32, hipMemcpyDeviceToHost, s->stream));
            HIPCHK(hipStreamSynchronize(s->stream));
        }
        for (int gi = 0; gi < ng; ++gi) {
            info[8 * gi + 2] = (int)allhdr[4 * gi]; info[8 * gi + 3] = (int)allhdr[4 * gi + 1]; terr[gi] = allhdr[4 * gi + 2];
            const size_t off = (size_t)s->owner[gates[gi].v1] * stride + slot[gi];
            Sptr[gi] = reinterpret_cast<const double*>(reinterpret_cast<const char*>(S_keep->p) + off + 32);
            const SiteJob& b = sj[2 * gi + 1];
            if (b.owned && s->owner[gates[gi].v1] != s->rank)          // the partner rank computed the SVD: its X2 is used in place (a view)
                ws[gi].X2 = sub_buffer(S_keep, off + 32 + (size_t)cap_max * 8, (size_t)ws[gi].n2 * b.sd.d * ws[gi].cap * esz);
        }
    } else {
        for (int gi = 0; gi < ng; ++gi) Sptr[gi] = (const double*)ws[gi].S->p;
    }
    // every gate's status is checked before anything of the handle is replaced: a failing batch leaves the state as it was
    for (int gi = 0; gi < ng; ++gi) if (info[8 * gi + 3] != 0) throw Err(TNQS_ERR_NUMERIC, "simple_update: internal bond capacity exceeded");
    // ---- 5. psi' = (psi x_outer P) x_(s,b) X  (simple_update.jl:62-64, net effect of gauge + ungauge) ----------------
    std::vector<Chain> pch(own_idx.size());
    for (size_t q = 0; q < own_idx.size(); ++q) {
        const SiteJob& j = sj[own_idx[q]];
        Chain& c = pch[q]; c.v = j.v; c.src = s->site[j.v]->p; c.sd = j.sd;
        for (size_t e = 0; e < j.env_idx.size(); ++e)
            if (!h_flags[2 * j.env_idx[e]]) c.steps.push_back({j.env_leg[e], envs[j.env_idx[e]].prj});  // rank-deficient message only
    }
    run_chains<T>(s, pch, TNQS_PROF_GATE_MODEPROD);
    if (!own_idx.empty()) {
        std::vector<FiberItem> items; std::vector<int> verts, tb, nt; std::vector<Buf> outs; std::vector<size_t> ne;
        int tiles = 0; size_t KKmax = 1, NNmax = 1; double bytes = 0, flops = 0;
        for (size_t q = 0; q < own_idx.size(); ++q) {
            size_t i = own_idx[q];
            KKmax = std::max<size_t>(KKmax, (size_t)sj[i].sd.d * sj[i].sd.chi[sj[i].bleg]);
            NNmax = std::max<size_t>(NNmax, (size_t)sj[i].sd.d * info[8 * (i / 2) + 2]);
        }
        int TR = pick_TR(KKmax, esz, 1);
        bool mf = false;
        if (std::is_same<T, float>::value && use_mfma() && KKmax >= 8) { int t = mfma_fiber_tile_rows((int)KKmax, (int)NNmax); if (t > 0) { TR = t; mf = true; } }
        // plane kernel for the common shape d = 2, chi_b = chi_b' = 32 (pair-kernel geometry, two waves per SIMD)
        std::vector<Apply64Item> a64; std::vector<XbItem> xbi; std::vector<int> a64_verts; std::vector<Buf> a64_outs; std::vector<size_t> a64_ne;
        std::vector<char> via64(own_idx.size(), 0); double a64_slices = 0;
        if (std::is_same<T, float>::value && use_mfma() && use_apply64()) {
            for (size_t q = 0; q < own_idx.size(); ++q) {
                size_t i = own_idx[q]; int gi = (int)i / 2; const SiteJob& j = sj[i];
                Apply64Item it{};
                if (info[8 * gi + 2] != 32 || j.sd.chi[j.bleg] != 32 || !apply64_geometry(j.sd.d, j.sd.z, j.sd.chi.data(), j.bleg, it.g)) continue;
                Buf out = dalloc(s, j.sd.n * esz); Buf xb = dalloc(s, 2048 * 16); s->keepalive.push_back(xb);
                it.in = pch[q].result; it.out = out->p; it.Xb = xb->p;
                xbi.push_back(XbItem{(i & 1) ? ws[gi].X2->p : ws[gi].X1->p, xb->p});
                a64.push_back(it); a64_verts.push_back(j.v); a64_outs.push_back(out); a64_ne.push_back(j.sd.n);
                a64_slices += (double)j.sd.n / 16384.0; via64[q] = 1;
            }
        }
        if (!a64.empty()) {
            const int spw = (int)std::max(1.0, std::min(8.0, a64_slices / 2048.0));
            std::vector<int> tb64, nt64; int wgs = 0;
            for (auto& it : a64) { int nwg = (it.g.n0 * it.g.n1 * it.g.n2 + spw - 1) / spw; it.spw = spw; it.wg_begin = wgs; tb64.push_back(wgs); nt64.push_back(nwg); wgs += nwg; }
            Buf np64 = dalloc(s, (size_t)wgs * sizeof(double));
            for (size_t k = 0; k < a64.size(); ++k) a64[k].norm_partial = ao.normalize_tensors ? reinterpret_cast<double*>(np64->p) + tb64[k] : nullptr;
            const XbItem* dx = upload(s, xbi); const Apply64Item* da = upload(s, a64);
            { ProfScope ps(s, TNQS_PROF_SMALL, 0, 0); launch_make_xb(s->stream, dx, (int)xbi.size()); }
            { ProfScope ps(s, TNQS_PROF_GATE_APPLY, 2.0 * a64_slices * 16384.0 * esz, 8.0 * a64_slices * 16384.0 * 64);
              launch_mfma_apply64(s->stream, da, (int)a64.size(), wgs); }
            norm_and_replace<T>(s, a64_verts, a64_outs, a64_ne, np64, tb64, nt64, ao.normalize_tensors != 0);
        }
        {   // chi = 64 sites: K = (s, b) = 128 -> N = (s', b') <= 128 on the register-direct MFMA kernel
            std::vector<FiberItem> rg; std::vector<int> rverts, rtb, rnt; std::vector<Buf> routs; std::vector<size_t> rne; double rt = 0, rby = 0, rfl = 0;
            if (std::is_same<T, float>::value && use_mfma() && use_rowgemm())
                for (size_t q = 0; q < own_idx.size(); ++q) {
                    if (via64[q]) continue;
                    size_t i = own_idx[q]; int gi = (int)i / 2; int chin = info[8 * gi + 2]; const SiteJob& j = sj[i];
                    FiberItem it{};
                    it.D = j.sd.d; it.PA = (int)(j.sd.pre(j.bleg) / j.sd.d); it.K = j.sd.chi[j.bleg]; it.PB = (int)j.sd.post(j.bleg); it.Do = j.sd.d; it.No = chin;
                    if (!rowgemm_covers(it) || it.D != 2) continue;
                    const size_t nout = j.sd.n / it.K * chin;
                    Buf out = dalloc(s, nout * esz);
                    it.in = pch[q].result; it.out = out->p; it.X = (i & 1) ? ws[gi].X2->p : ws[gi].X1->p;
                    rowgemm_tiles(it); it.want_norm = ao.normalize_tensors ? 1 : 0;
                    rg.push_back(it); rverts.push_back(j.v); routs.push_back(out); rne.push_back(nout); rt += (double)it.nta * it.ntb;
                    rby += (double)(j.sd.n + nout) * esz; rfl += 8.0 * j.sd.n * j.sd.d * chin; via64[q] = 1;
                }
            if (!rg.empty()) {
                int tpw = (int)std::max(4.0, std::min(32.0, rt / 2048.0)); tpw &= ~3; int wgs = 0;
                for (auto& it : rg) { const int nwg = (it.nta * it.ntb + tpw - 1) / tpw; it.tpw = tpw; it.tile_begin = wgs; rtb.push_back(wgs); rnt.push_back(nwg); wgs += nwg; }
                Buf npr = dalloc(s, (size_t)wgs * sizeof(double));
                const FiberItem* d = upload(s, rg);
                { ProfScope ps(s, TNQS_PROF_GATE_APPLY, rby, rfl); launch_mfma_rowgemm(s->stream, d, (int)rg.size(), wgs, 2, reinterpret_cast<double*>(npr->p)); }
                norm_and_replace<T>(s, rverts, routs, rne, npr, rtb, rnt, ao.normalize_tensors != 0);
            }
        }
        for (size_t q = 0; q < own_idx.size(); ++q) {
            if (via64[q]) continue;
            size_t i = own_idx[q];
            int gi = (int)i / 2; int chin = info[8 * gi + 2];
            const SiteJob& j = sj[i];
            size_t pre = j.sd.pre(j.bleg), post = j.sd.post(j.bleg);
            int chi = j.sd.chi[j.bleg];
            size_t nout = j.sd.n / chi * chin;
            FiberItem it{}; Buf out = dalloc(s, nout * esz);
            it.in = pch[q].result; it.out = out->p; it.X = (i & 1) ? ws[gi].X2->p : ws[gi].X1->p;
            it.D = j.sd.d; it.PA = (int)(pre / j.sd.d); it.K = chi; it.PB = (int)post; it.Do = j.sd.d; it.No = chin;
            tile_params(it.PA, it.PB, TR, it.TA, it.TB, it.nta, it.ntb);
            it.tpw = mf ? (TR == 32 ? 16 : 4) : 1;
            const int nwg = (it.nta * it.ntb + it.tpw - 1) / it.tpw;
            it.tile_begin = tiles; it.want_norm = ao.normalize_tensors ? 1 : 0;
            verts.push_back(j.v); outs.push_back(out); ne.push_back(nout); tb.push_back(tiles); nt.push_back(nwg);
            tiles += nwg; items.push_back(it);
            bytes += (double)(j.sd.n + nout) * esz; flops += 8.0 * j.sd.n * j.sd.d * chin;
        }
        Buf np = dalloc(s, std::max(1, tiles) * sizeof(double));
        const FiberItem* d = upload(s, items);
        if (!items.empty())
        { ProfScope ps(s, TNQS_PROF_GATE_APPLY, bytes, flops);
          if (mf) launch_mfma_fiber_gemm(s->stream, d, (int)items.size(), tiles, (int)KKmax, (int)NNmax, reinterpret_cast<double*>(np->p));
          else launch_fiber_gemm<T>(s->stream, d, (int)items.size(), tiles, TR, (int)KKmax, reinterpret_cast<double*>(np->p)); }
        norm_and_replace<T>(s, verts, outs, ne, np, tb, nt, ao.normalize_tensors != 0);
    }
    // ---- 6. both bond messages := diag(S)  (apply_gates.jl:126-135), new bond dimension ---------------------------
    {
        std::vector<DiagItem> di;
        for (int gi = 0; gi < ng; ++gi) {
            int e = g.edge(gates[gi].v1, gates[gi].v2); int chin = info[8 * gi + 2];
            s->chi[e] = chin;
            for (int dir = 0; dir < 2; ++dir) {
                Buf m = dalloc(s, (size_t)chin * chin * esz);
                di.push_back(DiagItem{m->p, Sptr[gi], chin});
                s->msg[2 * e + dir] = m;
            }
            if (errs) errs[gates[gi].index] = terr[gi];
        }
        const DiagItem* d = upload(s, di);
        { ProfScope ps(s, TNQS_PROF_SMALL, 0, 0); launch_diag<T>(s->stream, d, (int)di.size()); }
    }
    s->stats.n_two_site += ng;
    sync(s);   // workspace of this batch is released to the pool after the stream drained
}

template <class T> static void flush_batch(State* s, std::vector<Gate1>& b1, std::vector<Gate2>& b2, const tnqs_apply_opts& ao, double* errs) {
    if (b1.empty() && b2.empty()) return;
    apply_one_site_batch<T>(s, b1, ao.normalize_tensors != 0);
    apply_two_site_batch<T>(s, b2, ao, errs);
    s->stats.n_batches += 1;
    b1.clear(); b2.clear();
    sync(s);
}

template <class T> static void apply_gates_t(State* s, int ngates, const int32_t* nverts, const int32_t* verts, const double* mats,
                                             const tnqs_apply_opts* opts, const tnqs_bp_opts* bp, double* errs) {
    const Graph& g = *s->g;
    HIPCHK(hipSetDevice(s->device));
    tnqs_apply_opts ao; ao.maxdim = 0; ao.cutoff = -1; ao.normalize_tensors = 1; ao.sqrt_cutoff = -1; ao.update_cache = 1;
    if (opts) ao = *opts;
    // validation first (apply_gates.jl:109-120): nothing is mutated when an argument is bad
    std::vector<int> voff(ngates + 1, 0); std::vector<size_t> moff(ngates + 1, 0);
    for (int i = 0; i < ngates; ++i) {
        int nv = nverts[i];
        if (nv < 1 || nv > 2) throw Err(TNQS_ERR_INVALID, "apply_gate!: only one- and two-site gates are supported; received a gate acting on " + std::to_string(nv) + " vertices.");
        voff[i + 1] = voff[i] + nv;
        size_t dd = 1;
        for (int k = 0; k < nv; ++k) { int v = verts[voff[i] + k]; if (v < 0 || v >= g.nv) throw Err(TNQS_ERR_INVALID, "apply_gates: vertex out of range"); dd *= s->d[v]; }
        moff[i + 1] = moff[i] + 2 * dd * dd;
        if (nv == 2) {
            int a = verts[voff[i]], b = verts[voff[i] + 1];
            if (a == b || g.edge(a, b) < 0)
                throw Err(TNQS_ERR_INVALID, "apply_gate!: cannot apply a two-site gate on the non-adjacent vertices " + std::to_string(a) + " and " + std::to_string(b) +
                                                ". Simple update requires the two sites to share an edge of the tensor-network graph.");
        }
    }
    if (errs) std::fill(errs, errs + ngates, 0.0);
    if (s->real_io) {       // adapt_gate (apply_gates.jl:41-44): a real gate takes the state's real type, a complex gate stays complex and promotes
        bool cplx = false;
        for (size_t k = 1; k < moff[ngates] && !cplx; k += 2) cplx = mats[k] != 0.0;
        if (cplx) s->real_io = false;
    }
    std::set<int> affected, batch_verts;
    std::vector<Gate1> b1; std::vector<Gate2> b2;
    for (int i = 0; i < ngates; ++i) {
        const int nv = nverts[i]; const int32_t* vs = verts + voff[i];
        bool need = false;
        if (nv >= 2) for (int k = 0; k < nv; ++k) need = need || affected.count(vs[k]);            // apply_gates.jl:68
        if (ao.update_cache && need) {
            flush_batch<T>(s, b1, b2, ao, errs); batch_verts.clear();
            bp_update_t<T>(s, bp, nullptr, nullptr);                                               // :76
            affected.clear();                                                                      // :78
        }
        bool overlap = false;
        for (int k = 0; k < nv; ++k) overlap = overlap || batch_verts.count(vs[k]);
        if (overlap) { flush_batch<T>(s, b1, b2, ao, errs); batch_verts.clear(); }
        if (nv == 1) b1.push_back(Gate1{vs[0], mats + moff[i]}); else b2.push_back(Gate2{vs[0], vs[1], mats + moff[i], i});
        for (int k = 0; k < nv; ++k) { batch_verts.insert(vs[k]); affected.insert(vs[k]); }         // :88-90
    }
    flush_batch<T>(s, b1, b2, ao, errs);
    if (ao.update_cache) bp_update_t<T>(s, bp, nullptr, nullptr);                                   // :93-95
}

void apply_gates(State* s, int ngates, const int32_t* nverts, const int32_t* verts, const double* mats,
                 const tnqs_apply_opts* opts, const tnqs_bp_opts* bp, double* errs) {
    s->stats = tnqs_apply_stats{};
    if (s->dtype == TNQS_C64) apply_gates_t<float>(s, ngates, nverts, verts, mats, opts, bp, errs);
    else apply_gates_t<double>(s, ngates, nverts, verts, mats, opts, bp, errs);
}

// ---------------------------------------------------------------------------------------------------------------
// truncate (src/truncate.jl:12-38)
// ---------------------------------------------------------------------------------------------------------------
template <class T> static void truncate_t(State* s, int maxdim, double cutoff, int normalize, int ngroups, const int32_t* offs,
                                          const int32_t* eu, const int32_t* ev, const tnqs_bp_opts* bp) {
    const Graph& g = *s->g;
    HIPCHK(hipSetDevice(s->device));
    if (maxdim <= 0) throw Err(TNQS_ERR_INVALID, "truncate: maxdim must be a positive integer");
    tnqs_apply_opts ao; ao.maxdim = maxdim; ao.cutoff = cutoff; ao.normalize_tensors = normalize; ao.sqrt_cutoff = -1; ao.update_cache = 1;
    std::vector<std::vector<double>> idmats;
    auto ident = [&](int dd) { std::vector<double> m(2 * (size_t)dd * dd, 0.0); for (int i = 0; i < dd; ++i) m[2 * (size_t)(i + (size_t)dd * i)] = 1.0; return m; };
    auto run_group = [&](const std::vector<std::pair<int, int>>& edges) {
        std::vector<Gate2> b2; std::set<int> seen; idmats.clear(); idmats.reserve(edges.size());
        for (auto& pr : edges) {
            int e = g.edge(pr.first, pr.second);
            if (e < 0) throw Err(TNQS_ERR_INVALID, "truncate: colour group contains a non-edge");
            if (s->chi[e] == 1) continue;                                   // truncatable_edge (:5-10)
            if (seen.count(pr.first) || seen.count(pr.second)) throw Err(TNQS_ERR_INVALID, "truncate: edges of one colour group must be vertex-disjoint");
            seen.insert(pr.first); seen.insert(pr.second);
            idmats.push_back(ident(s->d[pr.first] * s->d[pr.second]));
            b2.push_back(Gate2{pr.first, pr.second, idmats.back().data(), 0});
        }
        apply_two_site_batch<T>(s, b2, ao, nullptr);
        if (!b2.empty()) s->stats.n_batches += 1;
        bp_update_t<T>(s, bp, nullptr, nullptr);                               // :28 / :34
    };
    if (ngroups > 0) {
        for (int c = 0; c < ngroups; ++c) {
            std::vector<std::pair<int, int>> edges;
            for (int i = offs[c]; i < offs[c + 1]; ++i) edges.push_back({eu[i], ev[i]});
            run_group(edges);
        }
    } else {
        for (int e = 0; e < g.ne; ++e) run_group({{g.esrc[e], g.edst[e]}});
    }
}
void truncate_bp(State* s, int maxdim, double cutoff, int normalize, int ngroups, const int32_t* offs,
                 const int32_t* eu, const int32_t* ev, const tnqs_bp_opts* bp) {
    s->stats = tnqs_apply_stats{};
    if (s->dtype == TNQS_C64) truncate_t<float>(s, maxdim, cutoff, normalize, ngroups, offs, eu, ev, bp);
    else truncate_t<double>(s, maxdim, cutoff, normalize, ngroups, offs, eu, ev, bp);
}

// ---------------------------------------------------------------------------------------------------------------
// parity probes (src/expect.jl:59-82)
// ---------------------------------------------------------------------------------------------------------------
template <class T> static void rdm_batch(State* s, const std::vector<int>& vs, double* out /* per vertex d*d complex128, packed */) {
    const Graph& g = *s->g;
    HIPCHK(hipSetDevice(s->device));
    std::vector<Chain> chains(vs.size());
    for (size_t i = 0; i < vs.size(); ++i) {
        int v = vs[i];
        if (!s->site[v]) throw Err(TNQS_ERR_INVALID, "rdm: vertex not owned by this rank");
        Chain& c = chains[i]; c.v = v; c.src = s->site[v]->p; c.sd = site_dims(s, v);
        for (int j = 0; j < c.sd.z; ++j) { int de = g.dedge(g.nbr[v][j], v); if (s->msg[de]) c.steps.push_back({j, s->msg[de]->p}); }
    }
    run_chains<T>(s, chains, TNQS_PROF_SMALL);
    std::vector<GramJob> jobs;
    for (size_t i = 0; i < vs.size(); ++i) { GramJob j{}; j.X = chains[i].result; j.Y = chains[i].src; j.sd = chains[i].sd; j.leg = -1; j.keep_site = true; jobs.push_back(j); }
    run_grams<T, double>(s, jobs, TNQS_PROF_SMALL);
    std::vector<ReduceItem> ri; int elems = 0; std::vector<int> off;
    for (size_t i = 0; i < vs.size(); ++i) { int n2 = jobs[i].KK * jobs[i].KK; off.push_back(elems); elems += n2; }
    Buf d_out = dalloc(s, (size_t)elems * 16);
    for (size_t i = 0; i < vs.size(); ++i) {
        int n2 = jobs[i].KK * jobs[i].KK;
        ri.push_back(ReduceItem{jobs[i].partial->p, reinterpret_cast<char*>(d_out->p) + (size_t)off[i] * 16, n2, jobs[i].nchunks, 0, off[i]});
    }
    const ReduceItem* dr = upload(s, ri);
    launch_reduce<double, double>(s->stream, dr, (int)ri.size(), elems);
    HIPCHK(hipMemcpyAsync(out, d_out->p, (size_t)elems * 16, hipMemcpyDeviceToHost, s->stream));
    std::vector<double> fac(vs.size(), 1.0);
    for (size_t i = 0; i < vs.size(); ++i) if (s->sscale[vs[i]]) HIPCHK(hipMemcpyAsync(&fac[i], s->sscale[vs[i]]->p, 8, hipMemcpyDeviceToHost, s->stream));
    sync(s);
    for (size_t i = 0; i < vs.size(); ++i) if (fac[i] != 1.0) { int n2 = jobs[i].KK * jobs[i].KK; for (int k = 0; k < 2 * n2; ++k) out[2 * (size_t)off[i] + k] *= fac[i] * fac[i]; }
}
void rdm_1site(State* s, int v, double* out) {
    if (v < 0 || v >= s->g->nv) throw Err(TNQS_ERR_INVALID, "rdm_1site: bad vertex");
    std::vector<int> vs{v};
    if (s->dtype == TNQS_C64) rdm_batch<float>(s, vs, out); else rdm_batch<double>(s, vs, out);
}
// ---------------------------------------------------------------------------------------------------------------
// BP scalars and normalisation: vertex_scalar (abstract...:22-28), edge_scalar (beliefpropagationcache.jl:47-49),
// rescale! = rescale_messages! (:127-140) then rescale_vertices! (:82-101)
// ---------------------------------------------------------------------------------------------------------------
void vertex_scalars(State* s, double* out /* nv complex128; NaN for vertices of other ranks */) {
    const Graph& g = *s->g;
    std::vector<int> vs; std::vector<size_t> off; size_t tot = 0;
    for (int v = 0; v < g.nv; ++v) if (s->owns(v)) { vs.push_back(v); off.push_back(tot); tot += 2 * (size_t)s->d[v] * s->d[v]; }
    std::vector<double> rho(tot);
    if (!vs.empty()) { if (s->dtype == TNQS_C64) rdm_batch<float>(s, vs, rho.data()); else rdm_batch<double>(s, vs, rho.data()); }
    for (int v = 0; v < g.nv; ++v) { out[2 * v] = std::nan(""); out[2 * v + 1] = std::nan(""); }
    for (size_t q = 0; q < vs.size(); ++q) {
        int v = vs[q], d = s->d[v]; double tre = 0, tim = 0;
        for (int si = 0; si < d; ++si) { tre += rho[off[q] + 2 * (si + d * si)]; tim += rho[off[q] + 2 * (si + d * si) + 1]; }
        out[2 * v] = tre; out[2 * v + 1] = tim;
    }
}
template <class T> static void edge_scalars_t(State* s, double* out) {
    const Graph& g = *s->g;
    HIPCHK(hipSetDevice(s->device));
    if (g.ne == 0) return;
    Buf d_out = dalloc(s, (size_t)g.ne * 16);
    std::vector<EdgeScalarItem> items;
    for (int e = 0; e < g.ne; ++e)
        items.push_back(EdgeScalarItem{s->msg[2 * e] ? s->msg[2 * e]->p : nullptr, s->msg[2 * e + 1] ? s->msg[2 * e + 1]->p : nullptr, s->chi[e],
                                       reinterpret_cast<double*>(d_out->p) + 2 * e});
    const EdgeScalarItem* d = upload(s, items);
    launch_edge_scalar<T>(s->stream, d, (int)items.size());
    HIPCHK(hipMemcpyAsync(out, d_out->p, (size_t)g.ne * 16, hipMemcpyDeviceToHost, s->stream));
    sync(s);
}
void edge_scalars(State* s, double* out) { if (s->dtype == TNQS_C64) edge_scalars_t<float>(s, out); else edge_scalars_t<double>(s, out); }

template <class T> static void rescale_t(State* s) {
    const Graph& g = *s->g;
    const size_t esz = s->esz();
    HIPCHK(hipSetDevice(s->device));
    // rescale_messages!: every edge, both directions (replicated on every rank when sharded)
    if (g.ne > 0) {
        std::vector<MsgRescaleItem> items; std::vector<Buf> na(g.ne), nb(g.ne);
        for (int e = 0; e < g.ne; ++e) {
            size_t bytes = (size_t)s->chi[e] * s->chi[e] * esz;
            na[e] = dalloc(s, bytes); nb[e] = dalloc(s, bytes);
            items.push_back(MsgRescaleItem{s->msg[2 * e] ? s->msg[2 * e]->p : nullptr, s->msg[2 * e + 1] ? s->msg[2 * e + 1]->p : nullptr, na[e]->p, nb[e]->p, s->chi[e]});
        }
        const MsgRescaleItem* d = upload(s, items);
        launch_msg_rescale<T>(s->stream, d, (int)items.size());
        for (int e = 0; e < g.ne; ++e) { s->keepalive.push_back(s->msg[2 * e]); s->keepalive.push_back(s->msg[2 * e + 1]); s->msg[2 * e] = na[e]; s->msg[2 * e + 1] = nb[e]; }
    }
    // rescale_vertices!: psi_v *= sign(vn) / sqrt(vn) with vn = vertex_scalar under the rescaled messages
    std::vector<double> vn(2 * (size_t)g.nv);
    vertex_scalars(s, vn.data());
    materialize_scale_all(s);
    std::vector<CScaleItem> cs; std::vector<Buf> outs; std::vector<int> vs;
    for (int v = 0; v < g.nv; ++v) {
        if (!s->owns(v) || !s->site[v]) continue;
        double re = vn[2 * v], im = vn[2 * v + 1];
        double sgn = 1.0;
        if (im == 0.0) sgn = (re > 0) - (re < 0);             // isreal(vn) ? sign(vn) : one(vn)
        const double mod = std::sqrt(re * re + im * im), arg = std::atan2(im, re);
        if (!(mod > 0)) throw Err(TNQS_ERR_NUMERIC, "rescale: a vertex scalar is zero");
        const double r = sgn / std::sqrt(mod), ph = -0.5 * arg;
        Buf out = dalloc(s, s->site[v]->bytes);
        cs.push_back(CScaleItem{s->site[v]->p, out->p, s->site[v]->bytes / esz, r * std::cos(ph), r * std::sin(ph)});
        outs.push_back(out); vs.push_back(v);
    }
    if (!cs.empty()) {
        const CScaleItem* d = upload(s, cs);
        launch_cscale<T>(s->stream, d, (int)cs.size());
        for (size_t i = 0; i < vs.size(); ++i) { s->keepalive.push_back(s->site[vs[i]]); s->site[vs[i]] = outs[i]; }
    }
    sync(s);
}
void rescale(State* s) { if (s->dtype == TNQS_C64) rescale_t<float>(s); else rescale_t<double>(s); }

// ---------------------------------------------------------------------------------------------------------------
// multi-site expectation value on a tree-shaped region (src/expect.jl:59-82): the norm network of the region's
// vertices with the cache's messages on the boundary edges and operators inserted, numerator (ops) over denominator (identities).
// The region is contracted leaves-to-root with the message kernels: m_{u->parent} = sum (O_u psi_u) conj(psi_u) prod(incoming).
// ---------------------------------------------------------------------------------------------------------------
template <class T> static void region_contract(State* s, int nr, const int32_t* rv, const int32_t* parent, const double* ops /* may be null */,
                                               double* out_re_im) {
    const Graph& g = *s->g;
    const size_t esz = s->esz();
    std::vector<int> pos(g.nv, -1);
    for (int i = 0; i < nr; ++i) pos[rv[i]] = i;
    // post-order: children before parents (depth descending)
    std::vector<int> depth(nr, 0), order(nr);
    for (int i = 0; i < nr; ++i) { int d = 0, p = i; while (parent[p] >= 0) { p = parent[p]; if (++d > nr) throw Err(TNQS_ERR_INVALID, "expect_region: parent array has a cycle"); } depth[i] = d; }
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return depth[a] > depth[b]; });
    std::vector<Buf> up(nr);                       // message from region vertex i to its parent
    const bool sharded = s->nranks > 1;
    // sharded handles: the owner of a region vertex contracts it; its message to the parent (chi x chi) -- or, at the root, the d x d
    // result -- reaches every rank through one exchange per region vertex (all ranks walk the region in the same order)
    for (int oi = 0; oi < nr; ++oi) {
        const int i = order[oi], u = rv[i], par = parent[i] >= 0 ? rv[parent[i]] : -1;
        const bool mine = s->owns(u);
        const int dU = s->d[u];
        const int n2 = par >= 0 ? s->chi[g.edge(u, par)] * s->chi[g.edge(u, par)] : dU * dU;
        const size_t out_esz = par >= 0 ? esz : 16;
        const size_t stride = round256((size_t)n2 * out_esz);
        if (sharded) check_exchange(s, stride);
        Buf result = dalloc(s, (size_t)n2 * out_esz);
        void* reduce_dst = sharded ? (void*)(reinterpret_cast<char*>(s->exch) + (size_t)s->rank * stride) : result->p;
        if (mine) {
            SD sd = site_dims(s, u);
            const void* ket = s->site[u]->p; Buf opbuf;
            if (ops) {                                  // ket := O_u psi_u   (out[s'] = sum_s O[s', s] psi[s])
                size_t off = 0; for (int q = 0; q < i; ++q) off += 2 * (size_t)s->d[rv[q]] * s->d[rv[q]];
                const double* m = ops + off;
                const int d = sd.d; bool ident = true;
                for (int aa = 0; aa < d && ident; ++aa) for (int bb = 0; bb < d; ++bb) if (m[2 * (aa + d * bb)] != (aa == bb ? 1.0 : 0.0) || m[2 * (aa + d * bb) + 1] != 0.0) { ident = false; break; }
                if (!ident) {
                    std::vector<T> hx;
                    for (int nn = 0; nn < d; ++nn) for (int kk = 0; kk < d; ++kk) { hx.push_back((T)m[2 * (nn + d * kk)]); hx.push_back((T)m[2 * (nn + d * kk) + 1]); }
                    std::vector<char> raw(reinterpret_cast<char*>(hx.data()), reinterpret_cast<char*>(hx.data()) + hx.size() * sizeof(T));
                    const char* dx = upload(s, raw);
                    opbuf = dalloc(s, sd.n * esz);
                    FiberItem it{}; it.in = ket; it.out = opbuf->p; it.X = dx;
                    it.D = d; it.PA = (int)(sd.n / d); it.K = 1; it.PB = 1; it.Do = d; it.No = 1;
                    const int TR = pick_TR(d, esz, 1);
                    tile_params(it.PA, it.PB, TR, it.TA, it.TB, it.nta, it.ntb);
                    it.tpw = 1; it.tile_begin = 0; it.want_norm = 0;
                    std::vector<FiberItem> items{it};
                    const FiberItem* dI = upload(s, items);
                    launch_fiber_gemm<T>(s->stream, dI, 1, it.nta * it.ntb, TR, d, nullptr);
                    ket = opbuf->p;
                }
            }
            std::vector<Chain> chains(1); Chain& c = chains[0]; c.v = u; c.src = ket; c.sd = sd;
            for (int j = 0; j < sd.z; ++j) {
                int k = g.nbr[u][j]; if (k == par) continue;
                const void* mp = nullptr;
                if (pos[k] >= 0) {
                    if (parent[pos[k]] < 0 || rv[parent[pos[k]]] != u) throw Err(TNQS_ERR_INVALID, "expect_region: the region's induced subgraph is not the given tree");
                    mp = up[pos[k]]->p;
                } else { int de = g.dedge(k, u); if (s->msg[de]) mp = s->msg[de]->p; }
                if (mp) c.steps.push_back({j, mp});
            }
            run_chains<T>(s, chains, TNQS_PROF_SMALL);
            std::vector<GramJob> jobs(1);
            GramJob& j = jobs[0]; j.X = chains[0].result; j.Y = s->site[u]->p; j.sd = sd;
            if (par >= 0) { j.leg = g.leg(u, par); j.keep_site = false; } else { j.leg = -1; j.keep_site = true; }
            if (par >= 0) run_grams<T, T>(s, jobs, TNQS_PROF_SMALL); else run_grams<T, double>(s, jobs, TNQS_PROF_SMALL);
            if (j.KK * j.KK != n2) throw Err(TNQS_ERR_HIP, "internal: expect_region result size");
            std::vector<ReduceItem> ri{ReduceItem{j.partial->p, reduce_dst, n2, j.nchunks, 0, 0}};
            const ReduceItem* dr = upload(s, ri);
            if (par >= 0) launch_reduce<T, T>(s->stream, dr, 1, n2); else launch_reduce<double, double>(s->stream, dr, 1, n2);
        } else {
            // the tree-shape check of the owner is repeated here so that every rank fails (or not) together
            for (size_t j = 0; j < g.nbr[u].size(); ++j) { int k = g.nbr[u][j]; if (k == par) continue;
                if (pos[k] >= 0 && (parent[pos[k]] < 0 || rv[parent[pos[k]]] != u)) throw Err(TNQS_ERR_INVALID, "expect_region: the region's induced subgraph is not the given tree"); }
        }
        if (sharded) {
            exchange(s, stride);
            HIPCHK(hipMemcpyAsync(result->p, reinterpret_cast<char*>(s->exch) + (size_t)s->owner[u] * stride, (size_t)n2 * out_esz, hipMemcpyDeviceToDevice, s->stream));
        }
        if (par >= 0) up[i] = result;
        else {
            std::vector<double> rho(2 * (size_t)n2);
            HIPCHK(hipMemcpyAsync(rho.data(), result->p, (size_t)n2 * 16, hipMemcpyDeviceToHost, s->stream));
            sync(s);
            double tre = 0, tim = 0; const int d = dU;
            for (int si = 0; si < d; ++si) { tre += rho[2 * (si + d * si)]; tim += rho[2 * (si + d * si) + 1]; }
            // a pending normalisation factor of a site tensor cancels between numerator and denominator
            out_re_im[0] = tre; out_re_im[1] = tim;
        }
    }
}
void expect_region(State* s, int nr, const int32_t* rv, const int32_t* parent, const double* ops, double* out4) {
    const Graph& g = *s->g;
    if (nr < 1 || !rv || !parent || !ops || !out4) throw Err(TNQS_ERR_INVALID, "expect_region: bad arguments");
    int roots = 0;
    for (int i = 0; i < nr; ++i) {
        if (rv[i] < 0 || rv[i] >= g.nv) throw Err(TNQS_ERR_INVALID, "expect_region: bad vertex");
        if (parent[i] < 0) ++roots; else if (parent[i] >= nr || g.edge(rv[i], rv[parent[i]]) < 0) throw Err(TNQS_ERR_INVALID, "expect_region: parent is not a neighbour");
    }
    if (roots != 1) throw Err(TNQS_ERR_INVALID, "expect_region: exactly one root expected");
    HIPCHK(hipSetDevice(s->device));
    if (s->dtype == TNQS_C64) { region_contract<float>(s, nr, rv, parent, ops, out4); region_contract<float>(s, nr, rv, parent, nullptr, out4 + 2); }
    else { region_contract<double>(s, nr, rv, parent, ops, out4); region_contract<double>(s, nr, rv, parent, nullptr, out4 + 2); }
}

// ---------------------------------------------------------------------------------------------------------------
// symmetric gauge (src/symmetric_gauge.jl:1-62).  The reference loops over the edges; every edge only touches
// its own leg of the two site tensors and its own two messages, so all edges are factorised in one batch and each site
// receives the mode products of all its legs in one chain (different legs commute).
// ---------------------------------------------------------------------------------------------------------------
template <class T> static void symmetric_gauge_t(State* s, double regularization) {
    const Graph& g = *s->g;
    const size_t esz = s->esz();
    // sharded handles: the per-edge algebra (2|E| eigen problems, |E| SVDs of chi x chi matrices) is replicated -- every rank runs the same
    // kernels on the same replicated messages -- and each rank gauges the site tensors it owns; nothing is exchanged
    HIPCHK(hipSetDevice(s->device));
    if (g.ne == 0) return;
    const double reg = regularization >= 0 ? regularization : 10.0 * (s->dtype == TNQS_C64 ? 1.1920928955078125e-07 : 2.220446049250313e-16);
    // Hermitian eigen factorisations of all 2|E| messages in f64 (safe_eigen, utils.jl:94-108)
    std::vector<Buf> H(2 * (size_t)g.ne), V(2 * (size_t)g.ne);
    std::vector<EnvItem> ei; std::vector<JacobiItem> ji;
    for (int de = 0; de < 2 * g.ne; ++de) {
        int n = s->chi[de / 2];
        if (n > 256) throw Err(TNQS_ERR_UNSUPPORTED, "symmetric_gauge: bond dimension > 256");
        H[de] = dalloc(s, (size_t)n * n * 16); V[de] = dalloc(s, (size_t)n * n * 16);
        ei.push_back(EnvItem{s->msg[de] ? s->msg[de]->p : nullptr, H[de]->p, V[de]->p, n});
        ji.push_back(JacobiItem{H[de]->p, V[de]->p, n, n, nullptr});
    }
    { const EnvItem* d = upload(s, ei); launch_env_prepare<T>(s->stream, d, (int)ei.size()); }
    { const JacobiItem* d = upload(s, ji); size_t lds = 0; for (auto& j : ji) lds = std::max(lds, jacobi_lds_bytes(j.n, j.n, true, 16));
      launch_jacobi<double>(s->stream, d, (int)ji.size(), 60, jacobi_lds(lds), mmax_of(ji)); }
    // per edge: roots, Ce, its SVD, the two gauge matrices
    Buf d_flag = dalloc(s, sizeof(int));
    HIPCHK(hipMemsetAsync(d_flag->p, 0, sizeof(int), s->stream));
    struct EdgeWS { Buf rx, ry, irx, iry, Ce, Ce0, Vs, Xs, Xd, S; };
    std::vector<EdgeWS> ws(g.ne); std::vector<SymGaugeItem> items; std::vector<JacobiItem> sj; std::vector<RecoverItem> rv; int nmax = 1;
    for (int e = 0; e < g.ne; ++e) {
        int n = s->chi[e]; size_t nn = (size_t)n * n; EdgeWS& w = ws[e];
        w.rx = dalloc(s, nn * 16); w.ry = dalloc(s, nn * 16); w.irx = dalloc(s, nn * 16); w.iry = dalloc(s, nn * 16);
        w.Ce = dalloc(s, nn * esz); w.Ce0 = dalloc(s, nn * esz); w.Vs = dalloc(s, nn * esz); w.Xs = dalloc(s, nn * esz); w.Xd = dalloc(s, nn * esz);
        w.S = dalloc(s, (size_t)n * 8);
        items.push_back(SymGaugeItem{H[2 * e]->p, V[2 * e]->p, H[2 * e + 1]->p, V[2 * e + 1]->p, w.rx->p, w.ry->p, w.irx->p, w.iry->p,
                                     w.Ce->p, w.Ce0->p, w.Vs->p, w.Xs->p, w.Xd->p, reinterpret_cast<double*>(w.S->p), n, reg, reinterpret_cast<int*>(d_flag->p)});
        sj.push_back(JacobiItem{w.Ce->p, nullptr, n, n, nullptr});
        rv.push_back(RecoverItem{w.Ce0->p, w.Ce->p, w.Vs->p, n, n, n}); nmax = std::max(nmax, n);
    }
    const SymGaugeItem* d_items = upload(s, items);
    launch_symg_build<T>(s->stream, d_items, (int)items.size());
    { const JacobiItem* d = upload(s, sj); size_t lds = 0; for (auto& j : sj) lds = std::max(lds, jacobi_lds_bytes(j.m, j.n, false, esz));
      launch_jacobi<T>(s->stream, d, (int)sj.size(), 60, jacobi_lds(lds), mmax_of(sj)); }
    { const RecoverItem* d = upload(s, rv);
      if (std::is_same<T, float>::value && use_mfma()) launch_recover_v_mfma(s->stream, d, (int)rv.size(), nmax); else launch_recover_v<T>(s->stream, d, (int)rv.size(), nmax); }
    launch_symg_finish<T>(s->stream, d_items, (int)items.size());
    int flag = 0;
    HIPCHK(hipMemcpyAsync(&flag, d_flag->p, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    if (flag) throw Err(TNQS_ERR_NUMERIC, "symmetric_gauge: a regularised message eigenvalue is negative (DomainError in the reference, symmetric_gauge.jl:18)");
    // site tensors: psi_v <- psi_v x_leg X for every leg (source end of edge e: Xs, destination end: Xd)
    std::vector<int> verts; std::vector<Chain> chains;
    for (int v = 0; v < g.nv; ++v) {
        if (!s->site[v] || g.nbr[v].empty()) continue;
        Chain c; c.v = v; c.src = s->site[v]->p; c.sd = site_dims(s, v);
        for (int j = 0; j < c.sd.z; ++j) { int e = g.nbr_e[v][j]; c.steps.push_back({j, (g.esrc[e] == v) ? ws[e].Xs->p : ws[e].Xd->p}); }
        chains.push_back(std::move(c)); verts.push_back(v);
    }
    run_chains<T>(s, chains, TNQS_PROF_SMALL);
    for (size_t i = 0; i < chains.size(); ++i) {
        Buf nb;
        for (int k = 0; k < 2; ++k) if (chains[i].tmp[k] && chains[i].tmp[k]->p == chains[i].result) nb = chains[i].tmp[k];
        if (!nb) throw Err(TNQS_ERR_HIP, "internal: symmetric_gauge chain result");
        s->keepalive.push_back(s->site[verts[i]]); s->site[verts[i]] = nb;
    }
    // both messages of an edge := diag(S)   (:54-55)
    std::vector<DiagItem> di;
    for (int e = 0; e < g.ne; ++e) {
        int n = s->chi[e];
        for (int dir = 0; dir < 2; ++dir) { Buf m = dalloc(s, (size_t)n * n * esz); di.push_back(DiagItem{m->p, reinterpret_cast<const double*>(ws[e].S->p), n});
                                             s->keepalive.push_back(s->msg[2 * e + dir]); s->msg[2 * e + dir] = m; }
    }
    { const DiagItem* d = upload(s, di); launch_diag<T>(s->stream, d, (int)di.size()); }
    sync(s);
}
void symmetric_gauge(State* s, double regularization) { if (s->dtype == TNQS_C64) symmetric_gauge_t<float>(s, regularization); else symmetric_gauge_t<double>(s, regularization); }

void expect_all(State* s, const double* ops, double* out) {
    const Graph& g = *s->g;
    std::vector<int> vs; std::vector<size_t> off; size_t tot = 0;
    for (int v = 0; v < g.nv; ++v) if (s->owns(v)) { vs.push_back(v); off.push_back(tot); tot += 2 * (size_t)s->d[v] * s->d[v]; }
    std::vector<double> rho(tot);
    if (!vs.empty()) { if (s->dtype == TNQS_C64) rdm_batch<float>(s, vs, rho.data()); else rdm_batch<double>(s, vs, rho.data()); }
    size_t opoff = 0; size_t q = 0;
    for (int v = 0; v < g.nv; ++v) {
        int d = s->d[v];
        if (q < vs.size() && vs[q] == v) {
            const double* r = rho.data() + off[q]; const double* o = ops + opoff;
            double nre = 0, nim = 0, tre = 0, tim = 0;
            for (int sp = 0; sp < d; ++sp) for (int si = 0; si < d; ++si) {       // sum op[s',s] rho[s,s']
                double ore = o[2 * (sp + d * si)], oim = o[2 * (sp + d * si) + 1];
                double rre = r[2 * (si + d * sp)], rim = r[2 * (si + d * sp) + 1];
                nre += ore * rre - oim * rim; nim += ore * rim + oim * rre;
            }
            for (int si = 0; si < d; ++si) { tre += r[2 * (si + d * si)]; tim += r[2 * (si + d * si) + 1]; }
            double den = tre * tre + tim * tim;
            out[2 * v] = (nre * tre + nim * tim) / den; out[2 * v + 1] = (nim * tre - nre * tim) / den;
            ++q;
        } else { out[2 * v] = std::nan(""); out[2 * v + 1] = std::nan(""); }
        opoff += 2 * (size_t)d * d;
    }
}

}  // namespace tnqs
