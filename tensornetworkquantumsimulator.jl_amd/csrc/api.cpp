// api.cpp -- extern "C" boundary of libtnqs_hip.so (include/tnqs.h).  No C++ exception crosses the ABI.
#include <cmath>
#include <cstring>
#include <string>
#include "engine.hpp"

using namespace tnqs;

struct tnqs_state_s { State* s; };

static thread_local std::string g_err;

template <class F> static int guard(F&& f) {
    try { f(); return TNQS_OK; }
    catch (const Err& e) { g_err = e.what(); return e.code; }
    catch (const std::bad_alloc&) { g_err = "out of host memory"; return TNQS_ERR_HIP; }
    catch (const std::exception& e) { g_err = e.what(); return TNQS_ERR_INVALID; }
    catch (...) { g_err = "unknown error"; return TNQS_ERR_INVALID; }
}
static State* S(tnqs_handle h) { if (!h || !h->s) throw Err(TNQS_ERR_INVALID, "null handle"); return h->s; }

extern "C" {

int tnqs_version(void) { return 101; }
const char* tnqs_last_error(void) { return g_err.c_str(); }
int tnqs_device_count(int* count) {
    return guard([&] { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) n = 0; if (count) *count = n; });
}

int tnqs_create(int nv, int ne, const int32_t* es, const int32_t* ed, const int32_t* sd, int dtype, int device, tnqs_handle* out) {
    return guard([&] {
        if (!out) throw Err(TNQS_ERR_INVALID, "tnqs_create: out is null");
        if (ne > 0 && (!es || !ed)) throw Err(TNQS_ERR_INVALID, "tnqs_create: edge arrays are null");
        State* s = state_create(nv, ne, es, ed, sd, dtype, device);
        *out = new tnqs_state_s{s};
    });
}
int tnqs_destroy(tnqs_handle h) {
    return guard([&] { if (!h) return; if (h->s) { (void)hipSetDevice(h->s->device); if (h->s->stream) (void)hipStreamSynchronize(h->s->stream); delete h->s; } delete h; });
}
int tnqs_copy(tnqs_handle h, tnqs_handle* out) {
    return guard([&] { if (!out) throw Err(TNQS_ERR_INVALID, "tnqs_copy: out is null"); *out = new tnqs_state_s{state_copy(S(h))}; });
}
int tnqs_scalartype(tnqs_handle h, int* dtype) { return guard([&] { if (!dtype) throw Err(TNQS_ERR_INVALID, "scalartype: null output"); *dtype = S(h)->scalartype(); }); }
int tnqs_set_stream(tnqs_handle h, void* stream) {
    return guard([&] {
        State* s = S(h);
        if (s->own_stream && s->stream) { (void)hipStreamSynchronize(s->stream); (void)hipStreamDestroy(s->stream); }
        s->stream = reinterpret_cast<hipStream_t>(stream); s->own_stream = false;
    });
}

int tnqs_set_site_tensor(tnqs_handle h, int v, const void* host, int ndim, const int64_t* dims, const int32_t* role) {
    return guard([&] { if (!dims || !role) throw Err(TNQS_ERR_INVALID, "set_site_tensor: null argument"); state_set_site(S(h), v, host, ndim, dims, role); });
}
int tnqs_set_site_random(tnqs_handle h, int v, int n_neighbours, const int64_t* bond_dims, uint64_t seed, double scale) {
    return guard([&] { if (!bond_dims && n_neighbours > 0) throw Err(TNQS_ERR_INVALID, "set_site_random: null bond_dims"); state_set_site_random(S(h), v, n_neighbours, bond_dims, seed, scale); });
}
int tnqs_get_site_tensor(tnqs_handle h, int v, void* host, int ndim, const int32_t* role) {
    return guard([&] { if (!host || !role) throw Err(TNQS_ERR_INVALID, "get_site_tensor: null argument"); state_get_site(S(h), v, host, ndim, role); });
}
int tnqs_site_tensor_size(tnqs_handle h, int v, int64_t* n) {
    return guard([&] { State* s = S(h); if (v < 0 || v >= s->g->nv) throw Err(TNQS_ERR_INVALID, "bad vertex"); *n = state_site_size(s, v); });
}
int tnqs_set_message(tnqs_handle h, int src, int dst, const void* host, int chi) {
    return guard([&] { if (!host) throw Err(TNQS_ERR_INVALID, "set_message: null"); state_set_message(S(h), src, dst, host, chi); });
}
int tnqs_get_message(tnqs_handle h, int src, int dst, void* host, int chi) {
    return guard([&] { if (!host) throw Err(TNQS_ERR_INVALID, "get_message: null"); state_get_message(S(h), src, dst, host, chi); });
}
int tnqs_bond_dim(tnqs_handle h, int u, int v, int* chi) {
    return guard([&] { State* s = S(h); int e = s->g->edge(u, v); if (e < 0) throw Err(TNQS_ERR_INVALID, "bond_dim: not an edge"); *chi = s->chi[e]; });
}
int tnqs_maxvirtualdim(tnqs_handle h, int* chi) {
    return guard([&] { State* s = S(h); int m = 1; for (int c : s->chi) m = c > m ? c : m; *chi = m; });
}

int tnqs_bp_update(tnqs_handle h, const tnqs_bp_opts* o, int* niter, double* diff) {
    return guard([&] { State* s = S(h); s->stats = tnqs_apply_stats{}; bp_update(s, o, niter, diff); });
}
int tnqs_apply_gates(tnqs_handle h, int ngates, const int32_t* nverts, const int32_t* verts, const double* mats,
                     const tnqs_apply_opts* opts, const tnqs_bp_opts* bp, double* errs, tnqs_apply_stats* stats) {
    return guard([&] {
        State* s = S(h);
        if (ngates < 0 || (ngates > 0 && (!nverts || !verts || !mats))) throw Err(TNQS_ERR_INVALID, "apply_gates: null argument");
        apply_gates(s, ngates, nverts, verts, mats, opts, bp, errs);
        if (stats) *stats = s->stats;
    });
}
int tnqs_truncate(tnqs_handle h, int maxdim, double cutoff, int normalize, int ngroups, const int32_t* offs,
                  const int32_t* eu, const int32_t* ev, const tnqs_bp_opts* bp, tnqs_apply_stats* stats) {
    return guard([&] { State* s = S(h); truncate_bp(s, maxdim, cutoff, normalize, ngroups, offs, eu, ev, bp); if (stats) *stats = s->stats; });
}
int tnqs_rdm_1site(tnqs_handle h, int v, double* out) {
    return guard([&] { if (!out) throw Err(TNQS_ERR_INVALID, "rdm_1site: null"); rdm_1site(S(h), v, out); });
}
int tnqs_expect_1site(tnqs_handle h, int v, const double* op, double* out) {
    return guard([&] {
        State* s = S(h);
        if (!op || !out) throw Err(TNQS_ERR_INVALID, "expect_1site: null");
        if (v < 0 || v >= s->g->nv) throw Err(TNQS_ERR_INVALID, "expect_1site: bad vertex");
        int d = s->d[v];
        std::vector<double> r(2 * (size_t)d * d);
        rdm_1site(s, v, r.data());
        double nre = 0, nim = 0, tre = 0, tim = 0;
        for (int sp = 0; sp < d; ++sp) for (int si = 0; si < d; ++si) {
            double ore = op[2 * (sp + d * si)], oim = op[2 * (sp + d * si) + 1];
            double rre = r[2 * (si + d * sp)], rim = r[2 * (si + d * sp) + 1];
            nre += ore * rre - oim * rim; nim += ore * rim + oim * rre;
        }
        for (int si = 0; si < d; ++si) { tre += r[2 * (si + d * si)]; tim += r[2 * (si + d * si) + 1]; }
        double den = tre * tre + tim * tim;
        out[0] = (nre * tre + nim * tim) / den; out[1] = (nim * tre - nre * tim) / den;
    });
}
int tnqs_expect_region(tnqs_handle h, int nr, const int32_t* rv, const int32_t* parent, const double* ops, double* out4) {
    return guard([&] { expect_region(S(h), nr, rv, parent, ops, out4); });
}
int tnqs_vertex_scalars(tnqs_handle h, double* out) { return guard([&] { if (!out) throw Err(TNQS_ERR_INVALID, "vertex_scalars: null"); vertex_scalars(S(h), out); }); }
int tnqs_edge_scalars(tnqs_handle h, double* out) { return guard([&] { if (!out) throw Err(TNQS_ERR_INVALID, "edge_scalars: null"); edge_scalars(S(h), out); }); }
int tnqs_rescale(tnqs_handle h) { return guard([&] { rescale(S(h)); }); }
int tnqs_rescale_messages(tnqs_handle h, int n_edges, const int32_t* eu, const int32_t* ev) {
    return guard([&] { if (n_edges < 0) throw Err(TNQS_ERR_INVALID, "rescale_messages: negative count"); rescale_messages(S(h), n_edges, eu, ev); });
}
int tnqs_rescale_vertices(tnqs_handle h, int n_vertices, const int32_t* verts) {
    return guard([&] { if (n_vertices < 0) throw Err(TNQS_ERR_INVALID, "rescale_vertices: negative count"); rescale_vertices(S(h), n_vertices, verts); });
}
int tnqs_symmetric_gauge(tnqs_handle h, double regularization) { return guard([&] { symmetric_gauge(S(h), regularization); }); }
int tnqs_expect_all(tnqs_handle h, const double* ops, double* out) {
    return guard([&] { if (!ops || !out) throw Err(TNQS_ERR_INVALID, "expect_all: null"); expect_all(S(h), ops, out); });
}

int tnqs_set_sharding(tnqs_handle h, int rank, int nranks, const int32_t* owner, tnqs_allgather_fn fn, void* ctx,
                      void* exch_dev, int64_t exch_bytes) {
    return guard([&] {
        State* s = S(h);
        if (nranks < 1 || rank < 0 || rank >= nranks) throw Err(TNQS_ERR_INVALID, "set_sharding: bad rank");
        if (nranks > 1) {
            if (!owner || !fn || !exch_dev || exch_bytes <= 0) throw Err(TNQS_ERR_INVALID, "set_sharding: owner, callback and exchange buffer are required for nranks > 1");
            for (int v = 0; v < s->g->nv; ++v) if (owner[v] < 0 || owner[v] >= nranks) throw Err(TNQS_ERR_INVALID, "set_sharding: owner out of range");
            s->owner.assign(owner, owner + s->g->nv);
        } else s->owner.clear();
        // one-site gates deferred while the handle was unsharded are applied NOW, while every tensor is still here: afterwards an accessor only
        // touches the vertices its rank owns, and a pending set that differs between ranks would give the two ranks of a gate different thetas
        materialize_pending_all(s);
        s->comm.reset();       // callback transport from here on: a communicator of an earlier tnqs_set_sharding_rccl must not keep serving exchange()
        s->rank = rank; s->nranks = nranks; s->ag_fn = fn; s->ag_ctx = ctx; s->exch = exch_dev; s->exch_bytes = (size_t)exch_bytes;
        if (nranks > 1) for (int v = 0; v < s->g->nv; ++v) if (!s->owns(v)) { s->site[v] = nullptr; s->sscale[v] = nullptr; }   // only owners hold site tensors
    });
}

int tnqs_rccl_unique_id(void* out128) { return guard([&] { if (!out128) throw Err(TNQS_ERR_INVALID, "rccl_unique_id: null output"); rccl_unique_id(out128); }); }
int tnqs_set_sharding_rccl(tnqs_handle h, int rank, int nranks, const int32_t* owner, const void* unique_id128, int64_t exch_bytes) {
    return guard([&] { set_sharding_rccl(S(h), rank, nranks, owner, unique_id128, exch_bytes); });
}
int tnqs_sharding_stats(tnqs_handle h, int64_t* n_exchanges, int64_t* bytes_exchanged) {
    return guard([&] { State* s = S(h); if (n_exchanges) *n_exchanges = s->comm ? s->comm->n_exchanges : 0; if (bytes_exchanged) *bytes_exchanged = s->comm ? s->comm->bytes_exchanged : 0; });
}
int tnqs_rccl_selftest(int device, int64_t bytes) { return guard([&] { rccl_selftest(device, bytes); }); }
int tnqs_rccl_preflight(void) { return guard([&] { rccl_preflight(); }); }

int tnqs_profile_enable(tnqs_handle h, int on) { return guard([&] { S(h)->prof->on = on != 0; }); }
int tnqs_profile_get(tnqs_handle h, int cls, int64_t* launches, double* ms, double* bytes, double* flops) {
    return guard([&] {
        State* s = S(h);
        if (cls < 0 || cls >= TNQS_PROF_NCLASSES) throw Err(TNQS_ERR_INVALID, "profile_get: bad class");
        prof_collect(s);
        const ProfClass& pc = s->prof->cls[cls];
        if (launches) *launches = pc.launches;
        if (ms) *ms = pc.ms;
        if (bytes) *bytes = pc.bytes;
        if (flops) *flops = pc.flops;
    });
}
int tnqs_profile_reset(tnqs_handle h) {
    return guard([&] { State* s = S(h); prof_collect(s); for (auto& p : s->prof->cls) p = ProfClass{}; });
}

}  // extern "C"

// ---- kernel-level debug entry points (include/tnqs_debug.h) ---------------------------------------------------
#include "../../include/tnqs_debug.h"
#include "kernels.hpp"
namespace tnqs { void dbg_default_sequence(const State* s, std::vector<int>& src, std::vector<int>& dst);
                 std::shared_ptr<Graph> dbg_make_graph(int nv, int ne, const int32_t* es, const int32_t* ed);
                 void dbg_default_sequence_graph(const Graph& g, std::vector<int>& src, std::vector<int>& dst, std::vector<int>& level);
                 void dbg_pair(int C0, int NMID, int NHI, const void* in, const void* Mx, const void* My, void* out);
                 void dbg_gram_fused(int PA, int K, int PB, const void* X, const void* Y, const void* M, void* out);
                 void dbg_gauge_gram(int z, const int* chi, int bleg, const void* X, const void* M, void* out);
                 void dbg_jacobi(int dtype, int m, int n, void* A, void* V, int* sweeps);
                 void dbg_chol(int n, const void* G, void* L, void* W, int* fail, double tau);
                 void dbg_theta_svd_pre(int m, int n, int nq, void* A, const void* Q, void* V, int* sweeps, int copies, int reps, double* ms, double* phase_us, int cap);
                 void dbg_time_jacobi_f32(int m, int n, const void* A, int copies, int reps, double* ms, int* sweeps);
                 void dbg_pair_legs(int d, int z, const int* chi, int lx, int ly, const void* in, const void* Mx, const void* My, void* out);
                 void dbg_pair_gram2(int d, int z, const int* chi, int lx, int ly, const void* X, const void* Y, const void* Mx, const void* My, void* out_y, void* out_x);
                 void dbg_bench_plane(int which, int nsites, int lx, int ly, int reps, double* ms);
                 void dbg_pair_gram(int d, int z, const int* chi, int lx, int ly, const void* X, const void* Y, const void* M, void* out);
                 void dbg_fiber_gemm(int dtype, int D, int PA, int K, int PB, int Do, int No, const void* in, const void* X, void* out, double* norm2, int use_mfma);
                 void dbg_gram(int dtype, int D, int PA, int K, int PB, const void* X, const void* Y, void* out, int acc64, int use_mfma); }
extern "C" {
int tnqs_dbg_default_sequence(tnqs_handle h, int* src, int* dst, int cap, int* n_out) {
    return guard([&] { std::vector<int> a, b; dbg_default_sequence(S(h), a, b); *n_out = (int)a.size();
                       for (int i = 0; i < (int)a.size() && i < cap; ++i) { src[i] = a[i]; dst[i] = b[i]; } });
}
int tnqs_dbg_default_sequence_graph(int nv, int ne, const int32_t* esrc, const int32_t* edst, int* src, int* dst, int* level, int cap, int* n_out) {
    return guard([&] { if (nv < 0 || ne < 0 || (ne > 0 && (!esrc || !edst)) || !n_out) throw Err(TNQS_ERR_INVALID, "tnqs_dbg_default_sequence_graph: bad arguments");
                       auto g = dbg_make_graph(nv, ne, esrc, edst); std::vector<int> a, b, l; dbg_default_sequence_graph(*g, a, b, l); *n_out = (int)a.size();
                       for (int i = 0; i < (int)a.size() && i < cap; ++i) { if (src) src[i] = a[i]; if (dst) dst[i] = b[i]; if (level) level[i] = l[i]; } });
}
int tnqs_dbg_jacobi(int dtype, int m, int n, void* A, void* V, int* sweeps) { return guard([&] { dbg_jacobi(dtype, m, n, A, V, sweeps); }); }
int tnqs_dbg_theta_svd_pre(int m, int n, int nq, void* A, const void* Q, void* V, int* sweeps, int copies, int reps, double* ms, double* phase_us, int cap) { return guard([&] { dbg_theta_svd_pre(m, n, nq, A, Q, V, sweeps, copies, reps, ms, phase_us, cap); }); }
int tnqs_dbg_time_jacobi_f32(int m, int n, const void* A, int copies, int reps, double* ms, int* sweeps) { return guard([&] { dbg_time_jacobi_f32(m, n, A, copies, reps, ms, sweeps); }); }
int tnqs_dbg_chol(int n, const void* G, void* L, void* W, int* fail, double tau) { return guard([&] { dbg_chol(n, G, L, W, fail, tau); }); }
int tnqs_dbg_fiber_gemm(int dtype, int D, int PA, int K, int PB, int Do, int No, const void* in, const void* X, void* out, double* norm2, int use_mfma) {
    return guard([&] { dbg_fiber_gemm(dtype, D, PA, K, PB, Do, No, in, X, out, norm2, use_mfma); });
}
int tnqs_dbg_pair(int C0, int NMID, int NHI, const void* in, const void* Mx, const void* My, void* out) { return guard([&] { dbg_pair(C0, NMID, NHI, in, Mx, My, out); }); }
int tnqs_dbg_pair_legs(int d, int z, const int* chi, int lx, int ly, const void* in, const void* Mx, const void* My, void* out) { return guard([&] { dbg_pair_legs(d, z, chi, lx, ly, in, Mx, My, out); }); }
int tnqs_dbg_pair_gram2(int d, int z, const int* chi, int lx, int ly, const void* X, const void* Y, const void* Mx, const void* My, void* out_y, void* out_x) {
    return guard([&] { dbg_pair_gram2(d, z, chi, lx, ly, X, Y, Mx, My, out_y, out_x); });
}
int tnqs_dbg_bench_plane(int which, int nsites, int lx, int ly, int reps, double* ms) { return guard([&] { dbg_bench_plane(which, nsites, lx, ly, reps, ms); }); }
int tnqs_dbg_pair_gram(int d, int z, const int* chi, int lx, int ly, const void* X, const void* Y, const void* M, void* out) { return guard([&] { dbg_pair_gram(d, z, chi, lx, ly, X, Y, M, out); }); }
int tnqs_dbg_gram_fused(int PA, int K, int PB, const void* X, const void* Y, const void* M, void* out) { return guard([&] { dbg_gram_fused(PA, K, PB, X, Y, M, out); }); }
int tnqs_dbg_gauge_gram(int z, const int* chi, int bleg, const void* X, const void* M, void* out) { return guard([&] { dbg_gauge_gram(z, chi, bleg, X, M, out); }); }
int tnqs_dbg_gram(int dtype, int D, int PA, int K, int PB, const void* X, const void* Y, void* out, int acc64, int use_mfma) {
    return guard([&] { dbg_gram(dtype, D, PA, K, PB, X, Y, out, acc64, use_mfma); });
}
}
