// engine_gates.cpp -- apply_gates (src/Apply/apply_gates.jl:46-143), simple_update as batched launches, truncate (src/truncate.jl).
#include "engine_internal.hpp"

namespace tnqs {

// ---------------------------------------------------------------------------------------------------------------
// gates
// ---------------------------------------------------------------------------------------------------------------
struct Gate1 { int v; const double* mat; };
struct Gate2 { int v1, v2; const double* mat; int index; };

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
template <class T> static void apply_one_site_batch(State* s, const std::vector<Gate1>& gates_in, bool normalize, bool force);
// apply the pending one-site gates of `verts` (State::pend1) in one streaming pass; they are unitary, so norms, pending scale factors
// and unit_norm stay as they are
void materialize_pending(State* s, const std::vector<int>& verts) {
    std::vector<std::vector<double>> mats; std::vector<Gate1> gs; std::vector<char> norm_was;
    for (int v : verts) {
        if (v < 0 || v >= (int)s->pend1.size() || s->pend1[v].empty()) continue;
        mats.push_back(std::move(s->pend1[v])); s->pend1[v].clear(); norm_was.push_back(s->unit_norm[v]);
        gs.push_back(Gate1{v, nullptr});
    }
    if (gs.empty()) return;
    for (size_t k = 0; k < gs.size(); ++k) gs[k].mat = mats[k].data();
    HIPCHK(hipSetDevice(s->device));
    if (s->dtype == TNQS_C64) apply_one_site_batch<float>(s, gs, false, true); else apply_one_site_batch<double>(s, gs, false, true);
    for (size_t k = 0; k < gs.size(); ++k) s->unit_norm[gs[k].v] = norm_was[k];
}
void materialize_pending_all(State* s) {
    std::vector<int> all(s->site.size()); std::iota(all.begin(), all.end(), 0);
    materialize_pending(s, all);
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
        const NormFactorItem* d = upload_small(s, nf);
        { ProfScope ps(s, TNQS_PROF_SMALL, 0, 0); launch_norm_factor(s->stream, d, (int)nf.size()); }
        for (size_t i = 0; i < verts.size(); ++i) { s->site[verts[i]] = outs[i]; s->sscale[verts[i]] = sub_buffer(fac, 256 * i, 8); }
    } else {
        for (size_t i = 0; i < verts.size(); ++i) s->site[verts[i]] = outs[i];       // a pending factor of the input carries over (linear map)
    }
}

// d x d complex matrices, column-major [s' + d s] as (re, im) pairs
static bool is_unitary(const double* m, int d, double tol) {
    for (int a = 0; a < d; ++a) for (int b = 0; b < d; ++b) {
        double re = 0, im = 0;                                    // (G^dagger G)[a][b] = sum_s conj(G[s][a]) G[s][b]
        for (int t = 0; t < d; ++t) { const double* x = m + 2 * (t + d * a); const double* y = m + 2 * (t + d * b); re += x[0] * y[0] + x[1] * y[1]; im += x[0] * y[1] - x[1] * y[0]; }
        if (std::fabs(re - (a == b ? 1.0 : 0.0)) > tol || std::fabs(im) > tol) return false;
    }
    return true;
}
static std::vector<double> matmul_dd(const double* a, const double* b, int d) {       // a . b
    std::vector<double> c(2 * (size_t)d * d, 0.0);
    for (int i = 0; i < d; ++i) for (int j = 0; j < d; ++j) {
        double re = 0, im = 0;
        for (int t = 0; t < d; ++t) { const double* x = a + 2 * (i + d * t); const double* y = b + 2 * (t + d * j); re += x[0] * y[0] - x[1] * y[1]; im += x[0] * y[1] + x[1] * y[0]; }
        c[2 * (i + d * j)] = re; c[2 * (i + d * j) + 1] = im;
    }
    return c;
}

template <class T> static void apply_one_site_batch(State* s, const std::vector<Gate1>& gates_in, bool normalize, bool force) {
    if (gates_in.empty()) return;
    const size_t esz = s->esz();
    // ---- deferral (State::pend1): a unitary gate on a tensor that needs no normalisation pass is only recorded -- BP does not see it, the
    // next two-site gate on the vertex absorbs it.  Anything else is applied now, composed with what was pending on the vertex.
    std::vector<std::vector<double>> composed; composed.reserve(gates_in.size());
    std::vector<Gate1> gates;
    const bool may_defer = !force && defer_site1() && (!s->sharded() || s->in_apply);      // (sharded handles: only inside apply_gates, State::in_apply)
    // unitary to what the state's precision resolves: a gate the caller built in complex64 is unitary to ~1e-7 only, and BP messages of a
    // ComplexF32 state do not see a deviation of that size either
    const double utol = s->dtype == TNQS_C64 ? 1e-6 : 1e-13;
    for (auto& g1 : gates_in) {
        const int d = s->d[g1.v];
        std::vector<double>& pend = s->pend1[g1.v];
        if (may_defer && is_unitary(g1.mat, d, utol) && (!normalize || s->unit_norm[g1.v])) {
            pend = pend.empty() ? std::vector<double>(g1.mat, g1.mat + 2 * (size_t)d * d) : matmul_dd(g1.mat, pend.data(), d);
            s->stats.n_deferred_1site += 1;
            continue;
        }
        if (!pend.empty()) { composed.push_back(matmul_dd(g1.mat, pend.data(), d)); pend.clear(); gates.push_back(Gate1{g1.v, composed.back().data()}); }
        else gates.push_back(g1);
        s->unit_norm[g1.v] = normalize ? 1 : 0;
    }
    if (gates.empty()) return;
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

// allow_spec: the batch may be enqueued WITHOUT its host round trip when its outcome is predictable (see `spec` below); it then leaves a Check behind (engine.hpp)
template <class T> static void apply_two_site_batch(State* s, const std::vector<Gate2>& gates_in, const tnqs_apply_opts& ao, double* errs, bool allow_spec = false) {
    if (gates_in.empty()) return;
    const Graph& g = *s->g;
    const size_t esz = s->esz();
    const bool sharded = s->sharded();
    // ---- run on assumptions?  Every bond of the batch already sits at its cap (a saturated evolution: the new bond dimension is the cap again unless the cutoff
    // bites), the whole chain can be sized from upper bounds (ComplexF32, every theta within the LDS-resident kernels), single rank.  Then the read-back of the
    // batch -- ranks, new bond dimensions, statuses, truncation errors, every fallback flag -- is only STAGED, the epilogue is launched for bond dimension = cap,
    // and the verification happens when the staged copy has arrived (settle).  Anything the assumptions do not cover fails the check and the batch is redone.
    // (ComplexF64 with the second factorisation pass: an evolution flags ill-conditioned sites in almost every batch -- 40 per layer of a 4 x 4 lattice --, which is a
    //  read-back the careful route needs anyway: nothing to run ahead on)
    bool spec = allow_spec && !sharded && ao.maxdim > 0 && (std::is_same<T, float>::value || !use_qr2());
    for (size_t k = 0; k < gates_in.size() && spec; ++k) {
        const int v1 = gates_in[k].v1, v2 = gates_in[k].v2; const int chi = s->chi[g.edge(v1, v2)];
        const int Mr = std::max(s->d[v1], s->d[v2]) * std::max(s->d[v1], s->d[v2]) * chi, Nc = std::min(s->d[v1], s->d[v2]) * std::min(s->d[v1], s->d[v2]) * chi;
        spec = chi == ao.maxdim && Nc >= chi && Mr <= 256 && jacobi_lds(jacobi_lds_bytes(Mr, Nc, false, esz)) > 0;
    }
    const double sqrt_cutoff = ao.sqrt_cutoff >= 0 ? ao.sqrt_cutoff : 10.0 * (s->dtype == TNQS_C64 ? 1.1920928955078125e-07 : 2.220446049250313e-16);
    const int ng = (int)gates_in.size();
    // pending one-site gates of the gate vertices are absorbed into the gate matrix: g' = g . (G1 (x) G2) is exactly what simple_update sees
    // when the one-site gates were applied to the tensors first (State::pend1); cleared once the batch has replaced the tensors
    std::vector<std::vector<double>> absorbed; absorbed.reserve(gates_in.size());
    std::vector<Gate2> gates = gates_in;
    for (auto& g2 : gates) {
        const std::vector<double>& p1 = s->pend1[g2.v1]; const std::vector<double>& p2 = s->pend1[g2.v2];
        if (p1.empty() && p2.empty()) continue;
        const int d1 = s->d[g2.v1], d2 = s->d[g2.v2], dd = d1 * d2;
        std::vector<double> kron(2 * (size_t)dd * dd, 0.0);                 // (G1 (x) G2)[(t1 t2),(s1 s2)], first vertex most significant
        for (int t1 = 0; t1 < d1; ++t1) for (int s1 = 0; s1 < d1; ++s1) for (int t2 = 0; t2 < d2; ++t2) for (int s2 = 0; s2 < d2; ++s2) {
            const double ar = p1.empty() ? (t1 == s1 ? 1.0 : 0.0) : p1[2 * (t1 + d1 * s1)], ai = p1.empty() ? 0.0 : p1[2 * (t1 + d1 * s1) + 1];
            const double br = p2.empty() ? (t2 == s2 ? 1.0 : 0.0) : p2[2 * (t2 + d2 * s2)], bi = p2.empty() ? 0.0 : p2[2 * (t2 + d2 * s2) + 1];
            const size_t e = (size_t)(t1 * d2 + t2) + (size_t)dd * (s1 * d2 + s2);
            kron[2 * e] = ar * br - ai * bi; kron[2 * e + 1] = ar * bi + ai * br;
        }
        absorbed.push_back(matmul_dd(g2.mat, kron.data(), dd));
        g2.mat = absorbed.back().data();
    }
    PhaseScope phase_scope(s, TNQS_PROF_PHASE_GATE_BATCH);
    HostTimer ht_a(3);                 // TNQS_HOST_TIMING=1: host time of the batch up to the first read-back (3), between the read-backs (4), after them (5)
    if (!ao.normalize_tensors) {       // without the final normalisation the result scales with the inputs: apply pending factors first
        std::vector<int> vs; for (auto& g2 : gates) { vs.push_back(g2.v1); vs.push_back(g2.v2); }
        materialize_scale(s, vs);
    }
    struct SiteJob { int v, other, bleg; bool owned; SD sd; std::vector<int> env_idx; std::vector<int> env_leg; };
    std::vector<SiteJob> sj(2 * (size_t)ng);
    std::vector<char> part(ng, 0);                  // this rank runs the small algebra of the gate
    HostTimer ht_s1(8);
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
    // Everything of a batch the host zeroes and reads back lives in ONE buffer -- [info | low-rank failure flags (two passes) | truncation errors | Cholesky failure
    // flags | message-eigenvalue flags]: one memset where there were four, one device-to-host copy where there were four (5 us of stream time each in a chain
    // of 20-60 us kernels)
    int npg0 = 0; for (int gi = 0; gi < ng; ++gi) npg0 += part[gi] ? 1 : 0;
    const size_t rb_info = 0, rb_low = rb_info + round256((size_t)npg0 * 32), rb_low2 = rb_low + round256((size_t)npg0 * 4), rb_terr = rb_low2 + round256((size_t)npg0 * 4),
                 rb_chol = rb_terr + round256((size_t)npg0 * 8), rb_env = rb_chol + round256(sj.size() * sizeof(int)), rb_total = rb_env + round256(h_flags.size() * sizeof(int));
    Buf d_rb = dalloc(s, rb_total);
    Buf d_flags = sub_buffer(d_rb, rb_env, h_flags.size() * sizeof(int));
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
        // everything up to here was host preparation.  The environment chain (small kernels that only READ the messages and write fresh buffers) is
        // enqueued behind the pending BP sweep right away; the verdict is awaited after that, in front of the tensor passes
        if (!envs.empty()) {
            const EnvItem* de = upload_small(s, ei); const JacobiItem* dj = upload_small(s, ji); const EnvFinishItem* df = upload_small(s, fi);
            { ProfScope ps(s, TNQS_PROF_SMALL, 0, 0); launch_env_prepare<T>(s->stream, de, (int)ei.size()); }
            size_t lds = 0; for (auto& r : envs) lds = std::max(lds, jacobi_lds_bytes(r.n, r.n, true, 16));
            { ProfScope ps(s, TNQS_PROF_JACOBI, 0, 0); launch_jacobi<double>(s->stream, dj, (int)ji.size(), 60, jacobi_lds(lds), mmax_of(ji)); }
            { ProfScope ps(s, TNQS_PROF_SMALL, 0, 0); launch_env_finish<T>(s->stream, df, (int)fi.size()); }
        }
    }
    ht_s1.stop(); HostTimer ht_s2(9);
    // ---- 2. gauge: psi~ = psi x_outer M^{1/2}  (simple_update.jl:43-44), owned sites only -----------------------------
    std::vector<int> own_idx;                        // indices into sj of the owned sites
    for (size_t i = 0; i < sj.size(); ++i) if (sj[i].owned) own_idx.push_back((int)i);
    std::vector<Chain> chains(own_idx.size());
    // bulk shape (d = 2, chi = 32, three gauged legs): the LAST gauge leg -- the fastest outer leg, which the two-leg kernel leaves over -- is
    // absorbed inside the Gram kernel instead of in a pass of its own (kernels_gate.hip); fused_M[q] = its matrix
    std::vector<const void*> fused_M(own_idx.size(), nullptr);
    const bool fuse_on = true;
    for (size_t q = 0; q < own_idx.size(); ++q) {
        const SiteJob& j = sj[own_idx[q]];
        Chain& c = chains[q]; c.v = j.v; c.src = s->site[j.v]->p; c.sd = j.sd;
        for (size_t e = 0; e < j.env_idx.size(); ++e) c.steps.push_back({j.env_leg[e], envs[j.env_idx[e]].msq});
        if (fuse_on && std::is_same<T, float>::value && use_mfma() && use_pair() && c.steps.size() == 3 && c.steps[0].first == (j.bleg == 0 ? 1 : 0) &&
            j.sd.n / ((size_t)j.sd.d * j.sd.chi[j.bleg]) >= (size_t)j.sd.d * j.sd.chi[j.bleg] &&
            gauge_gram64_covers(j.sd.d, j.sd.z, j.sd.chi.data(), j.bleg, c.steps[0].first)) {
            fused_M[q] = c.steps[0].second; c.steps.erase(c.steps.begin());
        }
        // 16-dimensional legs (degree 6, chi = 16: five gauge legs): the same with mfma_gauge_gram32_kernel -- four legs in two two-leg passes, the fifth inside the Gram
        else if (fuse_on && std::is_same<T, float>::value && use_mfma() && use_pair() && c.steps.size() >= 1 && c.steps[0].first == (j.bleg == 0 ? 1 : 0) &&
                 j.sd.n >= (size_t)(1u << 14) && gauge_gram32_covers(j.sd.d, j.sd.z, j.sd.chi.data(), j.bleg, c.steps[0].first)) {
            fused_M[q] = c.steps[0].second; c.steps.erase(c.steps.begin());
        }
    }
    if (!spec) settle(s, true);      // the careful route waits for what is pending (the BP update's verdict) in front of its tensor passes: a failure costs the environment chain only
    run_chains<T>(s, chains, TNQS_PROF_GATE_MODEPROD);
    std::vector<Buf> GA(sj.size()), GV(sj.size()), GW(sj.size()); std::vector<char> is_chol(sj.size(), 0), is_small(sj.size(), 0), small_done(sj.size(), 0);
    auto nof = [&](size_t i) { return sj[i].sd.d * sj[i].sd.chi[sj[i].bleg]; };
    auto small_shape = [&](size_t i) { const int n = nof(i); return sj[i].sd.n / (size_t)n < (size_t)n && n <= 256 && use_small_svd(); };
    // ---- 2b. sites with fewer fibers than columns (corners, low bond dimensions) are factorised without a Gram matrix, by a one-sided Jacobi of
    // the small matricised psi~ (f64): three dependent launches (0.2 ms on a 7 x 7 lattice) that only need the gauged tensor.  They start NOW on a
    // side stream, under the Gram pass, instead of in front of the Cholesky kernels afterwards (single rank) -------------------------------------
    hipEvent_t ev_small = nullptr;
    // (whatever happens before the regular wait below -- an exception in the Gram / reduce / Cholesky steps -- the main stream is ordered behind the side
    //  stream before this frame releases M / GA / GV to the stream-ordered pool: round-4 advisor finding)
    struct SmallJoin { State* s; hipEvent_t& ev; ~SmallJoin() { if (ev) (void)hipStreamWaitEvent(s->stream, ev, 0); } } small_join{s, ev_small};
    if (!sharded) {
        std::vector<SmallSvdItem> si; std::vector<JacobiItem> sji;
        for (size_t q = 0; q < own_idx.size(); ++q) {
            const size_t i = own_idx[q];
            if (!small_shape(i) || fused_M[q]) continue;
            const int n = nof(i); const size_t Nout = sj[i].sd.n / (size_t)n;
            GA[i] = dalloc(s, (size_t)n * n * 16); GV[i] = dalloc(s, (size_t)n * n * 16); GW[i] = GV[i]; is_small[i] = 1; small_done[i] = 1;
            Buf M = dalloc(s, (size_t)n * Nout * 16); s->keepalive.push_back(M);
            const SD& sd = sj[i].sd; const int b = sj[i].bleg;
            si.push_back(SmallSvdItem{chains[q].result, M->p, GA[i]->p, GV[i]->p, sd.d, (int)(sd.pre(b) / sd.d), sd.chi[b], (int)sd.post(b)});
            sji.push_back(JacobiItem{M->p, nullptr, n, (int)Nout, nullptr});
        }
        if (!si.empty()) {
            const SmallSvdItem* ds = upload_small(s, si); const JacobiItem* dj = upload_small(s, sji);
            hipStream_t side = aux_stream_of(s);
            HIPCHK(hipEventRecord(s->ev_fork, s->stream)); HIPCHK(hipStreamWaitEvent(side, s->ev_fork, 0));
            launch_small_svd_prepare<T>(side, ds, (int)si.size());
            size_t lds = 0; for (auto& j : sji) lds = std::max(lds, jacobi_lds_bytes(j.m, j.n, false, 16));
            launch_jacobi<double>(side, dj, (int)sji.size(), 60, jacobi_lds(lds), mmax_of(sji));
            launch_small_svd_finish(side, ds, (int)si.size());
            HIPCHK(hipEventRecord(s->ev_join, side)); ev_small = s->ev_join;
        }
    }
    ht_s2.stop(); HostTimer ht_s3(10);
    // ---- 3. G = psi~^dagger psi~ over the outer legs, f64 accumulation (replaces the thin QR, simple_update.jl:45-48) --
    std::vector<GramJob> jobs; std::vector<int> job_of(own_idx.size(), -1);
    for (size_t q = 0; q < own_idx.size(); ++q) {
        const SiteJob& sjq = sj[own_idx[q]];
        if (small_done[own_idx[q]]) continue;              // factorised without a Gram matrix (2b)
        GramJob j{}; j.X = chains[q].result; j.Y = chains[q].result; j.sd = sjq.sd; j.leg = sjq.bleg; j.keep_site = true; j.M = fused_M[q];
        job_of[q] = (int)jobs.size(); jobs.push_back(j);
    }
    {   // the fused and the plain Gram are different kernels: two batches, job order kept
        std::vector<GramJob> jf, jf16, jp; std::vector<size_t> idf, idf16, idp;
        for (size_t q = 0; q < jobs.size(); ++q) {
            if (jobs[q].M && jobs[q].sd.chi[jobs[q].leg] == 16) { jf16.push_back(jobs[q]); idf16.push_back(q); }
            else if (jobs[q].M) { jf.push_back(jobs[q]); idf.push_back(q); } else { jp.push_back(jobs[q]); idp.push_back(q); }
        }
        run_grams<T, double>(s, jf, TNQS_PROF_GATE_GRAM);
        run_grams<T, double>(s, jf16, TNQS_PROF_GATE_GRAM);
        run_grams<T, double>(s, jp, TNQS_PROF_GATE_GRAM);
        for (size_t q = 0; q < jf.size(); ++q) jobs[idf[q]] = jf[q];
        for (size_t q = 0; q < jf16.size(); ++q) jobs[idf16[q]] = jf16[q];
        for (size_t q = 0; q < jp.size(); ++q) jobs[idp[q]] = jp[q];
    }
    // G slots: in the sharded case both ranks of a gate that STRADDLES two ranks need G1 and G2 -> all-gather those (a gate whose two sites live on
    // one rank is that rank's business alone: round 4 gathered the Gram matrices of every site, 25 MB per colour batch of the 20 x 20 lattice, of which
    // a contiguous partition needs 20 gates' worth per cut).  The same layout serves the Gram matrices of the second factorisation pass further down.
    // Every rank derives the same slots from the gate list and the owner map, so the collective is entered by all ranks or by none.
    std::vector<size_t> slot(sj.size(), 0); size_t stride = 0;
    std::vector<char> cross(sj.size(), 0);        // the site's gate partner lives on another rank
    {
        std::vector<size_t> rank_bytes(s->nranks, 0);
        if (sharded) {
            for (size_t i = 0; i < sj.size(); ++i) {
                if (s->owner[sj[i].v] == s->owner[sj[i].other]) continue;
                cross[i] = 1;
                int r = s->owner[sj[i].v]; slot[i] = rank_bytes[r]; rank_bytes[r] += round256((size_t)nof(i) * nof(i) * 16);
            }
            for (size_t b : rank_bytes) stride = std::max(stride, b);
            if (stride) check_exchange(s, stride);
        }
        std::vector<ReduceItem> ri; int elems = 0;
        for (size_t q = 0; q < own_idx.size(); ++q) {
            if (job_of[q] < 0) continue;
            const GramJob& jb = jobs[job_of[q]];
            size_t i = own_idx[q]; int n = jb.KK; size_t nn = (size_t)n * n;
            GA[i] = dalloc(s, nn * 16);
            void* dst = cross[i] ? (void*)(reinterpret_cast<char*>(s->exch) + (size_t)s->rank * stride + slot[i]) : GA[i]->p;
            ri.push_back(ReduceItem{jb.partial->p, dst, (int)nn, jb.nchunks, 1, elems}); elems += (int)nn;
        }
        const ReduceItem* dr = upload(s, ri);
        { ProfScope ps(s, TNQS_PROF_SMALL, 0, 0); launch_reduce<double, double>(s->stream, dr, (int)ri.size(), elems); }
        if (sharded && stride) {
            exchange(s, stride);
            // one private copy of the gathered block (the exchange buffer is reused by the record exchange of this batch); the G of
            // every site this rank needs is a view into it
            Buf G_keep = dalloc(s, std::max<size_t>(256, stride * (size_t)s->nranks));
            HIPCHK(hipMemcpyAsync(G_keep->p, s->exch, stride * (size_t)s->nranks, hipMemcpyDeviceToDevice, s->stream));
            for (size_t i = 0; i < sj.size(); ++i) {
                if (!part[i / 2] || !cross[i]) continue;
                size_t nn = (size_t)nof(i) * nof(i);
                GA[i] = sub_buffer(G_keep, (size_t)s->owner[sj[i].v] * stride + slot[i], nn * 16);
            }
        }
    }
    // R factor of psi~ = Q R from G = R^dagger R: Cholesky (R = L^dagger) where G has full rank by construction (at least as
    // many fibers as columns); the f64 Jacobi eigen factorisation R = Lambda^1/2 W^dagger otherwise, and for the whole batch when
    // a Cholesky pivot collapses (numerically rank-deficient G; the eigen path drops the null space, rank_tau in kernels.hpp)
    // ComplexF64, single rank: ill-conditioned sites get a second factorisation pass below, which sorts out what is signal and what is
    // noise among the smallest directions -- so the first pass keeps everything above the f64 noise floor instead of rank_tau
    const bool qr2 = !std::is_same<T, float>::value && use_qr2();
    auto tau_of = [&](int n) { return qr2 ? 1e-15 : rank_tau(std::is_same<T, float>::value, n); };
    // sites with fewer fibers than columns are factorised by their owner without a Gram matrix (small-SVD route) and never refined; the
    // criterion must not depend on ownership, every rank taking part in a gate has to reach the same decision
    std::vector<const void*> gauged_of(sj.size(), nullptr);      // psi~ of the owned sites
    for (size_t q = 0; q < own_idx.size(); ++q) gauged_of[own_idx[q]] = chains[q].result;
    Buf d_cholfail = sub_buffer(d_rb, rb_chol, std::max<size_t>(1, sj.size()) * sizeof(int));      // one flag per site: only the sites whose pivot collapsed are redone
    std::vector<int> h_cholfail(sj.size(), 0);
    auto factor_G = [&](bool allow_chol, bool fallback = false) {
        std::vector<JacobiItem> ji, sji; std::vector<EnvItem> idn; std::vector<CholItem> ci; std::vector<SmallSvdItem> si; int cmax = 1;
        if (!fallback) HIPCHK(hipMemsetAsync(d_cholfail->p, 0, std::max<size_t>(1, sj.size()) * sizeof(int), s->stream));
        for (size_t i = 0; i < sj.size(); ++i) {
            if (!part[i / 2] || small_done[i]) continue;
            if (fallback && !(is_chol[i] && h_cholfail[i])) continue;      // fallback pass: only the Cholesky sites whose pivot collapsed (the eigen sites are factorised, GA rotated in place)
            int n = nof(i);
            if (!GV[i]) GV[i] = dalloc(s, (size_t)n * n * 16);
            const size_t Nout = sj[i].sd.n / (size_t)n;
            const bool ch = allow_chol && n <= (use_chi64() ? 128 : 96) && Nout >= (size_t)n;
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
            const SmallSvdItem* ds = upload_small(s, si); const JacobiItem* dj = upload_small(s, sji);
            ProfScope ps(s, TNQS_PROF_JACOBI, 0, 0);
            launch_small_svd_prepare<T>(s->stream, ds, (int)si.size());
            size_t lds = 0; for (auto& j : sji) lds = std::max(lds, jacobi_lds_bytes(j.m, j.n, false, 16));
            launch_jacobi<double>(s->stream, dj, (int)sji.size(), 60, jacobi_lds(lds), mmax_of(sji));
            launch_small_svd_finish(s->stream, ds, (int)si.size());
        }
        if (!ci.empty()) {      // n <= 96: square LDS array; 96 < n <= 128 (chi = 64 sites): packed triangle
            const CholItem* dc = upload_small(s, ci); ProfScope ps(s, TNQS_PROF_JACOBI, 0, 0);
            if (cmax <= 96) launch_chol(s->stream, dc, (int)ci.size(), cmax); else launch_chol_packed(s->stream, dc, (int)ci.size(), cmax);
        }
        if (!ji.empty()) {
            const EnvItem* di = upload_small(s, idn);
            { ProfScope ps(s, TNQS_PROF_SMALL, 0, 0); launch_env_prepare<T>(s->stream, di, (int)idn.size()); }
            const JacobiItem* dj = upload_small(s, ji);
            size_t lds = 0; for (auto& j : ji) lds = std::max(lds, jacobi_lds_bytes(j.n, j.n, true, 16));
            { ProfScope ps(s, TNQS_PROF_JACOBI, 0, 0); launch_jacobi<double>(s->stream, dj, (int)ji.size(), 60, jacobi_lds(lds), mmax_of(ji)); }
        }
    };
    factor_G(use_chol());
    ht_s3.stop(); HostTimer ht_s4(11);
    // ---- 4. theta = gate . (R1 R2), SVD, truncation, X1 / X2  (simple_update.jl:51-59) -----------------------------
    struct GateWS { Buf lam1, lam2, idx1, idx2, theta, thetaV, theta0, X1, X2, S, lowA, lowB, lowG, lowL, lowW, lowQ, lowB1, lowG2, lowL2, lowLc; int n1, n2, chi, cap; };
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
        // theta is at most 512 x 512 (d^2 chi <= 512: chi <= 128 for qubits); up to 256 rows everything has an LDS or MFMA-preprocessed route, beyond
        // that the factorisations run in the global-memory Jacobi kernel (8 rows per lane)
        if (Mr > 512 || Nc > 512) throw Err(TNQS_ERR_UNSUPPORTED, "two-site gate: d^2*chi > 512 is not supported by the theta SVD kernels");
        int cap = std::min(Mr, Nc); if (ao.maxdim > 0) cap = std::min(cap, ao.maxdim);
        w.cap = cap; cap_max = std::max(cap_max, cap);
        x2_max = std::max(x2_max, (size_t)w.n2 * b.sd.d * cap * esz);
    }
    // SVD of theta: the right factor is never accumulated from the rotations (in f32 its orthogonality degrades with the
    // rotation count, ~1e-5 at 150 columns) but recovered from an unrotated copy: V = theta0^dagger (U S) S^-2
    const bool theta0_used = true;
    bool lowrank_on_batch = false;          // some gate of the batch carries operator-sum factors
    {
        std::vector<char> raw;
        std::vector<size_t> off(pg.size()), offA(pg.size(), 0), offB(pg.size(), 0); std::vector<int> kappa(pg.size(), 0);
        // ComplexF64 takes the route as well (round 4): B is orthogonalised by CholeskyQR2 (kernels.hpp LowQr2Item), and the 128 x 64 factor of a
        // chi = 32 gate fits the LDS-resident Jacobi where the 128 x 128 theta (256 KiB) ran in the global-memory kernel
        const bool lowrank_on = use_lowrank();
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
        // the per-gate workspaces (sixteen small buffers per gate, all of them dead at the end of the batch) are views into a few 4 MiB slabs: ~1100 pool round
        // trips per heavy-hex batch, 0.13 ms of host time in front of gate_theta, were what the chip waited for after the Cholesky kernels
        struct BatchArena { State* s; Buf cur; size_t off = 0, cap = 0;
            Buf get(size_t bytes) {
                const size_t b = round256(std::max<size_t>(bytes, 1));
                if (b > ((size_t)1 << 20)) return dalloc(s, bytes);
                if (!cur || off + b > cap) { cap = (size_t)4 << 20; cur = dalloc(s, cap); off = 0; }
                Buf v = sub_buffer(cur, off, bytes); off += b; return v;
            } } arena{s};
        for (size_t q = 0; q < pg.size(); ++q) {
            int gi = pg[q];
            GateWS& w = ws[gi]; GateItem& it = gitems[q];
            const SiteJob& a = sj[2 * gi]; const SiteJob& b = sj[2 * gi + 1];
            int Mr = w.n1 * a.sd.d, Nc = w.n2 * b.sd.d, cap = w.cap;
            w.lam1 = arena.get(w.n1 * 8); w.lam2 = arena.get(w.n2 * 8); w.idx1 = arena.get(w.n1 * 4); w.idx2 = arena.get(w.n2 * 4);
            w.theta = arena.get((size_t)Mr * Nc * esz); w.thetaV = arena.get((size_t)std::max(Mr, Nc) * std::max(Mr, Nc) * esz);
            if (theta0_used) w.theta0 = arena.get((size_t)Mr * Nc * esz);
            w.X1 = arena.get((size_t)w.n1 * a.sd.d * cap * esz); w.X2 = arena.get((size_t)w.n2 * b.sd.d * cap * esz);
            w.S = arena.get(cap * 8);
            it.GA1 = GA[2 * gi]->p; it.GV1 = GV[2 * gi]->p; it.GA2 = GA[2 * gi + 1]->p; it.GV2 = GV[2 * gi + 1]->p;
            it.GW1 = GW[2 * gi]->p; it.GW2 = GW[2 * gi + 1]->p; it.chol1 = is_chol[2 * gi]; it.chol2 = is_chol[2 * gi + 1];
            it.n1 = w.n1; it.n2 = w.n2; it.d1 = a.sd.d; it.d2 = b.sd.d; it.chi = w.chi;
            it.gate = reinterpret_cast<const double*>(d_gm + off[q]);
            it.kappa = 0; it.opA = it.opB = nullptr; it.lowA = it.lowB = it.lowG = nullptr; it.lowL = nullptr; it.lowfail = nullptr; it.lowW = nullptr; it.lowQ = nullptr;
            {   // the operator-sum factors A ((r1 d1) x K), B ((r2 d2) x K), K = kappa chi: theta = A B^T is formed from them (gate_theta_mm_kernel);
                // the low-rank route of the theta SVD (lowG / lowL) only where it can apply -- K below the theta columns and chol_kernel's size
                const int K = kappa[q] * w.chi;
                if (lowrank_on && kappa[q] > 0) {
                    w.lowA = arena.get((size_t)Mr * K * 16); w.lowB = arena.get((size_t)Nc * K * 16);
                    it.kappa = kappa[q]; it.opA = reinterpret_cast<const double*>(d_gm + offA[q]); it.opB = reinterpret_cast<const double*>(d_gm + offB[q]);
                    it.lowA = w.lowA->p; it.lowB = w.lowB->p; lowrank_on_batch = true;
                    const bool lds_fits = std::is_same<T, float>::value || jacobi_lds(jacobi_lds_bytes(Mr, K, false, esz)) > 0;      // ComplexF64: only where it buys the LDS route
                    if (K < Nc && K <= 128 && cap <= K && Mr >= Nc && lds_fits) {
                        w.lowG = arena.get((size_t)K * K * 16); w.lowL = arena.get((size_t)K * K * 16); w.lowW = arena.get((size_t)K * K * 16);
                        it.lowG = w.lowG->p; it.lowL = w.lowL->p;
                        // ComplexF32, factor of at most 128 x 64: the preconditioned SVD kernel builds V from Q = B L^-dagger (lowrank_m_kernel writes it)
                        if (std::is_same<T, float>::value && use_precond_svd() && theta_svd_pre_covers(Mr, K) && K <= 96) { w.lowQ = arena.get((size_t)Nc * K * 16); it.lowW = w.lowW->p; it.lowQ = w.lowQ->p; }
                        if (!std::is_same<T, float>::value) {
                            w.lowB1 = arena.get((size_t)Nc * K * 16); w.lowG2 = arena.get((size_t)K * K * 16); w.lowL2 = arena.get((size_t)K * K * 16); w.lowLc = arena.get((size_t)K * K * 16);
                        }
                    }
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
    // ---- epilogue plan (step 5) for the register-direct MFMA kernel: items, output buffers, uploaded descriptors.  Built twice at most: speculatively
    // BEFORE the read-back of the batch -- assuming every new bond dimension equals its cap and no projector pass is needed, which is the steady state of a
    // saturated evolution -- so that after the synchronisation the epilogue is launched at once instead of after 0.25 ms of host preparation with an idle
    // chip (20x20: 380 items and output buffers); and again after the read-back when the assumption did not hold --------------------------------------
    struct RgGroup { int kk = 0; std::vector<FiberItem> sub; std::vector<int> sv, stb, snt; std::vector<Buf> so; std::vector<size_t> sn; int wgs = 0; Buf npr; const FiberItem* d = nullptr; };
    struct RgPlan { bool valid = false; std::vector<RgGroup> groups; std::vector<char> via; double rby = 0, rfl = 0; };
    auto plan_rowgemm = [&](auto chi_of, auto in_of, std::vector<char>* skip) {
        RgPlan P; P.via.assign(own_idx.size(), 0);
        std::vector<FiberItem> rg; std::vector<int> rverts; std::vector<Buf> routs; std::vector<size_t> rne;
        if (std::is_same<T, float>::value && use_mfma())
            for (size_t q = 0; q < own_idx.size(); ++q) {
                if (skip && (*skip)[q]) continue;
                size_t i = own_idx[q]; int gi = (int)i / 2; int chin = chi_of(gi); const SiteJob& j = sj[i];
                FiberItem it{};
                it.D = j.sd.d; it.PA = (int)(j.sd.pre(j.bleg) / j.sd.d); it.K = j.sd.chi[j.bleg]; it.PB = (int)j.sd.post(j.bleg); it.Do = j.sd.d; it.No = chin;
                if (!rowgemm_covers(it) || it.D != 2 || (it.K == 64 && !use_chi64())) continue;
                const size_t nout = j.sd.n / it.K * chin;
                Buf out = dalloc(s, nout * esz);
                it.in = in_of(q); it.out = out->p; it.X = (i & 1) ? ws[gi].X2->p : ws[gi].X1->p;
                rowgemm_tiles(it); it.want_norm = ao.normalize_tensors ? 1 : 0;
                rg.push_back(it); rverts.push_back(j.v); routs.push_back(out); rne.push_back(nout);
                P.rby += (double)(j.sd.n + nout) * esz; P.rfl += 8.0 * j.sd.n * j.sd.d * chin; P.via[q] = 1; if (skip) (*skip)[q] = 1;
            }
        for (int kk : {64, 32}) {
            RgGroup G; G.kk = kk; double st = 0;
            for (size_t q = 0; q < rg.size(); ++q) if (rg[q].K == kk) { G.sub.push_back(rg[q]); G.sv.push_back(rverts[q]); G.so.push_back(routs[q]); G.sn.push_back(rne[q]); st += (double)rg[q].nta * rg[q].ntb; }
            if (G.sub.empty()) continue;
            int tpw = (int)std::max(4.0, std::min(32.0, st / 2048.0)); tpw &= ~3;
            for (auto& it : G.sub) { const int nwg = (it.nta * it.ntb + tpw - 1) / tpw; it.tpw = tpw; it.tile_begin = G.wgs; G.stb.push_back(G.wgs); G.snt.push_back(nwg); G.wgs += nwg; }
            G.npr = dalloc(s, (size_t)G.wgs * sizeof(double));
            G.d = upload(s, G.sub);
            P.groups.push_back(std::move(G));
        }
        P.valid = true;
        return P;
    };
    RgPlan spec_plan;
    // per-gate (r1, r2, chi', status, sweeps, wide, -, -) and truncation error live in two contiguous arrays: one D2H each
    Buf d_info_all = sub_buffer(d_rb, rb_info, std::max<size_t>(1, (size_t)npg * 32));      // (zeroed by run_theta)
    Buf d_terr_all = sub_buffer(d_rb, rb_terr, std::max<size_t>(1, (size_t)npg * 8));
    for (int q = 0; q < npg; ++q) { gitems[q].info = reinterpret_cast<int*>(d_info_all->p) + 8 * q; gitems[q].truncerr = reinterpret_cast<double*>(d_terr_all->p) + q; }
    // low-rank route: one failure flag per gate for the Cholesky factorisation of B^dagger B
    Buf d_lowfail = sub_buffer(d_rb, rb_low, std::max<size_t>(1, (size_t)npg * sizeof(int)));
    Buf d_texp = dalloc(s, std::max<size_t>(1, (size_t)npg * sizeof(int)));
    Buf d_lowfail2 = sub_buffer(d_rb, rb_low2, std::max<size_t>(1, (size_t)npg * sizeof(int)));      // ComplexF64: second CholeskyQR pass of the low-rank route
    for (int q = 0; q < npg; ++q) { gitems[q].lowfail = reinterpret_cast<const int*>(d_lowfail->p) + q; gitems[q].texp = reinterpret_cast<int*>(d_texp->p) + q; }
    const GateItem* d_gitems = upload(s, gitems);       // (read again after the host synchronisations of the batch: a device copy, not upload_small)
    auto run_theta = [&]() {
        HIPCHK(hipMemsetAsync(d_rb->p, 0, rb_terr, s->stream));       // info and both low-rank failure flag arrays (contiguous)
        ProfScope ps(s, TNQS_PROF_SMALL, 0, 0);
        launch_gate_theta<T>(s->stream, d_gitems, npg);
        if (lowrank_on_batch) launch_gate_theta_mm<T>(s->stream, d_gitems, npg);        // theta = A B^T on the f64 matrix cores (gates with operator-sum factors)
        const bool f64 = !std::is_same<T, float>::value;
        std::vector<CholItem> lc, lc2; int kmax = 1;
        std::vector<GateItem> g2, g3; std::vector<LowQr2Item> qi;
        for (int q = 0; q < npg; ++q) {
            if (!gitems[q].lowG) continue;
            const int K = gitems[q].kappa * gitems[q].chi; GateWS& w = ws[pg[q]];
            int* fail1 = reinterpret_cast<int*>(d_lowfail->p) + q;
            // ComplexF32: tau at the f32 noise floor.  ComplexF64: 1e-12 on the pivots of pass 1 keeps kappa(B) <= 1e6, where the second pass restores
            // orthogonality to eps; anything worse falls back to the SVD of the full theta
            lc.push_back(CholItem{gitems[q].lowG, const_cast<void*>(gitems[q].lowL), (K <= 96 || f64) ? w.lowW->p : nullptr, K, fail1, f64 ? 1e-12 : rank_tau(true, K)});
            kmax = std::max(kmax, K);
            if (f64) {
                int* fail2 = reinterpret_cast<int*>(d_lowfail2->p) + q;
                lc2.push_back(CholItem{w.lowG2->p, w.lowL2->p, nullptr, K, fail2, 1e-3});      // G2 is the identity up to kappa(G1) eps: a pivot below 1e-3 means pass 1 was not good enough
                GateItem a = gitems[q]; a.lowB = w.lowB1->p; a.lowG = w.lowG2->p; a.lowL = w.lowL2->p; g2.push_back(a);
                GateItem b = gitems[q]; b.lowL = w.lowLc->p; g3.push_back(b);
                qi.push_back(LowQr2Item{w.lowB->p, w.lowW->p, w.lowB1->p, gitems[q].lowL, w.lowL2->p, w.lowLc->p, gitems[q].info, gitems[q].d2, fail1, fail2});
            }
        }
        if (!lc.empty()) {
            const CholItem* dc = upload_small(s, lc); launch_lowrank_g(s->stream, d_gitems, npg);
            if (kmax <= 96) launch_chol(s->stream, dc, (int)lc.size(), kmax); else launch_chol_packed(s->stream, dc, (int)lc.size(), kmax);      // ComplexF32: only L is used here
            if (!f64) launch_lowrank_m<T>(s->stream, d_gitems, npg);
            else {
                const LowQr2Item* dq = upload_small(s, qi); const GateItem* d2 = upload_small(s, g2); const GateItem* d3 = upload_small(s, g3); const CholItem* dc2 = upload_small(s, lc2);
                launch_lowrank_bw(s->stream, dq, (int)qi.size());                       // B1 = B L1^-dagger
                launch_lowrank_g(s->stream, d2, (int)g2.size());                        // G2 = B1^dagger B1
                if (kmax <= 96) launch_chol(s->stream, dc2, (int)lc2.size(), kmax); else launch_chol_packed(s->stream, dc2, (int)lc2.size(), kmax);
                launch_lowrank_ll(s->stream, dq, (int)qi.size());                       // Lc = L1 L2 (+ pass-2 failure -> the gate's flag)
                launch_lowrank_m<T>(s->stream, d3, (int)g3.size());                     // theta[:, :K] = A conj(Lc)
            }
        }
        launch_theta_scale<T>(s->stream, d_gitems, npg);       // theta (or M) and theta0 to O(1), exponent kept per gate for gate_finish
    };
    if (ev_small) { HIPCHK(hipStreamWaitEvent(s->stream, ev_small, 0)); ev_small = nullptr; }      // the early small-SVD factors (2b) are inputs of gate_theta
    run_theta();
    std::vector<int> info(8 * (size_t)ng, 0); std::vector<double> terr(ng, 0.0);
    {
        std::vector<int> hinfo(8 * (size_t)std::max(1, npg));
        std::vector<double> hterr(std::max(1, npg));
        // SVD of theta (rotated in place to U Sigma), recovery of V from the unrotated copy, truncation and X1 / X2.  `dims` = the ranks
        // read back from the device, or null: the kernels read them from the gates' info arrays themselves (JacobiItem::dyn) and the
        // host sizes the launches with upper bounds
        auto svd_and_finish = [&](const int* dims) {
            std::vector<JacobiItem> ji; std::vector<int> ncfull;
            for (int q = 0; q < npg; ++q) {
                int gi = pg[q];
                JacobiItem j{};
                if (dims) {
                    int Mr, Nc, ncolJ; theta_dims(dims + 8 * q, gitems[q].d1, gitems[q].d2, Mr, Nc, ncolJ);
                    j = JacobiItem{ws[gi].theta->p, ws[gi].thetaV->p, Mr, ncolJ, gitems[q].info + 4};
                    ncfull.push_back(Nc);
                } else {
                    const int Mr = std::max(ws[gi].n1 * gitems[q].d1, ws[gi].n2 * gitems[q].d2), Nc = std::min(ws[gi].n1 * gitems[q].d1, ws[gi].n2 * gitems[q].d2);
                    j = JacobiItem{ws[gi].theta->p, ws[gi].thetaV->p, Mr, Nc, gitems[q].info + 4, gitems[q].info, gitems[q].d1, gitems[q].d2,
                                   gitems[q].lowG ? gitems[q].kappa * gitems[q].chi : 0};      // low-rank route expected: K columns
                    // offered to theta_svd_pre_kernel, which decides on the device: the low-rank factor with its Q (2), or theta as it stands when the ranks the
                    // sites CAN have (a site with fewer fibers than columns: a corner) keep it within 128 x 64 (1) -- those gates took 12-13 plain sweeps on 128 rows
                    // and were what a colour batch of a small lattice waited for
                    if (std::is_same<T, float>::value && use_precond_svd()) {
                        auto rub = [&](size_t i) { const int n = nof(i); return (int)std::min<size_t>((size_t)n, sj[i].sd.n / (size_t)n); };
                        const int a1 = rub(2 * (size_t)gi) * gitems[q].d1, a2 = rub(2 * (size_t)gi + 1) * gitems[q].d2;
                        if (gitems[q].lowQ) { j.QB = gitems[q].lowQ; j.Vout = ws[gi].thetaV->p; j.pre = 2; }
                        else if (theta_svd_pre_covers(std::max(a1, a2), std::min(a1, a2))) { j.Vout = ws[gi].thetaV->p; j.pre = 1; }
                        j.cap = ws[gi].cap;
                    }
                    ncfull.push_back(Nc);
                }
                j.V = nullptr;          // V is never accumulated from the rotations (theta0_used): recovered below
                ji.push_back(j);
            }
            { ProfScope ps(s, TNQS_PROF_JACOBI, 0, 0); svd_batch<T>(s, ji, false); }
            std::vector<RecoverItem> rv;
            for (int q = 0; q < npg; ++q) rv.push_back(RecoverItem{ws[pg[q]].theta0->p, ws[pg[q]].theta->p, ws[pg[q]].thetaV->p, ji[q].m, ncfull[q], ji[q].n, ji[q].dyn, ji[q].dm, ji[q].dn, ji[q].pre});
            const RecoverItem* dr = upload_small(s, rv);
            { ProfScope ps(s, TNQS_PROF_JACOBI, 0, 0); { int nmax = 1; for (int nc : ncfull) nmax = std::max(nmax, nc); if (std::is_same<T, float>::value && use_mfma()) launch_recover_v_mfma(s->stream, dr, npg, nmax); else launch_recover_v<T>(s->stream, dr, npg, nmax); } }
            { ProfScope ps(s, TNQS_PROF_SMALL, 0, 0); launch_gate_finish<T>(s->stream, d_gitems, npg); }
        };
        const char* st_all = nullptr;        // the staged copy of d_rb
        auto read_results = [&]() {          // (r1, r2, chi', status, ...), the truncation errors of every gate and every flag of the batch: one copy, one synchronisation
            st_all = readback<char>(s, d_rb->p, rb_total);
            HIPCHK(hipStreamSynchronize(s->stream)); drained(s);
            if (npg) { const int* a = reinterpret_cast<const int*>(st_all + rb_info); const double* b = reinterpret_cast<const double*>(st_all + rb_terr);
                       std::copy(a, a + (size_t)npg * 8, hinfo.begin()); std::copy(b, b + npg, hterr.begin()); }
        };
        auto take_flags = [&]() {
            const int* f = reinterpret_cast<const int*>(st_all + rb_env); const int* c = reinterpret_cast<const int*>(st_all + rb_chol);
            if (!envs.empty()) std::copy(f, f + 2 * envs.size(), h_flags.begin());
            if (!sj.empty()) std::copy(c, c + sj.size(), h_cholfail.begin());
        };
        auto chol_failures = [&]() { int c = 0; for (size_t i = 0; i < sj.size(); ++i) c += (part[i / 2] && is_chol[i] && h_cholfail[i]) ? 1 : 0; return c; };
        auto redo_with_eigen = [&]() {      // numerically rank-deficient Gram matrix somewhere in the batch: those sites take the eigen path
            factor_G(false, true);
            for (int q = 0; q < npg; ++q) { int gi = pg[q]; GateItem& it = gitems[q]; it.GW1 = GW[2 * gi]->p; it.GW2 = GW[2 * gi + 1]->p; it.chol1 = is_chol[2 * gi]; it.chol2 = is_chol[2 * gi + 1]; }
            d_gitems = upload(s, gitems);
            run_theta();
            if (npg) HIPCHK(hipMemcpyAsync(hinfo.data(), d_info_all->p, (size_t)npg * 32, hipMemcpyDeviceToHost, s->stream));
            HIPCHK(hipStreamSynchronize(s->stream)); drained(s);
            s->stats.n_chol_fallbacks += 1;
        };
        // ONE host round trip per batch where the whole chain can be sized from upper bounds: ComplexF32 (no second factorisation pass), every
        // theta small enough for the LDS-resident Jacobi at its largest possible size.  The ranks of the R factors stay on the device; the
        // Cholesky failure flags and the message-eigenvalue flags are read together with the results, and a failure (rare) redoes the chain
        bool one_trip = (!qr2 || spec) && npg > 0;      // (ComplexF64 on assumptions: no site flagged for the second factorisation pass -- part of the check)
        for (int q = 0; q < npg && one_trip; ++q) {
            const int gi = pg[q];
            const int Mr = std::max(ws[gi].n1 * gitems[q].d1, ws[gi].n2 * gitems[q].d2), Nc = std::min(ws[gi].n1 * gitems[q].d1, ws[gi].n2 * gitems[q].d2);
            one_trip = jacobi_lds(jacobi_lds_bytes(Mr, Nc, false, esz)) > 0 && Mr <= 256;
        }
        // the four staged read-backs of a batch (flags, Cholesky flags, info, truncation errors) are consumed together after ONE synchronisation:
        // room for all of them is made up front, so that none of them can wrap the arena on top of another (round-3 advisor finding)
        const size_t rb_bytes = rb_total + 1024;
        char* spec_stage = (spec && one_trip) ? ring_alloc(s, rb_total) : nullptr;      // (may settle -- and throw -- first: nothing of the state has been touched)
        spec = spec && one_trip && spec_stage;
        if (one_trip) {
            svd_and_finish(nullptr);
            if (!sharded && ao.maxdim > 0) spec_plan = plan_rowgemm([&](int gi) { return ws[gi].cap; }, [&](size_t q) { return (const void*)s->site[sj[own_idx[q]].v]->p; }, nullptr);
            ht_a.stop(); ht_s4.stop();
            if (spec) {
                // no host round trip: the results travel to the check's staging and are verified when they have arrived; the rest of the batch runs on what they are
                // expected to be -- every factor of full rank, every new bond dimension at its cap, no fallback taken
                HIPCHK(hipMemcpyAsync(spec_stage, d_rb->p, rb_total, hipMemcpyDeviceToHost, s->stream));
                Check c; c.kind = 0; c.step = s->cur_step; c.ev = check_event(s);
                HIPCHK(hipEventRecord(c.ev, s->stream));
                struct PerGate { int index, cap, d1, d2, K, chi_cap; bool low; };
                std::vector<PerGate> pgv(npg);
                for (int q = 0; q < npg; ++q) pgv[q] = PerGate{gates[pg[q]].index, ws[pg[q]].cap, gitems[q].d1, gitems[q].d2, gitems[q].kappa * gitems[q].chi, gitems[q].chi_cap, gitems[q].lowG != nullptr};
                std::vector<char> chol_site(sj.size(), 0);
                for (size_t i = 0; i < sj.size(); ++i) chol_site[i] = (part[i / 2] && is_chol[i]) ? 1 : 0;
                const size_t nenv = envs.size();
                const char* st = spec_stage;
                c.eval = [st, rb_info, rb_terr, rb_chol, rb_env, pgv, chol_site, nenv, errs, qr2](State* z) {
                    const int* hi = reinterpret_cast<const int*>(st + rb_info); const double* ht = reinterpret_cast<const double*>(st + rb_terr);
                    const int* fl = reinterpret_cast<const int*>(st + rb_env); const int* cf = reinterpret_cast<const int*>(st + rb_chol);
                    for (size_t q = 0; q < pgv.size(); ++q) if (hi[8 * q + 2] != pgv[q].cap || hi[8 * q + 3] != 0 || (qr2 && hi[8 * q + 6] != 0)) return false;      // a bond below its cap, a failed gate, an ill-conditioned ComplexF64 site (second factorisation pass)
                    for (size_t i = 0; i < nenv; ++i) if (!fl[2 * i] || fl[2 * i + 1]) return false;                                  // a rank-deficient message (projector pass), or a negative eigenvalue
                    for (size_t i = 0; i < chol_site.size(); ++i) if (chol_site[i] && cf[i]) return false;                           // a collapsed Cholesky pivot
                    for (size_t q = 0; q < pgv.size(); ++q) {       // the assumptions held: book what the careful route books after its read-back
                        int Mr, Nc, ncolJ; theta_dims(hi + 8 * q, pgv[q].d1, pgv[q].d2, Mr, Nc, ncolJ);
                        z->stats.n_lowrank_svd += (ncolJ < Nc) ? 1 : 0; z->stats.n_svd_sweeps += hi[8 * q + 4]; z->stats.n_svd_sweeps_max = std::max(z->stats.n_svd_sweeps_max, hi[8 * q + 4]);
                        const int r1d = hi[8 * q] * pgv[q].d1, r2d = hi[8 * q + 1] * pgv[q].d2;
                        if (pgv[q].low && ncolJ == Nc && r1d >= r2d && pgv[q].K < r2d && pgv[q].chi_cap <= pgv[q].K) z->stats.n_lowrank_fallbacks += 1;
                        if (errs) errs[pgv[q].index] = ht[q];
                    }
                    return true;
                };
                s->checks.push_back(std::move(c));
                for (int q = 0; q < npg; ++q) { for (int k = 0; k < 8; ++k) hinfo[8 * q + k] = 0; hinfo[8 * q + 2] = ws[pg[q]].cap; hterr[q] = 0.0; }
                for (size_t i = 0; i < envs.size(); ++i) { h_flags[2 * i] = 1; h_flags[2 * i + 1] = 0; }
                s->stats.n_spec_batches += 1;
            } else {
            reserve_readback(s, rb_bytes);
            read_results(); take_flags();
            settle(s, true);              // (the stream is drained: what was pending has fired; a failed check unwinds this batch before it has replaced anything)
            if (chol_failures()) { redo_with_eigen(); svd_and_finish(hinfo.data()); read_results(); }
            }
        } else {
            // theta dims depend on the ranks found on the device: read them back (also where message-eigenvalue errors surface)
            reserve_readback(s, rb_bytes);
            ht_a.stop(); ht_s4.stop();
            read_results(); take_flags();
            settle(s, true);
            if (chol_failures()) redo_with_eigen();
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
#ifdef TNQS_EXPERIMENTS
                    static const bool all = [] { const char* v = std::getenv("TNQS_QR2_ALL"); return v && v[0] == '1'; }();      // refine every site
#else
                    const bool all = false;
#endif
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
                            if (cross[i]) dst = reinterpret_cast<char*>(s->exch) + (size_t)s->rank * stride + slot[i];
                            else { G2[k] = dalloc(s, (size_t)nn * 16); dst = G2[k]->p; }
                            rd.push_back(ReduceItem{gj[t].partial->p, dst, nn, gj[t].nchunks, 1, elems}); elems += nn;
                        }
                        const ReduceItem* d2 = upload(s, rd); ProfScope ps(s, TNQS_PROF_SMALL, 0, 0); launch_reduce<double, double>(s->stream, d2, (int)rd.size(), elems);
                    }
                    if (sharded && stride) {
                        exchange(s, stride);
                        Buf G2_keep = dalloc(s, std::max<size_t>(256, stride * (size_t)s->nranks));
                        HIPCHK(hipMemcpyAsync(G2_keep->p, s->exch, stride * (size_t)s->nranks, hipMemcpyDeviceToDevice, s->stream));
                        for (size_t k = 0; k < m; ++k) { const size_t i = rs[k]; if (!cross[i]) continue; const size_t nn = (size_t)nof(i) * nof(i); G2[k] = sub_buffer(G2_keep, (size_t)s->owner[sj[i].v] * stride + slot[i], nn * 16); }
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
                        HIPCHK(hipStreamSynchronize(s->stream)); drained(s);
                        for (size_t k = 0; k < m; ++k) s->stats.n_qr2_sites += sj[rs[k]].owned ? 1 : 0;
                    }
                }
            }
            svd_and_finish(hinfo.data());
            read_results();
        }
        HostTimer ht_b(4);
        for (size_t i = 0; i < envs.size(); ++i)
            if (h_flags[2 * i + 1]) throw Err(TNQS_ERR_NUMERIC, "simple_update: incoming message has a negative eigenvalue above sqrt_cutoff (DomainError in the reference, src/utils.jl:21)");
        for (int q = 0; q < npg; ++q) {
            if (spec) { for (int k = 0; k < 8; ++k) info[8 * pg[q] + k] = hinfo[8 * q + k]; terr[pg[q]] = 0.0; continue; }      // (booked by the check)
            int Mr, Nc, ncolJ; theta_dims(hinfo.data() + 8 * q, gitems[q].d1, gitems[q].d2, Mr, Nc, ncolJ);
            s->stats.n_lowrank_svd += (ncolJ < Nc) ? 1 : 0; s->stats.n_svd_sweeps += hinfo[8 * q + 4]; s->stats.n_svd_sweeps_max = std::max(s->stats.n_svd_sweeps_max, hinfo[8 * q + 4]);
            { static const bool dbg = envflag("TNQS_DEBUG_SWEEPS");      // diagnostics: which thetas the SVD launch of a batch waits for
              if (dbg) std::fprintf(stderr, "[tnqs sweeps] gate %d: r1 %d r2 %d theta %d x %d, SVD on %d columns, %d sweeps, chi' %d\n", pg[q], hinfo[8 * q], hinfo[8 * q + 1], Mr, Nc, ncolJ, hinfo[8 * q + 4], hinfo[8 * q + 2]); }
            {   // qualified for the low-rank route by its ranks, but gate_theta's offer was withdrawn on the device (lowrank_m: a refused pivot)
                const int K = gitems[q].kappa * gitems[q].chi, r1d = hinfo[8 * q] * gitems[q].d1, r2d = hinfo[8 * q + 1] * gitems[q].d2;
                if (gitems[q].lowG && ncolJ == Nc && r1d >= r2d && K < r2d && gitems[q].chi_cap <= K) s->stats.n_lowrank_fallbacks += 1;
            }
            for (int k = 0; k < 8; ++k) info[8 * pg[q] + k] = hinfo[8 * q + k]; terr[pg[q]] = hterr[q];
        }
    }
    // ---- 4b. sharded: the owner of the first vertex publishes (chi', status, truncerr, S, X2) of each gate ----------------
    std::vector<const double*> Sptr(ng, nullptr);
    Buf S_keep;
    if (sharded) {
        // every rank needs (chi', status, truncerr, S) of every gate (bond dimensions and messages are replicated); X2 only travels for a gate that straddles two ranks
        std::vector<size_t> slot(ng, 0); std::vector<size_t> rank_bytes(s->nranks, 0);
        for (int gi = 0; gi < ng; ++gi) {
            int r = s->owner[gates[gi].v1]; slot[gi] = rank_bytes[r];
            rank_bytes[r] += round256(32 + (size_t)cap_max * 8 + (s->owner[gates[gi].v1] != s->owner[gates[gi].v2] ? x2_max : 0));
        }
        size_t stride = 0; for (size_t b : rank_bytes) stride = std::max(stride, b);
        check_exchange(s, stride);
        char* base = reinterpret_cast<char*>(s->exch);
        {   // pack the records of the gates whose first vertex is ours (one launch)
            std::vector<RecordPackItem> rp;
            std::vector<int> qof(ng, -1); for (int q = 0; q < npg; ++q) qof[pg[q]] = q;
            for (int gi = 0; gi < ng; ++gi) {
                if (s->owner[gates[gi].v1] != s->rank) continue;
                const SiteJob& b = sj[2 * gi + 1]; const int q = qof[gi];
                const bool straddles = s->owner[gates[gi].v1] != s->owner[gates[gi].v2];
                rp.push_back(RecordPackItem{base + (size_t)s->rank * stride + slot[gi], gitems[q].info, gitems[q].truncerr, reinterpret_cast<const double*>(ws[gi].S->p),
                                            ws[gi].cap, ws[gi].X2->p, straddles ? (long long)((size_t)ws[gi].n2 * b.sd.d * ws[gi].cap * esz / 8) : 0LL, (long long)(32 + (size_t)cap_max * 8)});
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
            HIPCHK(hipStreamSynchronize(s->stream)); drained(s);
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
    HostTimer ht_c(5);
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
        const bool f64mf = std::is_same<T, double>::value && use_mfma() && KKmax >= 4 && KKmax <= 64 && NNmax <= 64;      // kernels_f64.hip
        std::vector<char> via64(own_idx.size(), 0);      // sites served by the register-direct matrix-core kernel
        {   // chi = 64 sites (K = (s, b) = 128 -> N = (s', b') <= 128) and chi = 32 sites on the register-direct MFMA kernel: the speculative plan built
            // before the read-back when it came true, a fresh one otherwise
            bool spec_ok = spec_plan.valid;
            for (size_t q = 0; q < own_idx.size() && spec_ok; ++q) { const int gi = (int)own_idx[q] / 2; spec_ok = info[8 * gi + 2] == ws[gi].cap && pch[q].steps.empty() && pch[q].result == s->site[sj[own_idx[q]].v]->p; }
            RgPlan fresh_plan;
            if (!spec_ok) {
                spec_plan = RgPlan{};              // (its output buffers go back to the pool)
                fresh_plan = plan_rowgemm([&](int gi) { return info[8 * gi + 2]; }, [&](size_t q) { return pch[q].result; }, &via64);
            } else for (size_t q = 0; q < own_idx.size(); ++q) via64[q] = via64[q] || spec_plan.via[q];
            RgPlan& P = spec_ok ? spec_plan : fresh_plan;
            bool booked = false;
            for (auto& G : P.groups) {          // one launch per contracted dimension
                { ProfScope ps(s, TNQS_PROF_GATE_APPLY, booked ? 0.0 : P.rby, booked ? 0.0 : P.rfl); booked = true; launch_mfma_rowgemm(s->stream, G.d, (int)G.sub.size(), G.wgs, 2, G.kk, reinterpret_cast<double*>(G.npr->p)); }
                norm_and_replace<T>(s, G.sv, G.so, G.sn, G.npr, G.stb, G.snt, ao.normalize_tensors != 0);
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
            if (f64mf) { fiber_gemm_f64_tiles(it); it.tpw = 32; }      // ComplexF64 epilogue on the f64 matrix cores: tiles of 16 fibers, 32 per workgroup
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
          if (f64mf) launch_mfma_fiber_gemm_f64(s->stream, d, (int)items.size(), tiles, (int)KKmax, (int)NNmax, reinterpret_cast<double*>(np->p), true);
          else if (mf) launch_mfma_fiber_gemm(s->stream, d, (int)items.size(), tiles, (int)KKmax, (int)NNmax, reinterpret_cast<double*>(np->p));
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
            if (errs && !spec) errs[gates[gi].index] = terr[gi];      // (spec: written by the check, from the staged truncation errors)
        }
        const DiagItem* d = upload_small(s, di);
        { ProfScope ps(s, TNQS_PROF_SMALL, 0, 0); launch_diag<T>(s->stream, d, (int)di.size()); }
    }
    for (auto& g2 : gates) { s->pend1[g2.v1].clear(); s->pend1[g2.v2].clear(); s->unit_norm[g2.v1] = s->unit_norm[g2.v2] = ao.normalize_tensors ? 1 : 0; }
    s->stats.n_two_site += ng;
    soft_sync(s);   // workspace of this batch goes back to the pool at the next stream synchronisation (the BP update's first read-back)
}

// ---------------------------------------------------------------------------------------------------------------
// apply_gates (src/Apply/apply_gates.jl:46-98)
// ---------------------------------------------------------------------------------------------------------------
// The walk over the gate list -- which gates form a batch, where a BP update is due -- depends on the gate list alone (apply_gates.jl:64-90: vertex sets), not on
// any number computed on the way.  So the SCHEDULE is built first: steps = maximal runs of pairwise-disjoint gates ("batch") and the cache updates between them
// ("bp"), in the reference's order.  It is then executed AHEAD of the device (engine.hpp: Check): a batch whose outcome is predictable and an update expected to
// converge in one sweep are enqueued without waiting for their results, the state in front of every step is remembered as a vector of references, and a step
// whose deferred verification fails is run again the careful way from that state.  Results are those of the sequential walk either way.
struct Step { bool is_bp = false; std::vector<Gate1> b1; std::vector<Gate2> b2; };

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
    // ---- the schedule (vertex flags instead of std::set: this walk sits in front of the first kernel of a call) ----------------------------------------
    struct VSet { std::vector<char> f; std::vector<int> l; explicit VSet(int n) : f(n, 0) {} bool count(int v) const { return f[v] != 0; }
                  void insert(int v) { if (!f[v]) { f[v] = 1; l.push_back(v); } } void clear() { for (int v : l) f[v] = 0; l.clear(); } };
    std::vector<Step> steps;
    {
        VSet affected(g.nv), batch_verts(g.nv);
        Step cur;
        auto flush = [&]() { if (!cur.b1.empty() || !cur.b2.empty()) { steps.push_back(std::move(cur)); cur = Step{}; } };
        for (int i = 0; i < ngates; ++i) {
            const int nv = nverts[i]; const int32_t* vs = verts + voff[i];
            bool need = false;
            if (nv >= 2) for (int k = 0; k < nv; ++k) need = need || affected.count(vs[k]);            // apply_gates.jl:68
            if (ao.update_cache && need) {
                flush(); batch_verts.clear();
                Step u; u.is_bp = true; steps.push_back(std::move(u));                                 // :76
                affected.clear();                                                                      // :78
            }
            bool overlap = false;
            for (int k = 0; k < nv; ++k) overlap = overlap || batch_verts.count(vs[k]);
            if (overlap) { flush(); batch_verts.clear(); }
            if (nv == 1) cur.b1.push_back(Gate1{vs[0], mats + moff[i]}); else cur.b2.push_back(Gate2{vs[0], vs[1], mats + moff[i], i});
            for (int k = 0; k < nv; ++k) { batch_verts.insert(vs[k]); affected.insert(vs[k]); }         // :88-90
        }
        flush();
        if (ao.update_cache) { Step u; u.is_bp = true; steps.push_back(std::move(u)); }                 // :93-95
    }
    struct InApply { State* s; explicit InApply(State* st) : s(st) { s->in_apply = true; } ~InApply() { s->in_apply = false; s->cur_step = -1; } } in_apply_guard(s);
    // ---- how far ahead?  Running ahead keeps the state in front of every unverified step alive (the site tensors a batch replaced): up to a layer's worth of
    // extra copies.  Handles with more than 2 GiB of site tensors stay one step deep -- the round-5 flow: a batch reads its results back, an update
    // leaves its verdict pending until the next batch has prepared itself --, and so does a handle on which a verification failed a moment ago
    // (Graph::spec_penalty, shared by the copies of a handle: an evolution whose updates need several sweeps would throw away a batch per update otherwise).
    size_t own_bytes = 0; for (auto& b : s->site) if (b) own_bytes += b->bytes;
    size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) != hipSuccess) tot = 0;
    static const bool spec_off = envflag("TNQS_NO_SPECULATION");
    (void)fr; (void)tot;
    // (measured, round 6: heavy-hex 3.0 -> 2.6 ms per layer, 7 x 7 unchanged -- its launch chain is 97 % busy either way --, 20 x 20 112.6 against 111.5 ms and 37 against
    //  22 GiB at the peak: where the tensor passes fill the device there is no idle time to win, only memory to lose.  Hence the bound: 2 GiB of site tensors)
    const bool deep = !spec_off && !s->sharded() && own_bytes <= (size_t(2) << 30);
    const size_t nst = steps.size();
    std::vector<std::unique_ptr<Snapshot>> snaps(nst + 1);
    size_t k = 0; bool careful = false;
    auto drop_old_snaps = [&]() { const size_t keep_from = s->checks.empty() ? k : (size_t)std::max(0, s->checks.front().step); for (size_t q = 0; q < keep_from && q < snaps.size(); ++q) snaps[q].reset(); };
    // after SpecFailed: nothing enqueued behind the failed step may leave a trace -- drain, drop the younger checks, put the state back
    auto recover = [&](const SpecFailed& f) {
        HIPCHK(hipStreamSynchronize(s->stream)); if (s->aux_stream) HIPCHK(hipStreamSynchronize(s->aux_stream));
        drained(s); drop_checks(s);
        const tnqs_apply_stats st_now = s->stats;
        g.spec_penalty = 12;
        if (f.kind == 0) {                          // a gate batch: back to the state in front of it, run it the careful way
            restore_snapshot(s, *snaps[f.step]);
            k = (size_t)f.step; careful = true;
        } else {                                    // a BP update whose first sweep missed the tolerance: the state right behind that sweep, then the remaining sweeps
            if ((size_t)f.step + 1 < snaps.size() && snaps[f.step + 1]) restore_snapshot(s, *snaps[f.step + 1]);
            s->cur_step = f.step;
            bp_update_t<T>(s, bp, nullptr, nullptr, false, f.iters_done);
            k = (size_t)f.step + 1; careful = false;
        }
        s->stats.n_spec_redone = st_now.n_spec_redone + 1;
        for (size_t q = k + 1; q < snaps.size(); ++q) snaps[q].reset();
    };
    try {
        while (k < nst || !s->checks.empty()) {
            try {
                if (k >= nst) { settle(s, true); break; }
                Step& st = steps[k];
                s->cur_step = (int)k;
                const bool ahead = deep && g.spec_penalty == 0 && !careful;
                if (deep || !s->checks.empty()) snaps[k] = std::make_unique<Snapshot>(take_snapshot(s));
                if (st.is_bp) {
                    bp_update_t<T>(s, bp, nullptr, nullptr, /*optimistic=*/true);
                } else {
                    apply_one_site_batch<T>(s, st.b1, ao.normalize_tensors != 0, false);
                    apply_two_site_batch<T>(s, st.b2, ao, errs, /*allow_spec=*/ahead);
                    s->stats.n_batches += 1;
                    soft_sync(s);
                }
                careful = false; ++k;
                if (g.spec_penalty > 0 && s->checks.empty()) g.spec_penalty -= 1;
                if (!ahead && s->checks.size() > 1) settle(s, true);        // one step deep: at most the verdict of the update just enqueued stays pending
                else if (s->checks.size() >= 12) settle(s, true);
                else settle(s, false);
                drop_old_snaps();
            } catch (const SpecFailed& f) { recover(f); }
        }
    } catch (...) {
        // an error of a step: what is still unverified is settled -- or rolled back to the last verified state -- before the handle is handed back; the first error wins
        try { settle(s, true); }
        catch (const SpecFailed& f) { (void)hipStreamSynchronize(s->stream); drop_checks(s); if (f.step >= 0 && (size_t)f.step < snaps.size() && snaps[f.step + (f.kind ? 1 : 0)]) restore_snapshot(s, *snaps[f.step + (f.kind ? 1 : 0)]); }
        catch (...) { drop_checks(s); }
        throw;
    }
    if (s->sharded()) materialize_pending_all(s);      // sharded: nothing stays pending between calls (State::in_apply)
    sync(s);                                                                                        // the call returns with the stream drained
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
    sync(s);
}
void truncate_bp(State* s, int maxdim, double cutoff, int normalize, int ngroups, const int32_t* offs,
                 const int32_t* eu, const int32_t* ev, const tnqs_bp_opts* bp) {
    s->stats = tnqs_apply_stats{};
    if (s->dtype == TNQS_C64) truncate_t<float>(s, maxdim, cutoff, normalize, ngroups, offs, eu, ev, bp);
    else truncate_t<double>(s, maxdim, cutoff, normalize, ngroups, offs, eu, ev, bp);
}

}  // namespace tnqs
