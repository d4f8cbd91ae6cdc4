// engine_obs.cpp -- observables (src/expect.jl), BP scalars and rescale!, symmetric gauge (src/symmetric_gauge.jl).
#include "engine_internal.hpp"

namespace tnqs {

// ---------------------------------------------------------------------------------------------------------------
// parity probes (src/expect.jl:59-82)
// ---------------------------------------------------------------------------------------------------------------
template <class T> static void rdm_batch(State* s, const std::vector<int>& vs, double* out /* per vertex d*d complex128, packed */) {
    const Graph& g = *s->g;
    HIPCHK(hipSetDevice(s->device));
    materialize_pending(s, vs);
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

// rescale_messages!(bpc, edges) (beliefpropagationcache.jl:127-140): both directions of every listed edge (replicated on every rank when
// sharded); edges == nullptr: all of them
template <class T> static void rescale_messages_t(State* s, int n, const int32_t* eu, const int32_t* ev) {
    const Graph& g = *s->g;
    const size_t esz = s->esz();
    HIPCHK(hipSetDevice(s->device));
    std::vector<int> es;
    if (!eu || !ev) { es.resize(g.ne); std::iota(es.begin(), es.end(), 0); }
    else for (int i = 0; i < n; ++i) {
        const int e = g.edge(eu[i], ev[i]);
        if (e < 0) throw Err(TNQS_ERR_INVALID, "rescale_messages: not an edge");
        if (std::find(es.begin(), es.end(), e) == es.end()) es.push_back(e);      // (u, v) and (v, u) name the same pair of messages
    }
    if (es.empty()) return;
    std::vector<MsgRescaleItem> items; std::vector<Buf> na(es.size()), nb(es.size());
    for (size_t q = 0; q < es.size(); ++q) {
        const int e = es[q];
        size_t bytes = (size_t)s->chi[e] * s->chi[e] * esz;
        na[q] = dalloc(s, bytes); nb[q] = dalloc(s, bytes);
        items.push_back(MsgRescaleItem{s->msg[2 * e] ? s->msg[2 * e]->p : nullptr, s->msg[2 * e + 1] ? s->msg[2 * e + 1]->p : nullptr, na[q]->p, nb[q]->p, s->chi[e]});
    }
    const MsgRescaleItem* d = upload(s, items);
    launch_msg_rescale<T>(s->stream, d, (int)items.size());
    for (size_t q = 0; q < es.size(); ++q) { const int e = es[q]; s->keepalive.push_back(s->msg[2 * e]); s->keepalive.push_back(s->msg[2 * e + 1]); s->msg[2 * e] = na[q]; s->msg[2 * e + 1] = nb[q]; }
    sync(s);
}
// rescale_vertices!(bpc, vertices) (beliefpropagationcache.jl:82-101): psi_v *= sign(vn) / sqrt(vn), vn = vertex_scalar under the CURRENT
// messages; vertices == nullptr: all of them (a sharded rank rescales the ones it owns)
template <class T> static void rescale_vertices_t(State* s, int n, const int32_t* verts) {
    const Graph& g = *s->g;
    const size_t esz = s->esz();
    HIPCHK(hipSetDevice(s->device));
    std::vector<char> want(g.nv, verts ? 0 : 1);
    if (verts) for (int i = 0; i < n; ++i) { if (verts[i] < 0 || verts[i] >= g.nv) throw Err(TNQS_ERR_INVALID, "rescale_vertices: bad vertex"); want[verts[i]] = 1; }
    std::vector<double> vn(2 * (size_t)g.nv);
    vertex_scalars(s, vn.data());
    materialize_scale_all(s);
    std::vector<CScaleItem> cs; std::vector<Buf> outs; std::vector<int> vs;
    for (int v = 0; v < g.nv; ++v) {
        if (!want[v] || !s->owns(v) || !s->site[v]) continue;
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
        for (size_t i = 0; i < vs.size(); ++i) { s->keepalive.push_back(s->site[vs[i]]); s->site[vs[i]] = outs[i]; s->unit_norm[vs[i]] = 0; }
    }
    sync(s);
}
void rescale_messages(State* s, int n, const int32_t* eu, const int32_t* ev) { if (s->dtype == TNQS_C64) rescale_messages_t<float>(s, n, eu, ev); else rescale_messages_t<double>(s, n, eu, ev); }
void rescale_vertices(State* s, int n, const int32_t* verts) { if (s->dtype == TNQS_C64) rescale_vertices_t<float>(s, n, verts); else rescale_vertices_t<double>(s, n, verts); }
// rescale! = rescale_messages! then rescale_vertices! (abstract...:318-322)
void rescale(State* s) { rescale_messages(s, 0, nullptr, nullptr); rescale_vertices(s, 0, nullptr); }


// ---------------------------------------------------------------------------------------------------------------
// multi-site expectation value on a tree-shaped region (src/expect.jl:59-82): the norm network of the region's
// vertices with the cache's messages on the boundary edges and operators inserted, numerator (ops) over denominator (identities).
// The region is contracted leaves-to-root with the message kernels: m_{u->parent} = sum (O_u psi_u) conj(psi_u) prod(incoming).
// ---------------------------------------------------------------------------------------------------------------
// Bonds BETWEEN region vertices that are not tree edges (the induced region has a loop: a plaquette, say) are part of the norm network too
// (norm_factors over steiner_vs, src/expect.jl:72: every internal bond is shared).  Such a bond is cut: with its ket index fixed to i and its
// bra index to j the region is the tree again, and the value is the sum over all (i, j).  Fixing the indices costs no new kernel: at both
// ends the bond absorbs the "message" e_i e_j^T (a chi x chi matrix with a single one at [ket i][bra j]) -- `cut_sel[edge]` below.
template <class T> static void region_contract(State* s, int nr, const int32_t* rv, const int32_t* parent, const double* ops /* may be null */,
                                               double* out_re_im, const std::unordered_map<int, Buf>& cut_sel) {
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
    const bool sharded = s->sharded();
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
                    if (parent[pos[k]] >= 0 && rv[parent[pos[k]]] == u) mp = up[pos[k]]->p;      // a child: its message
                    else mp = cut_sel.at(g.edge(u, k))->p;                                      // a cut bond: this term's index selector
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
    materialize_pending_all(s);
    for (int i = 0; i < nr; ++i) {
        if (rv[i] < 0 || rv[i] >= g.nv) throw Err(TNQS_ERR_INVALID, "expect_region: bad vertex");
        if (parent[i] < 0) ++roots; else if (parent[i] >= nr || g.edge(rv[i], rv[parent[i]]) < 0) throw Err(TNQS_ERR_INVALID, "expect_region: parent is not a neighbour");
    }
    if (roots != 1) throw Err(TNQS_ERR_INVALID, "expect_region: exactly one root expected");
    HIPCHK(hipSetDevice(s->device));
    // cut bonds: edges between two region vertices of which neither is the other's parent
    std::vector<int> pos(g.nv, -1); for (int i = 0; i < nr; ++i) { if (pos[rv[i]] >= 0) throw Err(TNQS_ERR_INVALID, "expect_region: repeated vertex"); pos[rv[i]] = i; }
    std::vector<int> cuts; double combos = 1;
    for (int e = 0; e < g.ne; ++e) {
        const int a = pos[g.esrc[e]], b = pos[g.edst[e]];
        if (a < 0 || b < 0 || parent[a] == b || parent[b] == a) continue;
        cuts.push_back(e); combos *= (double)s->chi[e] * s->chi[e];
    }
    // every term is two region contractions with a read-back each: 2^16 terms are tens of seconds of blocked time, 2^20 (two cut bonds at chi = 32) would be hours
    // without progress or a way to interrupt (round-4 advisor finding)
    if (combos > 65536.0) throw Err(TNQS_ERR_UNSUPPORTED, "expect_region: the region's loops need more than 2^16 terms (product of chi^2 over the bonds that close a loop)");
    std::unordered_map<int, Buf> sel;
    const size_t esz = s->esz();
    for (int e : cuts) { sel[e] = dalloc(s, (size_t)s->chi[e] * s->chi[e] * esz); }
    std::vector<int> ij(2 * cuts.size(), 0);          // odometer over (i, j) of every cut bond
    double acc[4] = {0, 0, 0, 0};
    const double one_d[2] = {1.0, 0.0}; const float one_f[2] = {1.f, 0.f};
    for (;;) {
        for (size_t c = 0; c < cuts.size(); ++c) {
            const int e = cuts[c], n = s->chi[e];
            HIPCHK(hipMemsetAsync(sel[e]->p, 0, (size_t)n * n * esz, s->stream));
            HIPCHK(hipMemcpyAsync(reinterpret_cast<char*>(sel[e]->p) + ((size_t)ij[2 * c] + (size_t)n * ij[2 * c + 1]) * esz,
                                  s->dtype == TNQS_C64 ? (const void*)one_f : (const void*)one_d, esz, hipMemcpyHostToDevice, s->stream));
        }
        double term[4];
        if (s->dtype == TNQS_C64) { region_contract<float>(s, nr, rv, parent, ops, term, sel); region_contract<float>(s, nr, rv, parent, nullptr, term + 2, sel); }
        else { region_contract<double>(s, nr, rv, parent, ops, term, sel); region_contract<double>(s, nr, rv, parent, nullptr, term + 2, sel); }
        for (int k = 0; k < 4; ++k) acc[k] += term[k];
        size_t c = 0;
        for (; c < cuts.size(); ++c) {
            const int n = s->chi[cuts[c]];
            if (++ij[2 * c] < n) break; ij[2 * c] = 0;
            if (++ij[2 * c + 1] < n) break; ij[2 * c + 1] = 0;
        }
        if (c == cuts.size()) break;
    }
    for (int k = 0; k < 4; ++k) out4[k] = acc[k];
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
        s->keepalive.push_back(s->site[verts[i]]); s->site[verts[i]] = nb; s->unit_norm[verts[i]] = 0;
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
