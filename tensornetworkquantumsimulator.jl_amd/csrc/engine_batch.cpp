// engine_batch.cpp -- batched building blocks of the engine: mode-product chains, Gram jobs, the SVD batch.  Host orchestration only; every
// flop runs in the HIP kernels of kernels*.hip.
#include "engine_internal.hpp"

namespace tnqs {

// ---------------------------------------------------------------------------------------------------------------
// batched building blocks
// ---------------------------------------------------------------------------------------------------------------
template <class T> void run_chains(State* s, std::vector<Chain>& chains, int cls, int cls_pair) {
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
            const int spw = pair_spw(tot_slices);
            for (auto& se : sel) {
                Chain& c = chains[se.first];
                int x = c.steps[se.second.first].first, y = c.steps[se.second.second].first;
                PairItem it{};
                Buf& dst = c.tmp[nt[se.first] & 1];
                if (!dst) dst = dalloc(s, c.sd.n * esz);
                it.in = c.result; it.out = dst->p; it.Mx = c.steps[se.second.first].second; it.My = c.steps[se.second.second].second;
                if (!pair_geometry(c.sd.d, c.sd.z, c.sd.chi.data(), x, y, it.g)) throw Err(TNQS_ERR_HIP, "internal: pair geometry");
                int nslices = it.g.n0 * it.g.n1 * it.g.n2;
                it.spw = spw; it.slice_begin = wgs; wgs += pair_wgs(nslices, spw);
                items.push_back(it);
                c.result = dst->p; nt[se.first]++; c.trail.push_back({x, y});
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
                if (c.ordered) {                                                     // the caller's first two legs, when they form a plane
                    Pair16Item it{};
                    if (plane_geometry(c.sd.d, c.sd.z, c.sd.chi.data(), c.steps[0].first, c.steps[1].first, 16, it.g)) {
                        it.Mx = c.steps[0].second; it.My = c.steps[1].second;
                        it16.push_back(it); sel16.push_back({ci, {0, 1}}); slices16 += (double)it.g.nslices(); found = true;
                    }
                }
                // the LOWEST remaining leg with the HIGHEST one (round 5): a plane that contains leg 0 streams at 4.4 - 4.55 TB/s, every other one at 3.7 - 3.9 and
                // (3,5) at 3.0 (profiles/plane16_bench.py, 12 sites, random data) -- "the two highest legs first" gave the y-lines of the cubic lattice (cross legs
                // 0, 2, 3, 5) the passes (3,5) + (0,2) = 3.57 ms per 12 sites, this rule (0,5) + (2,3) = 3.16; x-lines 3.17 -> 3.14, z-lines 3.33 -> 3.38
                for (int qx = 0; qx + 1 < (int)c.steps.size() && !found; ++qx)
                    for (int qy = (int)c.steps.size() - 1; qy > qx && !found; --qy) {
                        Pair16Item it{};
                        if (!plane_geometry(c.sd.d, c.sd.z, c.sd.chi.data(), c.steps[qx].first, c.steps[qy].first, 16, it.g)) continue;
                        it.Mx = c.steps[qx].second; it.My = c.steps[qy].second;
                        it16.push_back(it); sel16.push_back({ci, {qx, qy}}); slices16 += (double)it.g.nslices(); found = true;
                    }
            }
            if (it16.empty()) break;
            // slices per workgroup: a multiple of 4 (4 slices at a time, as 8 waves x half slices or 4 waves x whole slices), at least ~8 workgroups per CU overall
            int spw = 4; while (spw < 64 && slices16 / (2 * spw) >= 2048.0) spw *= 2;
            std::vector<Pair16Item> kind[2]; int wgs16[2] = {0, 0};           // [1]: whole 128-byte lines per wave (pair16_whole_lines)
            for (size_t q = 0; q < it16.size(); ++q) {
                Chain& c = chains[sel16[q].first]; Pair16Item& it = it16[q];
                const int wl = pair16_whole_lines(it.g) ? 1 : 0;
                Buf& dst = c.tmp[nt[sel16[q].first] & 1];
                if (!dst) dst = dalloc(s, c.sd.n * esz);
                it.in = c.result; it.out = dst->p; it.spw = spw; it.wg_begin = wgs16[wl]; wgs16[wl] += (it.g.nslices() + spw - 1) / spw;
                kind[wl].push_back(it);
                c.result = dst->p; nt[sel16[q].first]++;
                c.trail.push_back({c.steps[sel16[q].second.first].first, c.steps[sel16[q].second.second].first});
                c.steps.erase(c.steps.begin() + sel16[q].second.second); c.steps.erase(c.steps.begin() + sel16[q].second.first);
                by16 += 2.0 * c.sd.n * esz; fl16 += 2 * 8.0 * c.sd.n * 16;
            }
            ProfScope ps(s, cls_pair, by16, fl16);
            for (int wl = 0; wl < 2; ++wl) {
                if (kind[wl].empty()) continue;
                const Pair16Item* d = upload(s, kind[wl]);
                launch_mfma_pair16(s->stream, d, (int)kind[wl].size(), wgs16[wl], wl == 1);
            }
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
        std::vector<FiberItem> rg_items, rg32_items; double rg_tiles = 0, rg32_tiles = 0, rg_bytes = 0, rg_flops = 0;      // chi = 64 (32) legs: register-direct MFMA kernel
        for (size_t ci = 0; ci < chains.size(); ++ci) {
            Chain& c = chains[ci];
            if (c.steps.size() <= o) continue;
            int j = c.steps[o].first;
            c.trail.push_back({j});
            FiberItem it{};
            Buf& dst = c.tmp[nt[ci] & 1];
            if (!dst) dst = dalloc(s, c.sd.n * esz);
            it.in = c.result; it.out = dst->p; it.X = c.steps[o].second;
            it.D = 1; it.PA = (int)c.sd.pre(j); it.K = c.sd.chi[j]; it.PB = (int)c.sd.post(j); it.Do = 1; it.No = it.K;
            if (std::is_same<T, float>::value && use_mfma() && rowgemm_covers(it) && (it.K != 64 || use_chi64())) {
                rowgemm_tiles(it); it.want_norm = 0;
                (it.K == 64 ? rg_items : rg32_items).push_back(it); (it.K == 64 ? rg_tiles : rg32_tiles) += (double)it.nta * it.ntb;
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
        bool booked = false;
        for (int pass = 0; pass < 2; ++pass) {
            std::vector<FiberItem>& ri = pass ? rg32_items : rg_items;
            if (ri.empty()) continue;
            const double nt = pass ? rg32_tiles : rg_tiles;
            int tpw = (int)std::max(4.0, std::min(64.0, nt / 2048.0)); tpw &= ~3; int wgs = 0;
            for (auto& it : ri) { it.tpw = tpw; it.tile_begin = wgs; wgs += (it.nta * it.ntb + tpw - 1) / tpw; }
            const FiberItem* d = upload(s, ri);
            ProfScope ps(s, cls, booked ? 0.0 : rg_bytes, booked ? 0.0 : rg_flops); booked = true;      // bytes / flops of both groups are booked on the first scope
            launch_mfma_rowgemm(s->stream, d, (int)ri.size(), wgs, 1, pass ? 32 : 64, nullptr);
        }
        if (items.empty()) continue;
        // ComplexF64: the f64 matrix cores (kernels_f64.hip) when every product of the pass is one the kernel takes
        bool f64mf = std::is_same<T, double>::value && use_mfma();
        for (auto& it : items) f64mf = f64mf && fiber_gemm_f64_covers(it);
        if (f64mf) {
            double tot = 0; for (auto& it : items) { fiber_gemm_f64_tiles(it); tot += (double)it.nta * it.ntb; }
            int tpw = (int)std::max(8.0, std::min(64.0, tot / 4096.0)); tpw &= ~3;
            int wgs = 0;
            for (auto& it : items) { it.tpw = tpw; it.tile_begin = wgs; wgs += (it.nta * it.ntb + tpw - 1) / tpw; }
            const FiberItem* d = upload(s, items);
            ProfScope ps(s, cls, bytes, flops);
            launch_mfma_fiber_gemm_f64(s->stream, d, (int)items.size(), wgs, (int)KKmax, (int)KKmax, nullptr, false);
            continue;
        }
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
template <class T> void svd_batch(State* s, const std::vector<JacobiItem>& all, bool with_v) {
    const size_t esz = s->esz();
    const size_t cap = 160 * 1024 - 2048;
    std::vector<JacobiItem> fit, tall, rest, pre;
    static const bool force_global = [] { const char* e = std::getenv("TNQS_JACOBI_GLOBAL"); return e && e[0] == '1'; }();
    for (auto& j : all) {
        if (j.n < 1 || j.m < 1) continue;
        // a low-rank theta the caller offers to the preconditioned kernel (JacobiItem::pre): that kernel takes it when the dimensions found on the device
        // fit; the item stays in the lists below as well, whose kernels skip it in that case
        if (j.pre) pre.push_back(j);
        if (!force_global && jacobi_lds_bytes(j.m, j.n, with_v, esz) <= cap && std::max(j.m, j.n) <= 256) fit.push_back(j);
        else if (!force_global && std::is_same<T, float>::value && !with_v && !j.V && use_mfma() && use_chi64() && j.m >= j.n && j.n <= 128 && j.n >= 2 &&
                 jacobi_lds_bytes(j.n, j.n, false, esz) <= cap) tall.push_back(j);
        else rest.push_back(j);
    }
    if (!pre.empty()) {
        const JacobiItem* d = upload_small(s, pre);
        // LDS for the largest matrix the device may find: the low-rank factor (nhint columns) OR, when that route is withdrawn on the device, theta itself (j.n)
        int mm = 1, nn = 1; for (auto& j : pre) { mm = std::max(mm, std::min(j.m, 128)); nn = std::max(nn, std::min(j.n, 64)); }
        launch_theta_svd_pre(s->stream, d, (int)pre.size(), 60, mm, nn);
    }
    if (!fit.empty()) {
        size_t lds = 0; for (auto& j : fit) lds = std::max(lds, jacobi_lds_bytes(j.m, j.n, with_v, esz));
        const JacobiItem* d = upload_small(s, fit);
        int ncols = 1; for (auto& j : fit) ncols = std::max(ncols, j.nhint > 0 ? j.nhint : j.n);
        launch_jacobi<T>(s->stream, d, (int)fit.size(), 60, lds, mmax_of(fit), ncols);
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
            oR0[i] = off; off += round256(nn * 8); oRr[i] = off; off += round256(nn * 8); oJ[i] = off; off += round256(nn * 16); oT[i] = off; off += round256(mn * 8);
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
            wi.push_back(TallSvdItem{nullptr, reinterpret_cast<int*>(d_fail->p) + i, ap + oW[i], ap + oJ[i], ap + oRr[i], n, n});   // J = R^-1 (R J), f64; G slot: the item's Cholesky failure flag (J := I then)
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
        launch_tall_mj(s->stream, dg, (int)nt);                        // A J in f64 (J is complex128): column-relative accuracy, no polishing needed
        launch_copy_items(s->stream, dcp, (int)nt);
        // Polishing sweeps on A J itself are only needed where the preprocessing gave nothing to build on: an item whose Cholesky pivot
        // collapsed in spite of the shift (J := I above) is factorised from scratch here; every other item is skipped on the device
        // (JacobiItem::only_if).  Round 2 polished every item (2.6 ms per chi = 64 colour batch): its A J was an f32 product with an f32 J,
        // which leaves eps32 sigma_max of residue in every column -- the f64 product does not.
        std::vector<JacobiItem> pol;
        for (size_t i = 0; i < nt; ++i) { JacobiItem j{tall[i].A, nullptr, tall[i].m, tall[i].n, nullptr}; j.only_if = reinterpret_cast<const int*>(d_fail->p) + i; pol.push_back(j); }
        const JacobiItem* dp = upload(s, pol);
        launch_jacobi<T>(s->stream, dp, (int)nt, 60, 0, mmax);
        s->stats.n_tall_svd += (int)nt;
    }
}

template <class T, class Acc> void run_grams(State* s, std::vector<GramJob>& jobs, int cls) {
    if (jobs.empty()) return;
    const size_t esz = s->esz();
    size_t KKmax = 1;
    for (auto& j : jobs) { j.KK = (j.keep_site ? j.sd.d : 1) * (j.leg >= 0 ? j.sd.chi[j.leg] : 1); KKmax = std::max<size_t>(KKmax, j.KK); }
    int TR = pick_TR(KKmax + 1, esz, 2);
    // M set: the BP Gram absorbs the first row leg (f32 accumulation, mfma_gram32_fused_kernel); the gate-path Gram (f64 accumulation)
    // absorbs the last gauge leg (mfma_gauge_gram64_kernel) -- a batch is one or the other
    const bool gauge_fused = jobs[0].M != nullptr && std::is_same<T, float>::value && std::is_same<Acc, double>::value;
    const bool fused = jobs[0].M != nullptr && !gauge_fused;
    const bool mf = fused || (std::is_same<T, float>::value && std::is_same<Acc, float>::value && use_mfma() && KKmax <= (use_chi64() ? 64 : 32) && KKmax >= 8);
    bool mf64 = std::is_same<T, float>::value && std::is_same<Acc, double>::value && use_mfma() && KKmax <= 64 && KKmax >= 16;
    bool mf128 = std::is_same<T, float>::value && std::is_same<Acc, double>::value && use_mfma() && use_chi64() && KKmax <= 128 && KKmax > 64;
    for (auto& j : jobs) { mf64 = mf64 && (j.X == j.Y); mf128 = mf128 && (j.X == j.Y); }
    if (gauge_fused) { mf64 = true; mf128 = false; }
    const bool gauge16 = gauge_fused && KKmax == 32;      // 16-dimensional legs: the wave-private kernel (units of one fiber of r, one partial per chunk)
    if (mf || mf64 || mf128) TR = 64;
    // ComplexF64 operands: tiles of 32 fibers through LDS, f64 matrix cores (kernels_f64.hip)
    bool mf64in = std::is_same<T, double>::value && use_mfma() && jobs[0].M == nullptr;
    for (auto& j : jobs) mf64in = mf64in && j.M == nullptr && gram_f64in_covers(j.keep_site ? j.sd.d : 1, j.leg >= 0 ? j.sd.chi[j.leg] : 1);
    if (mf64in) TR = 32;
    // workgroups per launch.  The f64 Grams of the gate path write one 64 KiB partial per (site, chunk, tile parity) which reduce_kernel reads back: 2048 chunks were
    // 268 MB and 85-90 us per colour batch WHATEVER its size; 1024 (four workgroups per CU) halves that and costs the Gram pass nothing measurable
    const int target = (std::is_same<T, float>::value && std::is_same<Acc, double>::value) ? 1024 : 2048;
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
        if (gauge16) { it.nta = gauge_gram32_units(j.sd.z, j.sd.chi.data(), j.leg); it.ntb = 1; }
        int ntiles = it.nta * it.ntb;
        int nch = std::min(per_item, ntiles);
        it.tiles_per_chunk = (ntiles + nch - 1) / nch;
        it.nchunks = (ntiles + it.tiles_per_chunk - 1) / it.tiles_per_chunk;
        // 32 x 32 f32 MFMA kernels: one partial per wave; f64 64 x 64 MFMA kernel: one per tile parity; the chi = 64 kernels: one per chunk
        j.nchunks = (mf && (fused || KKmax <= 32)) ? 4 * it.nchunks : ((mf64 && !gauge16) ? 2 * it.nchunks : it.nchunks);
        j.partial = dalloc(s, (size_t)j.nchunks * j.KK * j.KK * 2 * sizeof(Acc));
        it.partial = j.partial->p; it.chunk_begin = chunks; chunks += it.nchunks;
        items.push_back(it);
        bytes += (j.X == j.Y ? 1.0 : 2.0) * j.sd.n * esz; flops += gauge_fused ? 8.0 * j.sd.n * (j.KK + (gauge16 ? 16.0 : 32.0)) : 8.0 * j.sd.n * j.KK * (j.M ? 2.0 : 1.0);
    }
    const GramItem* d = upload(s, items);
    ProfScope ps(s, cls, bytes, flops);
    if (gauge16) launch_mfma_gauge_gram32(s->stream, d, (int)items.size(), chunks);
    else if (gauge_fused) launch_mfma_gauge_gram64(s->stream, d, (int)items.size(), chunks);
    else if (fused) launch_mfma_gram32_fused(s->stream, d, (int)items.size(), chunks);
    else if (mf64) { bool all64 = true; for (auto& j : jobs) all64 = all64 && j.KK == 64; launch_mfma_gram64_f64(s->stream, d, (int)items.size(), chunks, (int)KKmax, all64); }
    else if (mf128) { bool all128 = true; for (auto& j : jobs) all128 = all128 && j.KK == 128; launch_mfma_gram128_f64(s->stream, d, (int)items.size(), chunks, (int)KKmax, all128); }
    else if (mf64in) launch_mfma_gram_f64in(s->stream, d, (int)items.size(), chunks);
    else if (mf) { if (KKmax <= 32) launch_mfma_gram32(s->stream, d, (int)items.size(), chunks, (int)KKmax); else launch_mfma_gram64(s->stream, d, (int)items.size(), chunks, (int)KKmax); }
    else launch_gram<T, Acc>(s->stream, d, (int)items.size(), chunks, TR, (int)KKmax);
}

template void run_chains<float>(State*, std::vector<Chain>&, int, int);
template void run_chains<double>(State*, std::vector<Chain>&, int, int);
template void svd_batch<float>(State*, const std::vector<JacobiItem>&, bool);
template void svd_batch<double>(State*, const std::vector<JacobiItem>&, bool);
template void run_grams<float, float>(State*, std::vector<GramJob>&, int);
template void run_grams<float, double>(State*, std::vector<GramJob>&, int);
template void run_grams<double, double>(State*, std::vector<GramJob>&, int);

}  // namespace tnqs
