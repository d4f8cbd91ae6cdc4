// debug.cpp -- kernel-level test entry points (include/tnqs_debug.h): host arrays in, host arrays out.
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "engine.hpp"
#include "kernels.hpp"
#include "launch_util.hpp"

namespace tnqs {
#define HIPCHK(x) hipchk((x), #x)
struct DBuf { void* p = nullptr; explicit DBuf(size_t n) { HIPCHK(hipMalloc(&p, n ? n : 1)); } ~DBuf() { (void)hipFree(p); }
              void up(const void* h, size_t n) { HIPCHK(hipMemcpy(p, h, n, hipMemcpyHostToDevice)); } void down(void* h, size_t n) { HIPCHK(hipMemcpy(h, p, n, hipMemcpyDeviceToHost)); } };
static void need_gpu() { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) throw Err(TNQS_ERR_HIP, "no HIP device available"); }

void dbg_jacobi(int dtype, int m, int n, void* A, void* V, int* sweeps) {
    need_gpu();
    if (m < 1 || n < 1 || m > 256 || n > 256) throw Err(TNQS_ERR_INVALID, "dbg_jacobi: 1 <= m, n <= 256");
    size_t esz = dtype == TNQS_C64 ? 8 : 16;
    DBuf dA((size_t)m * n * esz), dV((size_t)n * n * esz), dS(4), dI(sizeof(JacobiItem));
    dA.up(A, (size_t)m * n * esz);
    if (dtype == TNQS_C64) launch_identity<float>(nullptr, dV.p, n); else launch_identity<double>(nullptr, dV.p, n);
    // same residency policy as the engine: A+V in LDS, else A in LDS with V recovered, else global memory
    const size_t lim = 160 * 1024 - 2048;
    size_t lds_av = jacobi_lds_bytes(m, n, true, esz), lds_a = jacobi_lds_bytes(m, n, false, esz);
    const char* fg = std::getenv("TNQS_DBG_NOV_GLOBAL");      // the engine's combination for matrices beyond the LDS: global-memory kernel, V recovered
    const bool force_global_nov = fg && fg[0] == '1';
    const char* ft = std::getenv("TNQS_DBG_THETA_SVD");       // the engine's theta route: A only in LDS, V recovered from the unrotated copy
    const bool theta_route = ft && ft[0] == '1' && dtype == TNQS_C64 && lds_a <= lim;
    const bool nov = force_global_nov || theta_route || (lds_av > lim && lds_a <= lim);
    DBuf dA0((size_t)m * n * esz);
    if (nov) dA0.up(A, (size_t)m * n * esz);
    JacobiItem it{dA.p, nov ? nullptr : dV.p, m, n, (int*)dS.p};
    dI.up(&it, sizeof(it));
    size_t lds = force_global_nov ? 0 : (nov ? lds_a : (lds_av <= lim ? lds_av : 0));
    if (dtype == TNQS_C64) launch_jacobi<float>(nullptr, (const JacobiItem*)dI.p, 1, 60, lds, std::max(m, n)); else launch_jacobi<double>(nullptr, (const JacobiItem*)dI.p, 1, 60, lds, std::max(m, n));
    if (nov) {
        DBuf dRv(sizeof(RecoverItem)); RecoverItem rv{dA0.p, dA.p, dV.p, m, n, n}; dRv.up(&rv, sizeof(rv));
        if (dtype == TNQS_C64) launch_recover_v_mfma(nullptr, (const RecoverItem*)dRv.p, 1, n); else launch_recover_v<double>(nullptr, (const RecoverItem*)dRv.p, 1, n);
        HIPCHK(hipDeviceSynchronize());
    }
    HIPCHK(hipDeviceSynchronize());
    dA.down(A, (size_t)m * n * esz); dV.down(V, (size_t)n * n * esz);
    if (sweeps) dS.down(sweeps, 4);
}

// theta_svd_pre_kernel on one ComplexF32 factor A (m x n) of theta = A Q^T, Q (nq x n, complex128, orthonormal columns): A := U Sigma, V (nq x n) := conj(Q) U_L.
// reps > 0: additionally time `reps` launches on `copies` device-resident copies of A (one workgroup each) with HIP events -> *ms = average per launch
void dbg_theta_svd_pre(int m, int n, int nq, void* A, const void* Q, void* V, int* sweeps, int copies, int reps, double* ms, double* phase_us, int cap) {
    need_gpu();
    if (!Q) nq = n;                                 // no Q: theta itself is factorised, V is n x n
    if (!theta_svd_pre_covers(m, n) || nq < n || nq > m) throw Err(TNQS_ERR_INVALID, "dbg_theta_svd_pre: 2 <= n <= 64, n <= nq <= m <= 128");
    if (copies < 1) copies = 1;
    const size_t ab = (size_t)m * n * 8, vb = (size_t)nq * n * 8;
    DBuf dA(ab * copies), dA0(ab), dQ((size_t)nq * n * 16), dV(vb * copies), dS(4 * (size_t)copies), dInfo(32), dI(sizeof(JacobiItem) * (size_t)copies), dT(64);
    dA0.up(A, ab); if (Q) dQ.up(Q, (size_t)nq * n * 16);
    const int info[8] = {m, nq, 0, 0, 0, 0, 0, Q ? n : 0};      // theta_dims with d1 = d2 = 1: m rows, nq columns of theta, n columns of the factor (0: theta itself)
    dInfo.up(info, 32);
    std::vector<JacobiItem> its(copies);
    for (int c = 0; c < copies; ++c) {
        JacobiItem it{}; it.A = (char*)dA.p + ab * c; it.V = (c == 0 && phase_us) ? dT.p : nullptr; it.m = m; it.n = nq; it.sweeps_out = (int*)dS.p + c; it.dyn = (const int*)dInfo.p; it.dm = 1; it.dn = 1; it.nhint = n;
        it.QB = Q ? dQ.p : nullptr; it.Vout = (char*)dV.p + vb * c; it.pre = 0 /* the dimensions given decide, whatever the size */; it.cap = cap; its[c] = it;
    }
    dI.up(its.data(), sizeof(JacobiItem) * (size_t)copies);
    auto reset = [&]() { for (int c = 0; c < copies; ++c) HIPCHK(hipMemcpyAsync((char*)dA.p + ab * c, dA0.p, ab, hipMemcpyDeviceToDevice, nullptr)); };
    reset();
    launch_theta_svd_pre(nullptr, (const JacobiItem*)dI.p, copies, 60, m, n);
    HIPCHK(hipDeviceSynchronize());
    dA.down(A, ab); dV.down(V, vb);
    if (sweeps) dS.down(sweeps, 4);
    if (phase_us) {      // constant-rate clock (100 MHz) at the seven phase boundaries of workgroup 0 -> six durations in us
        unsigned long long t[8]; dT.down(t, 64);
        for (int k = 0; k < 6; ++k) phase_us[k] = (double)(t[k + 1] - t[k]) * 0.01;
    }
    if (reps > 0 && ms) {
        hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
        double tot = 0;
        for (int r = 0; r < reps; ++r) {
            reset();
            HIPCHK(hipEventRecord(e0, nullptr));
            launch_theta_svd_pre(nullptr, (const JacobiItem*)dI.p, copies, 60, m, n);
            HIPCHK(hipEventRecord(e1, nullptr)); HIPCHK(hipEventSynchronize(e1));
            float t = 0; HIPCHK(hipEventElapsedTime(&t, e0, e1)); tot += t;
        }
        *ms = tot / reps;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
}
// the plain LDS-resident Jacobi on the same factor, timed the same way (what the preconditioned kernel replaces)
void dbg_time_jacobi_f32(int m, int n, const void* A, int copies, int reps, double* ms, int* sweeps) {
    need_gpu();
    if (copies < 1) copies = 1;
    const size_t ab = (size_t)m * n * 8;
    DBuf dA(ab * copies), dA0(ab), dS(4 * (size_t)copies), dI(sizeof(JacobiItem) * (size_t)copies);
    dA0.up(A, ab);
    std::vector<JacobiItem> its(copies);
    for (int c = 0; c < copies; ++c) { JacobiItem it{}; it.A = (char*)dA.p + ab * c; it.m = m; it.n = n; it.sweeps_out = (int*)dS.p + c; its[c] = it; }
    dI.up(its.data(), sizeof(JacobiItem) * (size_t)copies);
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    double tot = 0;
    for (int r = 0; r < reps + 1; ++r) {
        for (int c = 0; c < copies; ++c) HIPCHK(hipMemcpyAsync((char*)dA.p + ab * c, dA0.p, ab, hipMemcpyDeviceToDevice, nullptr));
        HIPCHK(hipEventRecord(e0, nullptr));
        launch_jacobi<float>(nullptr, (const JacobiItem*)dI.p, copies, 60, jacobi_lds_bytes(m, n, false, 8), std::max(m, n), n);
        HIPCHK(hipEventRecord(e1, nullptr)); HIPCHK(hipEventSynchronize(e1));
        float t = 0; HIPCHK(hipEventElapsedTime(&t, e0, e1)); if (r > 0) tot += t;
    }
    if (ms) *ms = tot / std::max(1, reps);
    if (sweeps) dS.down(sweeps, 4);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}

// Cholesky kernels on one Hermitian n x n complex128 matrix: L (lower), W = (L^-1)^dagger, *fail; n <= 96: chol_kernel, else packed
void dbg_chol(int n, const void* G, void* Lout, void* Wout, int* fail, double tau) {
    need_gpu();
    if (n < 1 || n > 128) throw Err(TNQS_ERR_INVALID, "dbg_chol: 1 <= n <= 128");
    const size_t b = (size_t)n * n * 16;
    DBuf dG(b), dL(b), dW(b), dF(4), dI(sizeof(CholItem));
    dG.up(G, b); HIPCHK(hipMemset(dF.p, 0, 4)); HIPCHK(hipMemset(dW.p, 0, b));
    CholItem it{dG.p, dL.p, dW.p, n, (int*)dF.p, tau, 0.0};
    dI.up(&it, sizeof(it));
    if (n <= 96) launch_chol(nullptr, (const CholItem*)dI.p, 1, n); else launch_chol_packed(nullptr, (const CholItem*)dI.p, 1, n);
    HIPCHK(hipDeviceSynchronize());
    dL.down(Lout, b); dW.down(Wout, b); dF.down(fail, 4);
}

static void tile_params(size_t PA, size_t PB, int TR, int& TA, int& TB, int& nta, int& ntb) {
    TA = (int)std::min<size_t>(PA, TR); TB = std::max(1, TR / TA); TB = (int)std::min<size_t>(TB, PB);
    nta = (int)((PA + TA - 1) / TA); ntb = (int)((PB + TB - 1) / TB);
}
static int pick_TR(size_t KK, size_t esz, int copies) {
    for (int tr : {64, 32, 16, 8, 4}) if (KK * tr * esz * copies <= 64 * 1024) return tr;
    throw Err(TNQS_ERR_UNSUPPORTED, "too large");
}

void dbg_fiber_gemm(int dtype, int D, int PA, int K, int PB, int Do, int No, const void* in, const void* X, void* out, double* norm2, int use_mfma) {
    need_gpu();
    size_t esz = dtype == TNQS_C64 ? 8 : 16;
    size_t nin = (size_t)D * PA * K * PB, nout = (size_t)Do * PA * No * PB, nx = (size_t)D * K * Do * No;
    DBuf dIn(nin * esz), dX(nx * esz), dOut(nout * esz), dI(sizeof(FiberItem));
    dIn.up(in, nin * esz); dX.up(X, nx * esz);
    HIPCHK(hipMemset(dOut.p, 0xff, nout * esz));
    FiberItem it{}; it.in = dIn.p; it.out = dOut.p; it.X = dX.p; it.D = D; it.PA = PA; it.K = K; it.PB = PB; it.Do = Do; it.No = No;
    int TR = pick_TR((size_t)D * K, esz, 1);
    bool mf = use_mfma && dtype == TNQS_C64 && mfma_fiber_tile_rows(D * K, Do * No) > 0;
    if (mf) TR = mfma_fiber_tile_rows(D * K, Do * No);
    tile_params(PA, PB, TR, it.TA, it.TB, it.nta, it.ntb);
    it.tile_begin = 0; it.want_norm = 1; it.tpw = mf ? 5 : 1;
    const bool mf64 = use_mfma && dtype == TNQS_C128 && fiber_gemm_f64_covers(it);       // kernels_f64.hip (general form: D, Do, norm partial)
    if (mf64) { fiber_gemm_f64_tiles(it); it.tpw = 12; }
    int tiles = (it.nta * it.ntb + it.tpw - 1) / it.tpw;
    DBuf dN((size_t)tiles * 8);
    dI.up(&it, sizeof(it));
    if (mf64) launch_mfma_fiber_gemm_f64(nullptr, (const FiberItem*)dI.p, 1, tiles, D * K, Do * No, (double*)dN.p, true);
    else if (mf) launch_mfma_fiber_gemm(nullptr, (const FiberItem*)dI.p, 1, tiles, D * K, Do * No, (double*)dN.p);
    else if (dtype == TNQS_C64) launch_fiber_gemm<float>(nullptr, (const FiberItem*)dI.p, 1, tiles, TR, D * K, (double*)dN.p);
    else launch_fiber_gemm<double>(nullptr, (const FiberItem*)dI.p, 1, tiles, TR, D * K, (double*)dN.p);
    HIPCHK(hipDeviceSynchronize());
    dOut.down(out, nout * esz);
    if (norm2) { std::vector<double> np(tiles); dN.down(np.data(), (size_t)tiles * 8); double t = 0; for (double v : np) t += v; *norm2 = t; }
}

void dbg_gram(int dtype, int D, int PA, int K, int PB, const void* X, const void* Y, void* out, int acc64, int use_mfma) {
    need_gpu();
    size_t esz = dtype == TNQS_C64 ? 8 : 16;
    size_t nin = (size_t)D * PA * K * PB; int KK = D * K;
    bool same = (X == Y);
    DBuf dX(nin * esz), dY(same ? 1 : nin * esz), dI(sizeof(GramItem)), dR(sizeof(ReduceItem));
    dX.up(X, nin * esz); if (!same) dY.up(Y, nin * esz);
    GramItem it{}; it.X = dX.p; it.Y = same ? dX.p : dY.p; it.D = D; it.PA = PA; it.K = K; it.PB = PB;
    int TR = pick_TR((size_t)KK + 1, esz, 2);
    bool mf = use_mfma && dtype == TNQS_C64 && !acc64 && KK <= 32;
    bool mf64 = use_mfma && dtype == TNQS_C64 && acc64 && same && KK <= 64 && KK >= 16;
    if (mf || mf64) TR = 64;
    const bool mfz = use_mfma && dtype == TNQS_C128 && gram_f64in_covers(D, K);           // kernels_f64.hip
    if (mfz) TR = 32;
    tile_params(PA, PB, TR, it.TA, it.TB, it.nta, it.ntb);
    int ntiles = it.nta * it.ntb; int nch = std::min(7, ntiles);
    if (const char* e = std::getenv("TNQS_DBG_GRAM_CHUNKS")) nch = std::min(ntiles, std::max(1, std::atoi(e)));
    it.tiles_per_chunk = (ntiles + nch - 1) / nch; it.nchunks = (ntiles + it.tiles_per_chunk - 1) / it.tiles_per_chunk; it.chunk_begin = 0;
    bool a64 = acc64 || dtype == TNQS_C128;
    size_t asz = a64 ? 16 : 8;
    int npart = mf ? 4 * it.nchunks : (mf64 ? 2 * it.nchunks : it.nchunks);
    DBuf dP((size_t)npart * KK * KK * asz), dO((size_t)KK * KK * asz);
    it.partial = dP.p; dI.up(&it, sizeof(it));
    if (mfz) launch_mfma_gram_f64in(nullptr, (const GramItem*)dI.p, 1, it.nchunks);
    else if (mf64) launch_mfma_gram64_f64(nullptr, (const GramItem*)dI.p, 1, it.nchunks, KK, KK == 64);
    else if (mf) launch_mfma_gram32(nullptr, (const GramItem*)dI.p, 1, it.nchunks, KK);
    else if (dtype == TNQS_C64) { if (a64) launch_gram<float, double>(nullptr, (const GramItem*)dI.p, 1, it.nchunks, TR, KK); else launch_gram<float, float>(nullptr, (const GramItem*)dI.p, 1, it.nchunks, TR, KK); }
    else launch_gram<double, double>(nullptr, (const GramItem*)dI.p, 1, it.nchunks, TR, KK);
    ReduceItem ri{dP.p, dO.p, KK * KK, npart, 0, 0}; dR.up(&ri, sizeof(ri));
    if (a64) launch_reduce<double, double>(nullptr, (const ReduceItem*)dR.p, 1, KK * KK); else launch_reduce<float, float>(nullptr, (const ReduceItem*)dR.p, 1, KK * KK);
    HIPCHK(hipDeviceSynchronize());
    dO.down(out, (size_t)KK * KK * asz);
}
void dbg_pair(int C0, int NMID, int NHI, const void* in, const void* Mx, const void* My, void* out) {
    need_gpu();
    if (C0 % 16) throw Err(TNQS_ERR_INVALID, "dbg_pair: C0 % 16 != 0");
    size_t n = (size_t)C0 * 32 * NMID * 32 * NHI;
    DBuf dIn(n * 8), dOut(n * 8), dX(32 * 32 * 8), dY(32 * 32 * 8), dI(sizeof(PairItem));
    dIn.up(in, n * 8); dX.up(Mx, 32 * 32 * 8); dY.up(My, 32 * 32 * 8);
    HIPCHK(hipMemset(dOut.p, 0xff, n * 8));
    PairItem it{}; it.in = dIn.p; it.out = dOut.p; it.Mx = dX.p; it.My = dY.p; it.slice_begin = 0; it.spw = 3;
    it.g.cstr = 2; it.g.sx = C0; it.g.sy = (long long)C0 * 32 * NMID; it.g.n0 = C0 / 16; it.g.t0 = 16; it.g.n1 = NMID; it.g.t1 = (long long)C0 * 32;
    it.g.n2 = NHI; it.g.t2 = (long long)C0 * 32 * NMID * 32;
    int nslices = (C0 / 16) * NMID * NHI;
    dI.up(&it, sizeof(it));
    launch_mfma_pair(nullptr, (const PairItem*)dI.p, 1, pair_wgs(nslices, it.spw));
    HIPCHK(hipDeviceSynchronize());
    dOut.down(out, n * 8);
}
// pair product on the legs (lx, ly) of a site tensor [d][chi_0..chi_{z-1}] (ComplexF32): out = in x_lx Mx x_ly My
void dbg_pair_legs(int d, int z, const int* chi, int lx, int ly, const void* in, const void* Mx, const void* My, void* out) {
    need_gpu();
    PairItem it{};
    if (!pair_geometry(d, z, chi, lx, ly, it.g)) throw Err(TNQS_ERR_UNSUPPORTED, "dbg_pair_legs: shape not covered by the pair kernel");
    size_t n = d; for (int i = 0; i < z; ++i) n *= chi[i];
    DBuf dIn(n * 8), dOut(n * 8), dX(32 * 32 * 8), dY(32 * 32 * 8), dI(sizeof(PairItem));
    dIn.up(in, n * 8); dX.up(Mx, 32 * 32 * 8); dY.up(My, 32 * 32 * 8);
    HIPCHK(hipMemset(dOut.p, 0xff, n * 8));
    it.in = dIn.p; it.out = dOut.p; it.Mx = dX.p; it.My = dY.p; it.slice_begin = 0; it.spw = 3;
    int nslices = it.g.n0 * it.g.n1 * it.g.n2;
    if ((size_t)nslices * 16 * 1024 != n) throw Err(TNQS_ERR_INVALID, "dbg_pair_legs: slice count");
    dI.up(&it, sizeof(it));
    launch_mfma_pair(nullptr, (const PairItem*)dI.p, 1, pair_wgs(nslices, it.spw));
    HIPCHK(hipDeviceSynchronize());
    dOut.down(out, n * 8);
}
// out[b,b'] = sum (X x_lx M)[.., b, ..] conj(Y[.., b', ..]) with b on leg ly (ComplexF32, 32 x 32 output)
void dbg_pair_gram(int d, int z, const int* chi, int lx, int ly, const void* X, const void* Y, const void* M, void* out) {
    need_gpu();
    PairGramItem it{};
    if (!pair_geometry(d, z, chi, lx, ly, it.g)) throw Err(TNQS_ERR_UNSUPPORTED, "dbg_pair_gram: shape not covered by the pair kernel");
    size_t n = d; for (int i = 0; i < z; ++i) n *= chi[i];
    int nslices = it.g.n0 * it.g.n1 * it.g.n2;
    it.spw = 3; it.wg_begin = 0;
    const bool x3 = mfma_use_x3();       // the engine's route: the bf16 kernel in its one-message form (half-slice workgroups in groups of 16)
    int nwg = x3 ? 16 * (((nslices + it.spw - 1) / it.spw + 7) / 8) : (nslices + it.spw - 1) / it.spw, npart = nwg;
    DBuf dX(n * 8), dY(n * 8), dM(32 * 32 * 8), dI(sizeof(PairGram2Item)), dR(sizeof(ReduceItem)), dP((size_t)npart * 1024 * 8), dO(1024 * 8);
    dX.up(X, n * 8); dY.up(Y, n * 8); dM.up(M, 32 * 32 * 8);
    it.X = dX.p; it.Y = dY.p; it.M = dM.p; it.partial = dP.p;
    if (x3) {
        PairGram2Item one{}; one.X = it.X; one.Y = it.Y; one.Mx = it.M; one.My = nullptr; one.partial_y = it.partial; one.partial_x = nullptr; one.g = it.g; one.wg_begin = 0; one.spw = it.spw;
        dI.up(&one, sizeof(one));
        launch_x3_pair_gram1(nullptr, (const PairGram2Item*)dI.p, 1, nwg);
    } else {
        dI.up(&it, sizeof(it));
        launch_mfma_pair_gram(nullptr, (const PairGramItem*)dI.p, 1, nwg);
    }
    ReduceItem ri{dP.p, dO.p, 1024, npart, 0, 0}; dR.up(&ri, sizeof(ri));
    launch_reduce<float, float>(nullptr, (const ReduceItem*)dR.p, 1, 1024);
    HIPCHK(hipDeviceSynchronize());
    dO.down(out, 1024 * 8);
}
// psi' = psi x_(s,b) X for d = 2, chi_b = chi_b' = 32 through the plane kernel; returns |psi'|^2 in *norm2
// both messages of a plane in one pass: out_y keeps leg ly (lx absorbed with Mx), out_x keeps leg lx (ly absorbed with My)
void dbg_pair_gram2(int d, int z, const int* chi, int lx, int ly, const void* X, const void* Y, const void* Mx, const void* My, void* out_y, void* out_x) {
    need_gpu();
    PairGram2Item it{};
    if (!pair_geometry(d, z, chi, lx, ly, it.g)) throw Err(TNQS_ERR_UNSUPPORTED, "dbg_pair_gram2: shape not covered");
    size_t n = d; for (int i = 0; i < z; ++i) n *= chi[i];
    int nslices = it.g.n0 * it.g.n1 * it.g.n2;
    it.spw = 3; it.wg_begin = 0;
    int npairs = (nslices + it.spw - 1) / it.spw, nwg = pair_gram2_group() * ((npairs + 7) / 8), npart = nwg;
    DBuf dX(n * 8), dY(n * 8), dMx(1024 * 8), dMy(1024 * 8), dI(sizeof(PairGram2Item)), dR(2 * sizeof(ReduceItem)), dP1((size_t)npart * 1024 * 8), dP2((size_t)npart * 1024 * 8), dO(2 * 1024 * 8);
    dX.up(X, n * 8); dY.up(Y, n * 8); dMx.up(Mx, 1024 * 8); dMy.up(My, 1024 * 8);
    it.X = dX.p; it.Y = dY.p; it.Mx = dMx.p; it.My = dMy.p; it.partial_y = dP1.p; it.partial_x = dP2.p;
    dI.up(&it, sizeof(it));
    launch_mfma_pair_gram2(nullptr, (const PairGram2Item*)dI.p, 1, nwg);
    ReduceItem ri[2] = {{dP1.p, dO.p, 1024, npart, 0, 0}, {dP2.p, (char*)dO.p + 1024 * 8, 1024, npart, 0, 1024}};
    dR.up(ri, sizeof(ri));
    launch_reduce<float, float>(nullptr, (const ReduceItem*)dR.p, 2, 2048);
    HIPCHK(hipDeviceSynchronize());
    dO.down(out_y, 1024 * 8);
    HIPCHK(hipMemcpy(out_x, (char*)dO.p + 1024 * 8, 1024 * 8, hipMemcpyDeviceToHost));
}
// pseudo-random f32 fill in (-0.01, 0.01) for the timing entry points: constant data flatters the matrix kernels (fewer bits toggle, the chip clocks higher:
// the both-messages pair-Gram measured 1.0 ms per 100 sites on constant data and 1.29 in the benchmark)
static void fill_random(void* d, size_t nfloats, unsigned seed) {
    const size_t blk = (size_t)1 << 22;                   // 4 Mi floats generated on the host, tiled over the buffer with a shifting offset
    std::vector<float> h(blk + 4096);
    unsigned x = seed * 2654435761u + 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((int)(x >> 8) - (1 << 23)) * (0.01f / (1 << 23)); }
    size_t off = 0; unsigned k = 0;
    while (off < nfloats) {
        const size_t m = std::min(blk, nfloats - off);
        HIPCHK(hipMemcpy((char*)d + off * 4, h.data() + (k * 257u) % 4096u, m * 4, hipMemcpyHostToDevice));
        off += m; ++k;
    }
}
// timing of the chi = 32 plane kernels on `nsites` degree-4 site tensors [2][32]^4 resident in HBM (which: 0 pair product on legs (lx, ly),
// 1 both-messages pair-Gram); *ms = average launch duration over `reps` launches (HIP events), after one untimed launch
// which = 2 / 3: the chi = 16 plane kernels (mfma_pair16_kernel / mfma_pair_gram2x16_kernel, both messages) on degree-6 site tensors
static void dbg_bench_plane16(int which, int nsites, int lx, int ly, int reps, double* ms) {
    const int chi[6] = {16, 16, 16, 16, 16, 16};
    PlaneGeom g{};
    if (!plane_geometry(2, 6, chi, lx, ly, 16, g)) throw Err(TNQS_ERR_UNSUPPORTED, "dbg_bench_plane: legs not covered");
    const size_t n = (size_t)2 << 24;
    DBuf dA((size_t)nsites * n * 8), dB((size_t)nsites * n * 8), dM(2 * 256 * 8);
    fill_random(dA.p, (size_t)nsites * n * 2, 1); fill_random(dB.p, (size_t)nsites * n * 2, 2); fill_random(dM.p, 2 * 256 * 2, 3);
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    float t = 0.f;
    const double tot = (double)nsites * g.nslices();
    if (which == 2) {
        std::vector<Pair16Item> items(nsites);
        int spw = 4; while (spw < 64 && tot / (2 * spw) >= 2048.0) spw *= 2;
        int wgs = 0;
        for (int i = 0; i < nsites; ++i) {
            Pair16Item& it = items[i]; it.g = g; it.in = (char*)dA.p + (size_t)i * n * 8; it.out = (char*)dB.p + (size_t)i * n * 8;
            it.Mx = dM.p; it.My = (char*)dM.p + 256 * 8; it.wg_begin = wgs; it.spw = spw; wgs += (g.nslices() + spw - 1) / spw;
        }
        DBuf dI(items.size() * sizeof(Pair16Item)); dI.up(items.data(), items.size() * sizeof(Pair16Item));
        launch_mfma_pair16(nullptr, (const Pair16Item*)dI.p, nsites, wgs, pair16_whole_lines(g));
        HIPCHK(hipEventRecord(e0, nullptr));
        for (int r = 0; r < reps; ++r) launch_mfma_pair16(nullptr, (const Pair16Item*)dI.p, nsites, wgs, pair16_whole_lines(g));
        HIPCHK(hipEventRecord(e1, nullptr)); HIPCHK(hipEventSynchronize(e1)); HIPCHK(hipEventElapsedTime(&t, e0, e1));
    } else {
        std::vector<PairGram2x16Item> items(nsites);
        int spw = pair_gram2x16_slices_at_a_time(); while (spw < 128 && tot / (2 * spw) >= 2048.0) spw *= 2;
        int wgs = 0; const int nwg = (g.nslices() + spw - 1) / spw;
        DBuf dP((size_t)2 * nsites * nwg * 256 * 8);
        for (int i = 0; i < nsites; ++i) {
            PairGram2x16Item& it = items[i]; it.g = g; it.X = (char*)dA.p + (size_t)i * n * 8; it.Y = (char*)dB.p + (size_t)i * n * 8;
            it.Mx = dM.p; it.My = (char*)dM.p + 256 * 8; it.wg_begin = wgs; it.spw = spw;
            it.partial_y = (char*)dP.p + (size_t)(2 * i) * nwg * 256 * 8; it.partial_x = (char*)dP.p + (size_t)(2 * i + 1) * nwg * 256 * 8;
            wgs += nwg;
        }
        DBuf dI(items.size() * sizeof(PairGram2x16Item)); dI.up(items.data(), items.size() * sizeof(PairGram2x16Item));
        launch_mfma_pair_gram2x16(nullptr, (const PairGram2x16Item*)dI.p, nsites, wgs);
        HIPCHK(hipEventRecord(e0, nullptr));
        for (int r = 0; r < reps; ++r) launch_mfma_pair_gram2x16(nullptr, (const PairGram2x16Item*)dI.p, nsites, wgs);
        HIPCHK(hipEventRecord(e1, nullptr)); HIPCHK(hipEventSynchronize(e1)); HIPCHK(hipEventElapsedTime(&t, e0, e1));
    }
    HIPCHK(hipDeviceSynchronize());
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms = (double)t / reps;
}
void dbg_bench_plane(int which, int nsites, int lx, int ly, int reps, double* ms) {
    need_gpu();
    if (which == 2 || which == 3) { dbg_bench_plane16(which, nsites, lx, ly, reps, ms); return; }
    const int chi[4] = {32, 32, 32, 32};
    PairGeom g{};
    if (!pair_geometry(2, 4, chi, lx, ly, g)) throw Err(TNQS_ERR_UNSUPPORTED, "dbg_bench_plane: legs not covered");
    const size_t n = (size_t)2 * 32 * 32 * 32 * 32, nslices = (size_t)g.n0 * g.n1 * g.n2;
    DBuf dA((size_t)nsites * n * 8), dB((size_t)nsites * n * 8), dM(2 * 1024 * 8);
    fill_random(dA.p, (size_t)nsites * n * 2, 1); fill_random(dB.p, (size_t)nsites * n * 2, 2); fill_random(dM.p, 2 * 1024 * 2, 3);
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    float t = 0.f;
    if (which == 4) {
        // one BP level of a degree-4 site: T = psi x_lx Mx x_ly My (pair product), then both messages through the other two legs from (T, psi).  The sites are
        // processed in groups of TNQS_DBG_GROUP (0: all at once, as the engine does): with a few sites per group T (16 MiB per site) is still in the 256 MiB
        // Infinity Cache when the pair-Gram reads it, and psi is read from memory once instead of twice
        const char* e = std::getenv("TNQS_DBG_GROUP"); int G = e ? std::atoi(e) : 0; if (G <= 0 || G > nsites) G = nsites;
        int ox = -1, oy = -1; for (int q = 0; q < 4; ++q) if (q != lx && q != ly) { if (ox < 0) ox = q; else oy = q; }
        PairGeom g2{}; if (!pair_geometry(2, 4, chi, ox, oy, g2)) throw Err(TNQS_ERR_UNSUPPORTED, "dbg_bench_plane: other legs not covered");
        const size_t ns2 = (size_t)g2.n0 * g2.n1 * g2.n2;
        const int ngroups = (nsites + G - 1) / G;
        int spw_p = pair_spw((double)G * nslices);
        int spw_g = 16; while (spw_g > 1 && (long)G * pair_gram2_group() * (((ns2 + spw_g - 1) / spw_g + 7) / 8) < 1024) spw_g >>= 1;
        const int npairs = (int)((ns2 + spw_g - 1) / spw_g), nwg_g = pair_gram2_group() * ((npairs + 7) / 8);
        DBuf dT((size_t)G * n * 8), dP((size_t)2 * nsites * nwg_g * 1024 * 8);
        std::vector<PairItem> pit(nsites); std::vector<PairGram2Item> git(nsites);
        std::vector<int> pw(ngroups, 0), gwg(ngroups, 0);
        for (int i = 0; i < nsites; ++i) {
            const int gr = i / G, k = i % G;
            PairItem& a = pit[i]; a.g = g; a.in = (char*)dA.p + (size_t)i * n * 8; a.out = (char*)dT.p + (size_t)k * n * 8; a.Mx = dM.p; a.My = (char*)dM.p + 1024 * 8;
            a.slice_begin = pw[gr]; a.spw = spw_p; pw[gr] += pair_wgs((int)nslices, spw_p);
            PairGram2Item& b = git[i]; b.g = g2; b.X = a.out; b.Y = a.in; b.Mx = dM.p; b.My = (char*)dM.p + 1024 * 8; b.wg_begin = gwg[gr]; b.spw = spw_g; gwg[gr] += nwg_g;
            b.partial_y = (char*)dP.p + (size_t)(2 * i) * nwg_g * 1024 * 8; b.partial_x = (char*)dP.p + (size_t)(2 * i + 1) * nwg_g * 1024 * 8;
        }
        DBuf dPI(pit.size() * sizeof(PairItem)), dGI(git.size() * sizeof(PairGram2Item));
        dPI.up(pit.data(), pit.size() * sizeof(PairItem)); dGI.up(git.data(), git.size() * sizeof(PairGram2Item));
        auto level = [&]() {
            for (int gr = 0; gr < ngroups; ++gr) {
                const int cnt = std::min(G, nsites - gr * G);
                launch_mfma_pair(nullptr, (const PairItem*)dPI.p + (size_t)gr * G, cnt, pw[gr]);
                launch_mfma_pair_gram2(nullptr, (const PairGram2Item*)dGI.p + (size_t)gr * G, cnt, gwg[gr]);
            }
        };
        level();
        HIPCHK(hipEventRecord(e0, nullptr));
        for (int r = 0; r < reps; ++r) level();
        HIPCHK(hipEventRecord(e1, nullptr)); HIPCHK(hipEventSynchronize(e1)); HIPCHK(hipEventElapsedTime(&t, e0, e1));
        std::fprintf(stderr, "level bench: group %d, pair spw %d, gram spw %d\n", G, spw_p, spw_g);
    } else if (which == 0) {
        std::vector<PairItem> items(nsites);
        const int spw = pair_spw((double)nsites * nslices); int wgs = 0;
        for (int i = 0; i < nsites; ++i) {
            PairItem& it = items[i]; it.g = g; it.in = (char*)dA.p + (size_t)i * n * 8; it.out = (char*)dB.p + (size_t)i * n * 8;
            it.Mx = dM.p; it.My = (char*)dM.p + 1024 * 8; it.slice_begin = wgs; it.spw = spw; wgs += pair_wgs((int)nslices, spw);
        }
        DBuf dI(items.size() * sizeof(PairItem)); dI.up(items.data(), items.size() * sizeof(PairItem));
        launch_mfma_pair(nullptr, (const PairItem*)dI.p, nsites, wgs);
        HIPCHK(hipEventRecord(e0, nullptr));
        for (int r = 0; r < reps; ++r) launch_mfma_pair(nullptr, (const PairItem*)dI.p, nsites, wgs);
        HIPCHK(hipEventRecord(e1, nullptr)); HIPCHK(hipEventSynchronize(e1)); HIPCHK(hipEventElapsedTime(&t, e0, e1));
    } else {
        std::vector<PairGram2Item> items(nsites);
        const int spw = 16; int wgs = 0;
        const int npairs = (int)((nslices + spw - 1) / spw), nwg = pair_gram2_group() * ((npairs + 7) / 8);
        DBuf dP((size_t)2 * nsites * nwg * 1024 * 8);
        for (int i = 0; i < nsites; ++i) {
            PairGram2Item& it = items[i]; it.g = g; it.X = (char*)dA.p + (size_t)i * n * 8; it.Y = (char*)dB.p + (size_t)i * n * 8;
            it.Mx = dM.p; it.My = (char*)dM.p + 1024 * 8; it.wg_begin = wgs; it.spw = spw;
            it.partial_y = (char*)dP.p + (size_t)(2 * i) * nwg * 1024 * 8; it.partial_x = (char*)dP.p + (size_t)(2 * i + 1) * nwg * 1024 * 8;
            wgs += nwg;
        }
        DBuf dI(items.size() * sizeof(PairGram2Item)); dI.up(items.data(), items.size() * sizeof(PairGram2Item));
        launch_mfma_pair_gram2(nullptr, (const PairGram2Item*)dI.p, nsites, wgs);
        HIPCHK(hipEventRecord(e0, nullptr));
        for (int r = 0; r < reps; ++r) launch_mfma_pair_gram2(nullptr, (const PairGram2Item*)dI.p, nsites, wgs);
        HIPCHK(hipEventRecord(e1, nullptr)); HIPCHK(hipEventSynchronize(e1)); HIPCHK(hipEventElapsedTime(&t, e0, e1));
    }
    HIPCHK(hipDeviceSynchronize());
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms = (double)t / reps;
}
void dbg_gram_fused(int PA, int K, int PB, const void* X, const void* Y, const void* M, void* out) {
    need_gpu();
    size_t nin = (size_t)PA * K * PB;
    DBuf dX(nin * 8), dY(nin * 8), dM(32 * 32 * 8), dI(sizeof(GramItem)), dR(sizeof(ReduceItem));
    dX.up(X, nin * 8); dY.up(Y, nin * 8); dM.up(M, 32 * 32 * 8);
    GramItem it{}; it.X = dX.p; it.Y = dY.p; it.M = dM.p; it.D = 1; it.PA = PA; it.K = K; it.PB = PB;
    tile_params(PA, PB, 64, it.TA, it.TB, it.nta, it.ntb);
    if (it.TA * it.TB != 64) throw Err(TNQS_ERR_INVALID, "dbg_gram_fused: tiles must hold 64 fibers");
    int ntiles = it.nta * it.ntb; int nch = std::min(3, ntiles);
    it.tiles_per_chunk = (ntiles + nch - 1) / nch; it.nchunks = (ntiles + it.tiles_per_chunk - 1) / it.tiles_per_chunk; it.chunk_begin = 0;
    int npart = 4 * it.nchunks;
    DBuf dP((size_t)npart * K * K * 8), dO((size_t)K * K * 8);
    it.partial = dP.p; dI.up(&it, sizeof(it));
    launch_mfma_gram32_fused(nullptr, (const GramItem*)dI.p, 1, it.nchunks);
    ReduceItem ri{dP.p, dO.p, K * K, npart, 0, 0}; dR.up(&ri, sizeof(ri));
    launch_reduce<float, float>(nullptr, (const ReduceItem*)dR.p, 1, K * K);
    HIPCHK(hipDeviceSynchronize());
    dO.down(out, (size_t)K * K * 8);
}
// the gate path's fused gauge + f64 Gram kernels (kernels_gate.hip) on one site tensor: out[i + KK j] = sum_fibers X'[i, .] conj(X'[j, .]),
// X' = X x_r M with r the lowest leg that is not the bond leg, (i, j) = (s, k); chi_b = 32 (mfma_gauge_gram64_kernel) or 16 (mfma_gauge_gram32_kernel)
void dbg_gauge_gram(int z, const int* chi, int bleg, const void* X, const void* M, void* out) {
    need_gpu();
    if (z < 2 || z > 8 || bleg < 0 || bleg >= z) throw Err(TNQS_ERR_INVALID, "dbg_gauge_gram: bad shape");
    const int rleg = bleg == 0 ? 1 : 0, K = chi[bleg], KK = 2 * K;
    const bool k16 = gauge_gram32_covers(2, z, chi, bleg, rleg), k32 = gauge_gram64_covers(2, z, chi, bleg, rleg);
    if (!k16 && !k32) throw Err(TNQS_ERR_UNSUPPORTED, "dbg_gauge_gram: shape not covered by the fused kernels");
    size_t n = 2; for (int i = 0; i < z; ++i) n *= chi[i];
    long long PA = 1, PB = 1; for (int i = 0; i < bleg; ++i) PA *= chi[i]; for (int i = bleg + 1; i < z; ++i) PB *= chi[i];
    DBuf dX(n * 8), dM((size_t)chi[rleg] * chi[rleg] * 8), dI(sizeof(GramItem)), dR(sizeof(ReduceItem));
    dX.up(X, n * 8); dM.up(M, (size_t)chi[rleg] * chi[rleg] * 8);
    GramItem it{}; it.X = dX.p; it.Y = dX.p; it.M = dM.p; it.D = 2; it.PA = (int)PA; it.K = K; it.PB = (int)PB;
    tile_params(PA, PB, 64, it.TA, it.TB, it.nta, it.ntb);
    if (k16) { it.nta = gauge_gram32_units(z, chi, bleg); it.ntb = 1; }
    const int ntiles = it.nta * it.ntb, nch = std::min(5, ntiles);
    it.tiles_per_chunk = (ntiles + nch - 1) / nch; it.nchunks = (ntiles + it.tiles_per_chunk - 1) / it.tiles_per_chunk; it.chunk_begin = 0;
    const int npart = k16 ? it.nchunks : 2 * it.nchunks;
    DBuf dP((size_t)npart * KK * KK * 16), dO((size_t)KK * KK * 16);
    HIPCHK(hipMemset(dP.p, 0, (size_t)npart * KK * KK * 16));
    it.partial = dP.p; dI.up(&it, sizeof(it));
    if (k16) launch_mfma_gauge_gram32(nullptr, (const GramItem*)dI.p, 1, it.nchunks); else launch_mfma_gauge_gram64(nullptr, (const GramItem*)dI.p, 1, it.nchunks);
    ReduceItem ri{dP.p, dO.p, KK * KK, npart, 0, 0}; dR.up(&ri, sizeof(ri));
    launch_reduce<double, double>(nullptr, (const ReduceItem*)dR.p, 1, KK * KK);
    HIPCHK(hipDeviceSynchronize());
    dO.down(out, (size_t)KK * KK * 16);
}
}  // namespace tnqs
