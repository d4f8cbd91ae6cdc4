// kernels_f64.hip -- ComplexF64 mode products on the f64 matrix cores (v_mfma_f64_16x16x4_f64).
//
// ComplexF64 is the reference's default element type (README.md:86).  Its mode products ran on the generic LDS-tiled vector kernel
// (fiber_gemm_kernel<double>: 13 TFLOP/s, 1.6 TB/s at chi = 32); one product moves 32 bytes per 8 K flops, so at chi = 32 it is HBM-bound
// on the matrix cores (ridge of the f64 pipe: 78.6 TFLOP/s / 8 TB/s ~ 10 flop/B, the product has 8).
//
//      out[a, n, b] = sum_k in[a, k, b] X[k, n]          element (a, k, b) at a + PA (k + K b),  X[k, n] at k + K n
//
// computed TRANSPOSED, one wave per tile of 16 fibers:  C[n][fiber] = sum_k X^T[n][k] in[fiber][k]
//      A operand [i = n = lane & 15][k = lane >> 4]   from LDS: X^T staged once per workgroup, odd row pitch (conflict-free b128 reads)
//      B operand [k = lane >> 4][j = fiber = lane & 15] STRAIGHT from global memory into registers: 16 lanes = 16 consecutive a = 256 bytes
//      C [row = n = (lane >> 4) + 4 r][col = fiber = lane & 15]: stored along the lanes, 256-byte runs again
// The 16 fibers of a tile are (al, bl) = (l % TA, l / TA) with TA = min(PA, 16): on the first leg (PA = 2) a tile takes 8 values of b.
// Four real products per complex one: the kernel is memory-bound, the f64 pipe runs at a quarter of its rate.
// The next tile's operand is loaded before the products of the current one (double register set).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdexcept>
#include <string>
#define TNQS_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) throw std::runtime_error(std::string("HIP kernel launch failed (") + __func__ + "): " + hipGetErrorString(e_)); } while (0)
#include "kernels.hpp"
#include "mfma_common.hpp"
#include "launch_util.hpp"

namespace tnqs {

struct alignas(16) zc { double re, im; };
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ zc ldgz(const zc* p) { const v2d t = *(const v2d TNQS_AS1*)(p); zc r; r.re = t[0]; r.im = t[1]; return r; }
__device__ __forceinline__ void stgz(zc* p, zc v) { v2d t = {v.re, v.im}; *(v2d TNQS_AS1*)(p) = t; }
__device__ __forceinline__ double block_sum_f64(double v, double* sh /* >= 17 doubles */) {      // 256 threads
    v = wave_sum_d(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// General form (the gate epilogue contracts the site index too):  out[(s', n), (a, b)] = sum_{(s, k)} in[(s, k), (a, b)] X[(s, k), (s', n)],
// in at s + D (a + PA (k + K b)), out at s' + Do (a + PA (n + No b)), X at kk + KK nn with kk = s + D k, nn = s' + Do n.  For D = 2 the lanes
// kq = 0, 1 of a k-step read s = 0, 1 of the same fiber: 32 contiguous bytes per fiber, 512 per 16 fibers.
template <int NBLK, int KS, bool GEN>       // NN <= 16 NBLK, KK <= 4 KS; GEN = false: plain mode product (D = Do = 1, no norm partial), fewer registers
__global__ __launch_bounds__(256) void mfma_fiber_gemm_f64_kernel(const FiberItem* __restrict__ items, int nitems, double* __restrict__ norm_partials) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double sh_red[17];
    zc* const XT = reinterpret_cast<zc*>(smem);                 // X^T[nn][kk] at nn * KP + kk
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    int lo = 0, hi = nitems - 1;
    const int gw = blockIdx.x;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (items[mid].tile_begin <= gw) lo = mid; else hi = mid - 1; }
    const FiberItem it = items[lo];
    const int D = GEN ? it.D : 1, Do = GEN ? it.Do : 1, K = it.K, No = it.No, KK = D * K, NN = Do * No;
    const int KP = ((KK + 3) & ~3) + 1;                          // row pitch: whole k-steps + 1 (odd: the 16 rows of a b128 read fall into different banks)
    const long long PA = it.PA, PB = it.PB;
    const zc* __restrict__ in = reinterpret_cast<const zc*>(it.in);
    zc* __restrict__ out = reinterpret_cast<zc*>(it.out);
    const zc* __restrict__ X = reinterpret_cast<const zc*>(it.X);
    { zc z; z.re = 0; z.im = 0; for (int e = tid; e < 16 * NBLK * KP; e += 256) XT[e] = z; }       // pad rows / columns the last blocks read
    lds_barrier();
    for (int e = tid; e < KK * NN; e += 256) { const int kk = e % KK, nn = e / KK; XT[nn * KP + kk] = ldgz(X + e); }
    const int TA = it.TA, TB = it.TB;                            // TA * TB = 16 fibers per tile
    const int al = l15 % TA, bl = l15 / TA;
    const int ntiles = it.nta * it.ntb;
    const int t_begin = (gw - it.tile_begin) * it.tpw, t_end = min(ntiles, t_begin + it.tpw);
    double nrm = 0;
    zc bv[2][KS];
    auto tile_fiber = [&](int t, long long& ibase, long long& obase, bool& valid) {
        const int ta = t % it.nta, tb = t / it.nta;
        const long long a = (long long)ta * TA + al, b = (long long)tb * TB + bl;
        valid = a < PA && b < PB;
        ibase = D * (a + PA * (long long)K * b); obase = Do * (a + PA * (long long)No * b);
    };
    auto load_tile = [&](int t, zc (&dst)[KS]) {
        long long ibase, obase; bool valid; tile_fiber(t, ibase, obase, valid);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kk = 4 * ks + kq;
            zc v; v.re = 0; v.im = 0;
            if (valid && kk < KK) v = ldgz(in + ibase + (D == 1 ? PA * kk : (kk % D) + (long long)D * PA * (kk / D)));       // (D == 1: uniform fast path, no division)
            dst[ks] = v;
        }
    };
    // one tile: prefetch the operand of the wave's next tile into `nxt`, multiply `cur`
    auto process = [&](int t, const zc (&cur)[KS], zc (&nxt)[KS]) {
        if (t + 4 < t_end) load_tile(t + 4, nxt);
        long long ibase, obase; bool valid; tile_fiber(t, ibase, obase, valid);
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) {
            if (16 * nb >= NN) break;                            // (uniform)
            v4d cr = {0, 0, 0, 0}, ci = {0, 0, 0, 0};
            const zc* xr = XT + (16 * nb + l15) * KP + kq;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (4 * ks >= KK) break;                         // (uniform)
                const zc a = xr[4 * ks];
                const zc b = cur[ks];
                cr = __builtin_amdgcn_mfma_f64_16x16x4f64(a.re, b.re, cr, 0, 0, 0);
                cr = __builtin_amdgcn_mfma_f64_16x16x4f64(-a.im, b.im, cr, 0, 0, 0);
                ci = __builtin_amdgcn_mfma_f64_16x16x4f64(a.re, b.im, ci, 0, 0, 0);
                ci = __builtin_amdgcn_mfma_f64_16x16x4f64(a.im, b.re, ci, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nn = 16 * nb + kq + 4 * r;
                if (valid && nn < NN) {
                    zc o; o.re = cr[r]; o.im = ci[r];
                    stgz(out + obase + (Do == 1 ? PA * nn : (nn % Do) + (long long)Do * PA * (nn / Do)), o);
                    if (GEN) nrm += o.re * o.re + o.im * o.im;
                }
            }
        }
    };
    lds_barrier();                                               // X^T staged
    int t = t_begin + w;
    if (t < t_end) load_tile(t, bv[0]);
    for (; t < t_end; t += 8) {
        process(t, bv[0], bv[1]);
        if (t + 4 < t_end) process(t + 4, bv[1], bv[0]);
    }
    if (GEN && it.want_norm) {                                   // (uniform per workgroup) sum |out|^2 of this workgroup's tiles
        const double tsum = block_sum_f64(nrm, sh_red);
        if (tid == 0) norm_partials[gw] = tsum;
    }
}

// shapes the kernel takes: contracted and produced index up to 64 each (X^T within the LDS, accumulators within the registers)
bool fiber_gemm_f64_covers(const FiberItem& it) { return it.D * it.K >= 4 && it.D * it.K <= 64 && it.Do * it.No >= 1 && it.Do * it.No <= 64; }
// tile grid of an item: TA x TB = 16 fibers
void fiber_gemm_f64_tiles(FiberItem& it) {
    it.TA = it.PA >= 16 ? 16 : (it.PA >= 8 ? 8 : (it.PA >= 4 ? 4 : (it.PA >= 2 ? 2 : 1)));
    it.TB = 16 / it.TA;
    it.nta = (it.PA + it.TA - 1) / it.TA; it.ntb = (it.PB + it.TB - 1) / it.TB;
}
template <int NBLK, int KS, bool GEN> static void launch_one(hipStream_t s, const FiberItem* d_items, int nitems, int total_wgs, size_t lds, double* np) {
    set_max_dynamic_lds((const void*)mfma_fiber_gemm_f64_kernel<NBLK, KS, GEN>, lds);
    hipLaunchKernelGGL((mfma_fiber_gemm_f64_kernel<NBLK, KS, GEN>), dim3(total_wgs), dim3(256), lds, s, d_items, nitems, np);
}
// Kmax / Nmax: largest contracted / produced index (D K, Do No) among the items; d_norm_partials: one double per workgroup (FiberItem::want_norm);
// general: some item has D != 1, Do != 1 or wants its norm
void launch_mfma_fiber_gemm_f64(hipStream_t s, const FiberItem* d_items, int nitems, int total_wgs, int Kmax, int Nmax, double* d_norm_partials, bool general) {
    if (total_wgs <= 0) return;
    const int nblk = Nmax <= 16 ? 1 : (Nmax <= 32 ? 2 : 4), ks = Kmax <= 16 ? 4 : (Kmax <= 32 ? 8 : 16);
    const size_t lds = (size_t)16 * nblk * (((Kmax + 3) & ~3) + 1) * sizeof(zc);
#define TNQS_F64(NB, KSV) if (nblk == NB && ks == KSV) { if (general) launch_one<NB, KSV, true>(s, d_items, nitems, total_wgs, lds, d_norm_partials); else launch_one<NB, KSV, false>(s, d_items, nitems, total_wgs, lds, d_norm_partials); }
    TNQS_F64(1, 4) TNQS_F64(1, 8) TNQS_F64(1, 16) TNQS_F64(2, 4) TNQS_F64(2, 8) TNQS_F64(2, 16) TNQS_F64(4, 4) TNQS_F64(4, 8) TNQS_F64(4, 16)
#undef TNQS_F64
    TNQS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// Gram over the fibers for ComplexF64 operands:  partial[i + KK j] = sum_f X[i, f] conj(Y[j, f]),  (i, j) = (s, k) columns, KK = D K <= 64
// (BP messages: X = the absorbed product, Y = psi; gate path: X = Y = the gauged tensor).  Tiles of 32 fibers x KK columns of X and Y go
// through LDS (coalesced loads along the fibers, element (column c, fiber f) at c * 33 + f: the 16 rows of an operand read fall into
// different banks), double-buffered with a register prefetch: one barrier per tile.  The 16 x 16 output blocks are dealt to the four
// waves; a block is 8 k-steps of four real products per tile.  Memory-bound like the product (64 KiB per tile against 512 matrix instructions).
// X == Y: only the upper block triangle is computed, the lower one is mirrored when the partial is written.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mfma_gram_f64in_kernel(const GramItem* __restrict__ items, int nitems) {
    constexpr int TF = 32, TFP = TF + 1, MAXC = 64, NE = 8;      // fibers per tile, pitch, columns, elements per thread and tensor
    extern __shared__ __attribute__((aligned(16))) char smem[];
    zc* const Lb = reinterpret_cast<zc*>(smem);                  // [buf 2][tensor 2][MAXC * TFP]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    int lo = 0, hi = nitems - 1;
    const int gc = blockIdx.x;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (items[mid].chunk_begin <= gc) lo = mid; else hi = mid - 1; }
    const GramItem it = items[lo];
    const int lc = gc - it.chunk_begin;
    const int D = it.D, K = it.K, TA = it.TA, TB = it.TB, KK = D * K;
    const long long PA = it.PA, PB = it.PB;
    const zc* __restrict__ Xg = reinterpret_cast<const zc*>(it.X);
    const zc* __restrict__ Yg = reinterpret_cast<const zc*>(it.Y);
    const bool same = it.X == it.Y;
    const int ntiles = it.nta * it.ntb;
    const int t_begin = lc * it.tiles_per_chunk, t_end = min(ntiles, t_begin + it.tiles_per_chunk);
    const int nb = (KK + 15) >> 4;
    // blocks of this wave: idx = w + 4 q over the block list (all nb^2, or the upper triangle when X == Y)
    const int nblk = same ? nb * (nb + 1) / 2 : nb * nb;
    int bI[4], bJ[4]; bool bOn[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = w + 4 * q; bOn[q] = idx < nblk;
        int I = 0, J = 0;
        if (bOn[q]) { if (same) { int rem = idx; while (rem >= nb - I) { rem -= nb - I; ++I; } J = I + rem; } else { I = idx % nb; J = idx / nb; } }
        bI[q] = I; bJ[q] = J;
    }
    v4d cr[4], ci[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) { cr[q][r] = 0.0; ci[q][r] = 0.0; }
    { zc z; z.re = 0; z.im = 0; for (int e = tid; e < 4 * MAXC * TFP; e += 256) Lb[e] = z; }       // columns >= KK and pad rows stay zero
    // element e of a tile: s + D (al + TA (k + K bl)); thread takes e = tid + 256 u
    const int ntile_el = D * TA * K * TB;
    zc px[NE], py[NE];
    auto tile_origin = [&](int t, long long& a0, long long& b0) { const int ta = t % it.nta, tb = t / it.nta; a0 = (long long)ta * TA; b0 = (long long)tb * TB; };
    auto issue = [&](int t) {
        long long a0, b0; tile_origin(t, a0, b0);
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int e = tid + 256 * u;
            zc vx; vx.re = 0; vx.im = 0; zc vy = vx;
            if (e < ntile_el) {
                const int s = e % D; const int r1 = e / D; const int al = r1 % TA; const int r2 = r1 / TA; const int k = r2 % K; const int bl = r2 / K;
                if (a0 + al < PA && b0 + bl < PB) {
                    const long long o = s + D * ((a0 + al) + PA * ((long long)k + (long long)K * (b0 + bl)));
                    vx = ldgz(Xg + o); if (!same) vy = ldgz(Yg + o);
                }
            }
            px[u] = vx; py[u] = vy;
        }
    };
    auto commit = [&](int buf) {
        zc* Xl = Lb + buf * (2 * MAXC * TFP); zc* Yl = Xl + MAXC * TFP;
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int e = tid + 256 * u;
            if (e < ntile_el) {
                const int s = e % D; const int r1 = e / D; const int al = r1 % TA; const int r2 = r1 / TA; const int k = r2 % K; const int bl = r2 / K;
                const int o = (s + D * k) * TFP + (al + TA * bl);
                Xl[o] = px[u]; if (!same) Yl[o] = py[u];
            }
        }
    };
    lds_barrier();
    if (t_begin < t_end) { issue(t_begin); commit(0); if (t_begin + 1 < t_end) issue(t_begin + 1); }
    lds_barrier();
    for (int t = t_begin; t < t_end; ++t) {
        const int cur = (t - t_begin) & 1;
        if (t + 1 < t_end) { commit(cur ^ 1); if (t + 2 < t_end) issue(t + 2); }
        const zc* Xl = Lb + cur * (2 * MAXC * TFP); const zc* Yl = same ? Xl : Xl + MAXC * TFP;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!bOn[q]) continue;                               // (wave-uniform)
            const zc* pa = Xl + (16 * bI[q] + l15) * TFP + kq;
            const zc* pb = Yl + (16 * bJ[q] + l15) * TFP + kq;
#pragma unroll
            for (int ks = 0; ks < TF / 4; ++ks) {
                const zc a = pa[4 * ks], b = pb[4 * ks];
                cr[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.re, b.re, cr[q], 0, 0, 0);       // a conj(b)
                cr[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.im, b.im, cr[q], 0, 0, 0);
                ci[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.im, b.re, ci[q], 0, 0, 0);
                ci[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a.re, b.im, ci[q], 0, 0, 0);
            }
        }
        lds_barrier();                                           // tile t consumed, tile t + 1 committed
    }
    zc* __restrict__ part = reinterpret_cast<zc*>(it.partial) + (size_t)lc * KK * KK;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (!bOn[q]) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * bI[q] + kq + 4 * r, j = 16 * bJ[q] + l15;
            if (i < KK && j < KK) {
                zc v; v.re = cr[q][r]; v.im = ci[q][r];
                part[i + (size_t)KK * j] = v;
                if (same && bI[q] != bJ[q]) { zc c; c.re = v.re; c.im = -v.im; part[j + (size_t)KK * i] = c; }
            }
        }
    }
}
// D K <= 64 columns, tiles of 32 fibers with at most 8 elements per thread and tensor
bool gram_f64in_covers(int D, int K) { return D * K >= 4 && D * K <= 64; }
void launch_mfma_gram_f64in(hipStream_t s, const GramItem* d_items, int nitems, int total_chunks) {
    if (total_chunks <= 0) return;
    const size_t lds = (size_t)4 * 64 * 33 * sizeof(zc);
    set_max_dynamic_lds((const void*)mfma_gram_f64in_kernel, lds);
    hipLaunchKernelGGL(mfma_gram_f64in_kernel, dim3(total_chunks), dim3(256), lds, s, d_items, nitems); TNQS_CHECK_LAUNCH();
}

}  // namespace tnqs
