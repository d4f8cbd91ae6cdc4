// kernels_chi64.hip -- CDNA4 (gfx950) matrix-core kernels for bond dimension 64 (ComplexF32): the per-site shape of BASELINE
// configs[4] (32x32 square lattice, chi = 64: bulk site tensors of 2 x 64^4 elements = 268 MB, theta of 256 x 256).
//
// At chi = 64 every contraction of the path is above the ridge of the machine (a single mode product: 32 flop/B against 19.7), so unlike
// at chi = 16 / 32 nothing here fuses passes -- each kernel has to keep the matrix cores busy on its own:
//   mfma_gram64_kernel        BP message Gram  out[b,b'] = sum X[.,b] conj Y[.,b'],  64 x 64 output (f32 MFMA)
//   mfma_gram128_f64_kernel   gate-path Gram G = psi~^dagger psi~ over the outer legs, 128 x 128 (s, b) output, f64 MFMA accumulation
//   mfma_rowgemm_kernel       mode products / gate epilogue  out[.., n] = sum_k in[.., k] X[k, n]  with K, N up to 128: the tensor operand
//                             goes from global memory straight into MFMA operand registers (no LDS staging of the tensor at all)
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdexcept>
#include <string>
#define TNQS_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) throw std::runtime_error(std::string("HIP kernel launch failed (") + __func__ + "): " + hipGetErrorString(e_)); } while (0)
#include "kernels.hpp"
#include "launch_util.hpp"
#include "mfma_common.hpp"

namespace tnqs {

// ------------------------------------------------------------------------------------------------------------
// Gram (f32 accumulate), KK = D*K <= 64:  partial[4*c + w][i + KK*j] = sum_{rows of chunk c handled by wave w} X[i,row] conj(Y[j,row])
// Tiles of 64 fibers, LDS layout [kk][row] (rows contiguous = memory order); wave w takes 16 rows of every tile and the whole 64 x 64
// output (four 32 x 32 accumulator pairs); the next tile's loads are in flight during the MFMA block.  32 flop/B when X != Y.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mfma_gram64_kernel(const GramItem* __restrict__ items, int nitems) {
    constexpr int TR = 64, TRP = TR + 4, NU = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const Xr = reinterpret_cast<float*>(smem);
    float* const Xi = Xr + 64 * TRP;
    float* const Yr = Xi + 64 * TRP;
    float* const Yi = Yr + 64 * TRP;
    const int tid = threadIdx.x;
    int lo = 0, hi = nitems - 1;
    const int gc = blockIdx.x;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (items[mid].chunk_begin <= gc) lo = mid; else hi = mid - 1; }
    const GramItem it = items[lo];
    const int lc = gc - it.chunk_begin;
    const int D = it.D, K = it.K, TA = it.TA, TB = it.TB, KK = D * K;
    const long long PA = it.PA;
    const bool same = (it.X == it.Y);
    const cf* __restrict__ Xg = reinterpret_cast<const cf*>(it.X);
    const cf* __restrict__ Yg = reinterpret_cast<const cf*>(it.Y);
    const int ntiles = it.nta * it.ntb;
    const int t_begin = lc * it.tiles_per_chunk;
    const int t_end = min(ntiles, t_begin + it.tiles_per_chunk);
    const int lane = tid & 63, w = tid >> 6, ln = lane & 31, h = lane >> 5;
    v16f Cr[2][2], Ci[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { Cr[a][b][r] = 0.f; Ci[a][b][r] = 0.f; }
    for (int e = tid; e < 64 * TRP; e += 256) { Xr[e] = 0.f; Xi[e] = 0.f; Yr[e] = 0.f; Yi[e] = 0.f; }
    const TileMap m = make_map(tid, D, TA, TB, PA, K);
    const long long kstride = (long long)D * PA;
    const bool fast = m.U <= 256 && (K + m.KP - 1) / m.KP <= NU;
    v4f px[NU], py[NU];
    auto tile_origin = [&](int t, int& a0, int& b0, int& na, int& nb) {
        int ta = t % it.nta, tb = t / it.nta;
        a0 = ta * TA; b0 = tb * TB; na = min(TA, it.PA - a0); nb = min(TB, it.PB - b0);
    };
    auto issue_loads = [&](int t) {
        int a0, b0, na, nb; tile_origin(t, a0, b0, na, nb);
        const long long org = (long long)D * (a0 + PA * (long long)K * b0);
        const bool v0 = m.active && m.al < na && m.bl < nb, v1 = v0 && m.al1 < na;
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            int k = m.kp + m.KP * j;
            v4f vx, vy; vx[0] = vx[1] = vx[2] = vx[3] = 0.f; vy = vx;
            if (k < K && v0) {
                const long long o = org + m.off + kstride * k;
                if (m.vec == 2 && v1) { vx = *reinterpret_cast<const v4f*>(Xg + o); vy = same ? vx : *reinterpret_cast<const v4f*>(Yg + o); }
                else { cf x = Xg[o]; vx[0] = x.re; vx[1] = x.im; if (same) vy = vx; else { cf y = Yg[o]; vy[0] = y.re; vy[1] = y.im; } }
            }
            px[j] = vx; py[j] = vy;
        }
    };
    auto commit_loads = [&]() {
        if (!m.active) return;
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            int k = m.kp + m.KP * j;
            if (k < K) {
                int o0 = (m.c0 + D * k) * TRP + m.row0;
                Xr[o0] = px[j][0]; Xi[o0] = px[j][1]; Yr[o0] = py[j][0]; Yi[o0] = py[j][1];
                if (m.vec == 2) { int o1 = (m.c1 + D * k) * TRP + m.row1; Xr[o1] = px[j][2]; Xi[o1] = px[j][3]; Yr[o1] = py[j][2]; Yi[o1] = py[j][3]; }
            }
        }
    };
    if (fast && t_begin < t_end) issue_loads(t_begin);
    for (int t = t_begin; t < t_end; ++t) {
        lds_barrier();
        if (fast) commit_loads();       // invalid cells were loaded as zeros, so edge tiles need no extra clearing
        else {
            int a0, b0, na, nb; tile_origin(t, a0, b0, na, nb);
            const int ntile_el = D * TA * K * TB;
            for (int e = tid; e < ntile_el; e += 256) {
                int s = e % D; int r1 = e / D; int al = r1 % TA; int r2 = r1 / TA; int k = r2 % K; int bl = r2 / K;
                cf vx, vy; vx.re = vx.im = vy.re = vy.im = 0.f;
                if (al < na && bl < nb) {
                    long long off = s + D * ((long long)(a0 + al) + PA * ((long long)k + (long long)K * (b0 + bl)));
                    vx = Xg[off]; vy = same ? vx : Yg[off];
                }
                int o = (s + D * k) * TRP + (al + TA * bl);
                Xr[o] = vx.re; Xi[o] = vx.im; Yr[o] = vy.re; Yi[o] = vy.im;
            }
        }
        lds_barrier();
        if (fast && t + 1 < t_end) issue_loads(t + 1);
        // wave w: rows 16w .. 16w+15; lane half h takes rows 16w + 8h + q
        float yr[2][8], yi[2][8];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ro = (32 * b + ln) * TRP + 16 * w + 8 * h;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                v4f t2 = *reinterpret_cast<const v4f*>(Yr + ro + 4 * q), t3 = *reinterpret_cast<const v4f*>(Yi + ro + 4 * q);
#pragma unroll
                for (int c = 0; c < 4; ++c) { yr[b][4 * q + c] = t2[c]; yi[b][4 * q + c] = t3[c]; }
            }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float xr[8], xi[8];
            const int ro = (32 * a + ln) * TRP + 16 * w + 8 * h;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                v4f t0 = *reinterpret_cast<const v4f*>(Xr + ro + 4 * q), t1 = *reinterpret_cast<const v4f*>(Xi + ro + 4 * q);
#pragma unroll
                for (int c = 0; c < 4; ++c) { xr[4 * q + c] = t0[c]; xi[4 * q + c] = t1[c]; }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    // out[i][j] += x[i] * conj(y[j])
                    Cr[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[q], yr[b][q], Cr[a][b], 0, 0, 0);
                    Ci[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xi[q], yr[b][q], Ci[a][b], 0, 0, 0);
                    Cr[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xi[q], yi[b][q], Cr[a][b], 0, 0, 0);
                    Ci[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(-xr[q], yi[b][q], Ci[a][b], 0, 0, 0);
                }
            }
        }
    }
    cf* __restrict__ part = reinterpret_cast<cf*>(it.partial) + (size_t)(4 * lc + w) * KK * KK;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int i = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h, j = 32 * b + ln;
                if (i < KK && j < KK) { cf v; v.re = Cr[a][b][r]; v.im = Ci[a][b][r]; part[i + (size_t)KK * j] = v; }
            }
}
bool launch_mfma_gram64(hipStream_t s, const GramItem* d_items, int nitems, int total_chunks, int KKmax) {
    if (KKmax > 64) return false;
    if (total_chunks <= 0) return true;
    const size_t lds = (size_t)4 * 64 * 68 * sizeof(float);
    set_max_dynamic_lds((const void*)mfma_gram64_kernel, lds);
    hipLaunchKernelGGL(mfma_gram64_kernel, dim3(total_chunks), dim3(256), lds, s, d_items, nitems); TNQS_CHECK_LAUNCH();
    return true;
}

// ------------------------------------------------------------------------------------------------------------
// theta SVD for matrices that do not fit the LDS-resident one-sided Jacobi (256 x 128 at chi = 64): Cholesky-QR preprocessing.
//   G = A^dagger A + delta I (f64, n x n)  ->  G = L L^dagger (chol_packed_kernel, shift delta)  ->  R = L^dagger (n x n, f32)
//   one-sided Jacobi on R in LDS: R J = U_R Sigma_R            (jacobi_lds_kernel<float>)
//   J = R^dagger (U_R Sigma_R) Sigma_R^-2                      (recover_v: the rotations are never accumulated in f32)
//   A <- A J = U Sigma of A                                    (small_cgemm: the same rotations orthogonalise the columns of A, because
//                                                               R^dagger R and A^dagger A have the same eigenvectors; the shift only
//                                                               bounds the condition number of R, it cancels in R^-1 (R J) = J)
// The kernels below are the three small pieces around the existing ones.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tall_gram_kernel(const TallSvdItem* __restrict__ items) {
    // one workgroup per (item, 32 x 32 tile (I <= J) of G); rows staged 64 at a time, [col][row] in LDS
    const TallSvdItem it = items[blockIdx.x];
    const int m = it.m, n = it.n, nt = (n + 31) >> 5;
    int I = 0, rem = blockIdx.y; while (I < nt && rem >= nt - I) { rem -= nt - I; ++I; }
    if (I >= nt) return;
    const int J = I + rem;
    __shared__ float Ar[32][65], Ai[32][65], Br[32][65], Bi[32][65];
    const cf* __restrict__ A = reinterpret_cast<const cf*>(it.A);
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;           // thread -> outputs (2 tx + {0,1}, 2 ty + {0,1}) of the tile
    double cr[2][2] = {{0, 0}, {0, 0}}, ci[2][2] = {{0, 0}, {0, 0}};
    for (int r0 = 0; r0 < m; r0 += 64) {
        for (int e = tid; e < 32 * 64; e += 256) {
            const int r = e & 63, c = e >> 6;
            cf a = {0.f, 0.f}, b = {0.f, 0.f};
            if (r0 + r < m) { if (32 * I + c < n) a = A[(r0 + r) + (size_t)m * (32 * I + c)]; if (32 * J + c < n) b = A[(r0 + r) + (size_t)m * (32 * J + c)]; }
            Ar[c][r] = a.re; Ai[c][r] = a.im; Br[c][r] = b.re; Bi[c][r] = b.im;
        }
        __syncthreads();
        for (int r = 0; r < 64; ++r) {
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const double ar = Ar[2 * tx + p][r], ai = Ai[2 * tx + p][r], br = Br[2 * ty + q][r], bi = Bi[2 * ty + q][r];
                    cr[p][q] += ar * br + ai * bi; ci[p][q] += ar * bi - ai * br;          // conj(a) b
                }
        }
        __syncthreads();
    }
    struct alignas(16) cd { double re, im; };
    cd* __restrict__ G = reinterpret_cast<cd*>(it.G);
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = 32 * I + 2 * tx + p, j = 32 * J + 2 * ty + q;
            if (i < n && j < n) { G[i + (size_t)n * j] = cd{cr[p][q], ci[p][q]}; if (I != J) G[j + (size_t)n * i] = cd{cr[p][q], -ci[p][q]}; }
        }
}
__global__ __launch_bounds__(256) void tall_rt_kernel(const TallSvdItem* __restrict__ items) {
    const TallSvdItem it = items[blockIdx.x];
    const int n = it.n;
    struct alignas(16) cd { double re, im; };
    const cd* __restrict__ L = reinterpret_cast<const cd*>(it.L);
    cf* __restrict__ R0 = reinterpret_cast<cf*>(it.R0); cf* __restrict__ Rr = reinterpret_cast<cf*>(it.Rrot);
    for (int e = threadIdx.x; e < n * n; e += 256) {
        const int i = e % n, j = e / n;                      // R[i, j] = conj(L[j, i]) for i <= j
        cf v = {0.f, 0.f};
        if (i <= j) { const cd l = L[j + (size_t)n * i]; v.re = (float)l.re; v.im = (float)(-l.im); }
        R0[e] = v; Rr[e] = v;
    }
}
// C (m x n) = A (m x k) B (k x n), ComplexF32 column-major, one wave per 32 x 32 tile of C, operands straight from L2 (all <= 512 KiB)
__global__ __launch_bounds__(256) void small_cgemm_kernel(const SmallGemmItem* __restrict__ items) {
    const SmallGemmItem it = items[blockIdx.x];
    const cf* __restrict__ A = reinterpret_cast<const cf*>(it.A);
    const cf* __restrict__ B = reinterpret_cast<const cf*>(it.B);
    cf* __restrict__ C = reinterpret_cast<cf*>(it.C);
    const int m = it.m, n = it.n, k = it.k;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, ln = lane & 31, h = lane >> 5;
    const int mt = (m + 31) >> 5, ntl = (n + 31) >> 5;
    const int tile = blockIdx.y * 4 + w;
    if (tile >= mt * ntl) return;
    const int r0 = 32 * (tile % mt), c0 = 32 * (tile / mt);
    const int row = r0 + ln, col = c0 + ln;
    const bool okr = row < m, okc = col < n;
    const cf* pa = A + min(row, m - 1);                       // A[row][kk] at row + m kk
    const cf* pb = B + (size_t)k * min(col, n - 1);          // B[kk][col] at kk + k col
    v16f Cr, Ci;
#pragma unroll
    for (int r = 0; r < 16; ++r) { Cr[r] = 0.f; Ci[r] = 0.f; }
    for (int k0 = 0; k0 < k; k0 += 2) {
        const int kk = k0 + h;
        cf a = {0.f, 0.f}, b = {0.f, 0.f};
        if (kk < k) { if (okr) a = pa[(size_t)m * kk]; if (okc) b = pb[kk]; }
        Cr = __builtin_amdgcn_mfma_f32_32x32x2f32(a.re, b.re, Cr, 0, 0, 0);
        Ci = __builtin_amdgcn_mfma_f32_32x32x2f32(a.re, b.im, Ci, 0, 0, 0);
        Cr = __builtin_amdgcn_mfma_f32_32x32x2f32(-a.im, b.im, Cr, 0, 0, 0);
        Ci = __builtin_amdgcn_mfma_f32_32x32x2f32(a.im, b.re, Ci, 0, 0, 0);
    }
    if (okc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = r0 + (r & 3) + 8 * (r >> 2) + 4 * h;      // C[row = i][col = ln]
            if (i < m) { cf v; v.re = Cr[r]; v.im = Ci[r]; C[i + (size_t)m * col] = v; }
        }
    }
}
__global__ __launch_bounds__(256) void copy_items_kernel(const CopyItem* __restrict__ items) {
    const CopyItem it = items[blockIdx.x];
    const v4f* __restrict__ src = reinterpret_cast<const v4f*>(it.src); v4f* __restrict__ dst = reinterpret_cast<v4f*>(it.dst);
    for (size_t e = blockIdx.y * 256 + threadIdx.x; e < it.n16; e += (size_t)gridDim.y * 256) dst[e] = src[e];
}
void launch_tall_gram(hipStream_t s, const TallSvdItem* d_items, int nitems, int nmax) {
    if (nitems <= 0) return;
    const int nt = (nmax + 31) / 32;
    hipLaunchKernelGGL(tall_gram_kernel, dim3(nitems, nt * (nt + 1) / 2), dim3(256), 0, s, d_items); TNQS_CHECK_LAUNCH();
}
void launch_tall_rt(hipStream_t s, const TallSvdItem* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL(tall_rt_kernel, dim3(nitems), dim3(256), 0, s, d_items); TNQS_CHECK_LAUNCH();
}
void launch_small_cgemm(hipStream_t s, const SmallGemmItem* d_items, int nitems, int mmax, int nmax) {
    if (nitems <= 0) return;
    const int tiles = ((mmax + 31) / 32) * ((nmax + 31) / 32);
    hipLaunchKernelGGL(small_cgemm_kernel, dim3(nitems, (tiles + 3) / 4), dim3(256), 0, s, d_items); TNQS_CHECK_LAUNCH();
}
void launch_copy_items(hipStream_t s, const CopyItem* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL(copy_items_kernel, dim3(nitems, 16), dim3(256), 0, s, d_items); TNQS_CHECK_LAUNCH();
}

}  // namespace tnqs
