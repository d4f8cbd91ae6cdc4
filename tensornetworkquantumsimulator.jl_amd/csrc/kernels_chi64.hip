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
// Gram (f32 accumulate), KK = D*K <= 64:  partial[c][i + KK*j] = sum_{rows of chunk c} X[i,row] conj(Y[j,row])
// Tiles of 64 fibers, LDS layout [kk][row] (rows contiguous = memory order); wave w takes 16 rows of every tile and the whole 64 x 64
// output (four 32 x 32 accumulator pairs); the next tile's loads are in flight during the MFMA block.  32 flop/B when X != Y.
// ------------------------------------------------------------------------------------------------------------
template <bool M3>
__global__ __launch_bounds__(256, 2) void mfma_gram64_kernel(const GramItem* __restrict__ items, int nitems) {
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
    // wave w: output rows block a = w & 1 (i in [32 a, 32 a + 32)) x all 64 columns, tile rows 32 (w >> 1) .. + 32 -- 64 accumulator
    // registers per wave instead of 128, so that TWO workgroups fit a CU (one's barriers and LDS commits hide behind the other's MFMAs)
    const int a = w & 1, rh = w >> 1;
    CAcc32<M3> C[2];                                             // M3: three-multiplication product (mfma_common.hpp)
    C[0].zero(); C[1].zero();
    for (int e = tid; e < 64 * TRP; e += 256) { Xr[e] = 0.f; Xi[e] = 0.f; Yr[e] = 0.f; Yi[e] = 0.f; }
    const TileMap m = make_map(tid, D, TA, TB, PA, K);
    const long long kstride = (long long)D * PA;
    const bool fast = m.U <= 256 && (K + m.KP - 1) / m.KP <= NU;
    v4f px[NU], py[NU];
    auto tile_origin = [&](int t, int& a0, int& b0, int& na, int& nb) {
        int ta = t % it.nta, tb = t / it.nta;
        a0 = ta * TA; b0 = tb * TB; na = min(TA, it.PA - a0); nb = min(TB, it.PB - b0);
    };
    // full tiles of the usual shape (every thread owns NU two-element units): straight-line loads.  The guarded path below compiles
    // into one branch per load with an s_waitcnt vmcnt(0) in front of the next one, i.e. the NU loads of a tile are SERIALISED (seen in the
    // ISA; a 64-row tile then costs NU memory latencies, ~10 us instead of the 3.4 us of its matrix work)
    const bool straight = fast && m.active && m.vec == 2 && K == m.KP * NU;
    auto issue_loads = [&](int t) {
        int a0, b0, na, nb; tile_origin(t, a0, b0, na, nb);
        const long long org = (long long)D * (a0 + PA * (long long)K * b0);
        if (straight && na == TA && nb == TB) {
            const cf* px0 = Xg + org + m.off + kstride * m.kp; const cf* py0 = Yg + org + m.off + kstride * m.kp;
            const long long st = kstride * m.KP;
            if (same) {
#pragma unroll
                for (int j = 0; j < NU; ++j) { px[j] = ldg4(px0 + st * j); py[j] = px[j]; }
            } else {
#pragma unroll
                for (int j = 0; j < NU; ++j) { px[j] = ldg4(px0 + st * j); py[j] = ldg4(py0 + st * j); }
            }
            return;
        }
        const bool v0 = m.active && m.al < na && m.bl < nb, v1 = v0 && m.al1 < na;
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            int k = m.kp + m.KP * j;
            v4f vx, vy; vx[0] = vx[1] = vx[2] = vx[3] = 0.f; vy = vx;
            if (k < K && v0) {
                const long long o = org + m.off + kstride * k;
                if (m.vec == 2 && v1) { vx = ldg4(Xg + o); vy = same ? vx : ldg4(Yg + o); }
                else { cf x = ldgc(Xg + o); vx[0] = x.re; vx[1] = x.im; if (same) vy = vx; else { cf y = ldgc(Yg + o); vy[0] = y.re; vy[1] = y.im; } }
            }
            px[j] = vx; py[j] = vy;
        }
    };
    auto commit_loads = [&]() {                  // invalid cells were loaded as zeros, so edge tiles need no extra clearing
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
        if (fast) commit_loads();
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
        // rows 32 rh .. 32 rh + 31 in two halves of 16 (lane half h takes 8 of them): bounds the operand registers
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int r0 = 32 * rh + 16 * hf + 8 * h;
            float xr[8], xi[8], yr[2][8], yi[2][8];
            {
                const int ro = (32 * a + ln) * TRP + r0;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    v4f t0 = *reinterpret_cast<const v4f*>(Xr + ro + 4 * q), t1 = *reinterpret_cast<const v4f*>(Xi + ro + 4 * q);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { xr[4 * q + c] = t0[c]; xi[4 * q + c] = t1[c]; }
                }
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int ro = (32 * b + ln) * TRP + r0;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    v4f t2 = *reinterpret_cast<const v4f*>(Yr + ro + 4 * q), t3 = *reinterpret_cast<const v4f*>(Yi + ro + 4 * q);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { yr[b][4 * q + c] = t2[c]; yi[b][4 * q + c] = t3[c]; }
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    C[b].mac_conj(xr[q], xi[q], yr[b][q], yi[b][q]);       // out[i][j] += x[i] * conj(y[j])
                }
            }
        }
    }
    // one partial per chunk: the two row halves (waves w, w + 2) are summed through the free tile buffers
    lds_barrier();
    C[0].finish_conj(); C[1].finish_conj();
    v2f* const R = reinterpret_cast<v2f*>(smem);                // [rh][j][i], pitch 65: 2 * 64 * 65 * 8 B = 66.5 KB
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h, j = 32 * b + ln;
            v2f v = {C[b].a[r], C[b].b[r]};
            R[(rh * 64 + j) * 65 + i] = v;
        }
    lds_barrier();
    cf* __restrict__ part = reinterpret_cast<cf*>(it.partial) + (size_t)lc * KK * KK;
    for (int e = tid; e < KK * KK; e += 256) {
        const int i = e % KK, j = e / KK;
        const v2f u = R[j * 65 + i], v = R[(64 + j) * 65 + i];
        cf o; o.re = u[0] + v[0]; o.im = u[1] + v[1]; part[e] = o;
    }
}
bool launch_mfma_gram64(hipStream_t s, const GramItem* d_items, int nitems, int total_chunks, int KKmax) {
    if (KKmax > 64) return false;
    if (total_chunks <= 0) return true;
    if (mfma_use_x3()) return launch_x3_gram64(s, d_items, nitems, total_chunks, KKmax);
    const size_t lds = (size_t)4 * 64 * 68 * sizeof(float);
    set_max_dynamic_lds((const void*)mfma_gram64_kernel<true>, lds); hipLaunchKernelGGL(mfma_gram64_kernel<true>, dim3(total_chunks), dim3(256), lds, s, d_items, nitems);
    TNQS_CHECK_LAUNCH();
    return true;
}

// ------------------------------------------------------------------------------------------------------------
// theta SVD for matrices that do not fit the LDS-resident one-sided Jacobi (256 x 128 at chi = 64): Cholesky-QR preprocessing.
//   G = A^dagger A + delta I (f64, n x n)  ->  G = L L^dagger (chol_packed_kernel, shift delta)  ->  R = L^dagger (n x n, f32)
//   one-sided Jacobi on R in LDS: R J = U_R Sigma_R            (jacobi_lds_kernel<float>)
//   J = R^-1 (U_R Sigma_R)                                      (tall_w_kernel, f64: the rotations are never accumulated in f32)
//   A <- A J = U Sigma of A                                    (tall_mj_kernel: the same rotations orthogonalise the columns of A, because
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
// W (n x n, ComplexF32) = Rinv (n x n upper triangular, complex128) Rrot (n x n, ComplexF32), accumulated in f64: with Rrot = R J from the
// Jacobi sweeps this is J itself, accurate to the Jacobi tolerance in EVERY direction (R^-1 damps the large singular directions; the
// V-recovery formula R^dagger (R J) S^-2 would amplify them by sigma_max / sigma_j, which turns a rank-deficient theta into wrong large
// singular values after the product A J).  The terms of the sum are huge and cancel (|Rinv| ~ 1 / sigma_min): f64 throughout.
__global__ __launch_bounds__(256) void tall_w_kernel(const TallSvdItem* __restrict__ items) {
    const TallSvdItem it = items[blockIdx.x];
    const int n = it.n, nt = (n + 31) >> 5;
    const int I = blockIdx.y % nt, J = blockIdx.y / nt;
    if (J >= nt) return;
    struct alignas(16) cd { double re, im; };
    __shared__ double Ar[32][33], Ai[32][33], Br[32][33], Bi[32][33];       // A = Rinv[32 I + i][j0 + j], B = Rrot[j0 + j][32 J + c]
    const cd* __restrict__ Rinv = reinterpret_cast<const cd*>(it.L);        // the caller passes R^-1 in the L slot of this launch
    const cf* __restrict__ Rrot = reinterpret_cast<const cf*>(it.Rrot);
    cd* __restrict__ W = reinterpret_cast<cd*>(it.R0);                     // and the output in the R0 slot, complex128: J is multiplied into A in f64 (tall_mj_kernel)
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    // a Cholesky pivot of this item collapsed in spite of the shift (flag in the G slot of this launch): R is not a usable preconditioner,
    // so J := I -- A stays as it is and the sweeps on A that follow (svd_batch, full sweep cap) do the whole factorisation themselves
    const int* fail = reinterpret_cast<const int*>(it.G);
    if (fail && *fail) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int i = 32 * I + 2 * tx + p, c = 32 * J + 2 * ty + q;
                if (i < n && c < n) { cd v; v.re = (i == c) ? 1.0 : 0.0; v.im = 0.0; W[i + (size_t)n * c] = v; }
            }
        return;
    }
    double cr[2][2] = {{0, 0}, {0, 0}}, ci[2][2] = {{0, 0}, {0, 0}};
    for (int j0 = 32 * I; j0 < n; j0 += 32) {                               // Rinv is upper triangular: rows 32 I.. only see columns >= 32 I
        for (int e = tid; e < 1024; e += 256) {
            const int a = e & 31, b = e >> 5;
            cd x = {0.0, 0.0}; cf y = {0.f, 0.f};
            if (32 * I + a < n && j0 + b < n) x = Rinv[(32 * I + a) + (size_t)n * (j0 + b)];
            if (j0 + a < n && 32 * J + b < n) y = Rrot[(j0 + a) + (size_t)n * (32 * J + b)];
            Ar[a][b] = x.re; Ai[a][b] = x.im; Br[a][b] = (double)y.re; Bi[a][b] = (double)y.im;
        }
        __syncthreads();
        for (int j = 0; j < 32; ++j) {
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const double ar = Ar[2 * tx + p][j], ai = Ai[2 * tx + p][j], br = Br[j][2 * ty + q], bi = Bi[j][2 * ty + q];
                    cr[p][q] += ar * br - ai * bi; ci[p][q] += ar * bi + ai * br;
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = 32 * I + 2 * tx + p, c = 32 * J + 2 * ty + q;
            if (i < n && c < n) { cd v; v.re = cr[p][q]; v.im = ci[p][q]; W[i + (size_t)n * c] = v; }
        }
}
void launch_tall_w(hipStream_t s, const TallSvdItem* d_items, int nitems, int nmax) {
    if (nitems <= 0) return;
    const int nt = (nmax + 31) / 32;
    hipLaunchKernelGGL(tall_w_kernel, dim3(nitems, nt * nt), dim3(256), 0, s, d_items); TNQS_CHECK_LAUNCH();
}
// C (m x n, ComplexF32) = A (m x k, ComplexF32) J (k x n, complex128) accumulated in f64 on v_mfma_f64_16x16x4_f64: one wave per 16 x 16
// tile, tile rows = columns j of C, lanes = rows i (contiguous in A and C).  A J is the rotated theta factor of the Cholesky-QR route: with J
// in f64 and the product in f64 every column of A J keeps the accuracy the Jacobi sweeps on R gave it RELATIVE TO ITS OWN NORM
// (A J = Q (R J): the columns are Q times the sweeps' output), so no polishing sweeps on A J are needed -- the f32 product they followed
// left a residue of eps32 sigma_max in every column.
typedef double v4d_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void tall_mj_kernel(const SmallGemmItem* __restrict__ items) {
    const SmallGemmItem it = items[blockIdx.x];
    struct alignas(16) cd { double re, im; };
    const cf* __restrict__ A = reinterpret_cast<const cf*>(it.A);
    const cd* __restrict__ J = reinterpret_cast<const cd*>(it.B);
    cf* __restrict__ C = reinterpret_cast<cf*>(it.C);
    const int m = it.m, n = it.n, k = it.k;
    const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
    const int w = blockIdx.y * (blockDim.x >> 6) + (threadIdx.x >> 6), nw = gridDim.y * (blockDim.x >> 6);
    const int tr = (n + 15) >> 4, tc = (m + 15) >> 4;
    for (int t = w; t < tr * tc; t += nw) {
        const int j0 = 16 * (t % tr), i0 = 16 * (t / tr);
        const int jj = j0 + l15, ii = i0 + l15;
        v4d_t cr = {0, 0, 0, 0}, ci = {0, 0, 0, 0};
        for (int k0 = 0; k0 < k; k0 += 4) {
            const int kk = k0 + kq;
            cd a = {0.0, 0.0}; double br = 0.0, bi = 0.0;
            if (kk < k) { if (jj < n) a = J[kk + (size_t)k * jj]; if (ii < m) { const cf v = A[ii + (size_t)m * kk]; br = (double)v.re; bi = (double)v.im; } }
            cr = __builtin_amdgcn_mfma_f64_16x16x4f64(a.re, br, cr, 0, 0, 0);       // C'[j][i] = sum_k J[k, j] A[i, k]
            cr = __builtin_amdgcn_mfma_f64_16x16x4f64(-a.im, bi, cr, 0, 0, 0);
            ci = __builtin_amdgcn_mfma_f64_16x16x4f64(a.re, bi, ci, 0, 0, 0);
            ci = __builtin_amdgcn_mfma_f64_16x16x4f64(a.im, br, ci, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + kq + 4 * r, i = i0 + l15;
            if (i < m && j < n) { cf v; v.re = (float)cr[r]; v.im = (float)ci[r]; C[i + (size_t)m * j] = v; }
        }
    }
}
void launch_tall_mj(hipStream_t s, const SmallGemmItem* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL(tall_mj_kernel, dim3(nitems, 4), dim3(1024), 0, s, d_items); TNQS_CHECK_LAUNCH();
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
void launch_copy_items(hipStream_t s, const CopyItem* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL(copy_items_kernel, dim3(nitems, 16), dim3(256), 0, s, d_items); TNQS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// register-direct fiber GEMM:  out[(s',n),(a,b)] = sum_{(s,k)} in[(s,k),(a,b)] X[(s,k),(s',n)]  with D*K = 32 KB exactly, Do*No <= 32 NB,
// any leg of a chi = 64 site (K = 64 mode products and the K = N = 128 gate epilogue).
// The product is computed TRANSPOSED, C'[nn][row] = sum_kk X^T[nn][kk] in[row][kk], with row = 32 CONSECUTIVE a-indices on the lanes:
//   * B' operand = the tensor itself: lane (ln = row, h) loads in[row][kk(q, h)] for every k-step q straight from global memory into
//     the registers the MFMAs read -- 256-byte (D = 1) / 512-byte (D = 2, both site components as one 16-byte load) runs per half wave,
//     no LDS staging, no transposition, no bank conflicts; the next tile's loads are issued before the current tile's matrix work;
//   * A' operand = X^T, staged once per workgroup in LDS in exact operand order (one conflict-free ds_read_b64 per lane and k-step);
//   * C' accumulators hold out[row = ln][nn = kappa(r, h) + 32 nb]: stores run along the lanes again (D = 2: registers 2j, 2j+1 are the
//     two site components of one n, one 16-byte store).
// One wave per SIMD (accumulators + the two half-tile operand sets: up to 256 registers); the wave never waits on a workgroup barrier.
// ------------------------------------------------------------------------------------------------------------
template <int KB, int NB, int D, bool M3>
__global__ __launch_bounds__(256, (KB * NB <= 4 ? 2 : 1)) void mfma_rowgemm_kernel(const FiberItem* __restrict__ items, int nitems, double* __restrict__ norm_partials) {
    constexpr int KK = 32 * KB, NQ = KK / 2;                   // k-steps per tile
    constexpr int NL = (D == 1) ? NQ : NQ / 2;                 // loads per lane and tile (8 bytes each for D = 1, 16 bytes for D = 2)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2f* const Xl = reinterpret_cast<v2f*>(smem);             // [q][nb][lane]
    __shared__ double sh_red[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, ln = lane & 31, h = lane >> 5;
    int lo = 0, hi = nitems - 1;
    const int gw = blockIdx.x;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (items[mid].tile_begin <= gw) lo = mid; else hi = mid - 1; }
    const FiberItem it = items[lo];
    const int K = it.K, No = it.No, NN = it.Do * No;
    const long long PA = it.PA;
    const int ntiles = it.nta * it.ntb;                        // nta = PA / 32 row blocks, ntb = PB
    const int t_begin = (gw - it.tile_begin) * it.tpw, t_end = min(ntiles, t_begin + it.tpw);
    const cf* __restrict__ in = reinterpret_cast<const cf*>(it.in);
    const cf* __restrict__ X = reinterpret_cast<const cf*>(it.X);
    cf* __restrict__ out = reinterpret_cast<cf*>(it.out);
    // k-step q of lane half hh reads kk: D = 1: kk = 2 q + hh;  D = 2: q = 2 t + s -> kk = s + 2 (2 t + hh)
    for (int e = tid; e < NQ * NB * 64; e += 256) {
        const int l = e & 63, nb = (e >> 6) % NB, q = e / (64 * NB);
        const int hh = l >> 5, nn = 32 * nb + (l & 31);
        const int kk = (D == 1) ? 2 * q + hh : (q & 1) + 2 * (2 * (q >> 1) + hh);
        cf v = {0.f, 0.f};
        if (nn < NN) v = X[kk + (size_t)KK * nn];
        v2f o = {v.re, v.im}; Xl[e] = o;
    }
    __syncthreads();                                           // the only workgroup barrier
    // Operand registers: TWO HALF tiles (k-steps [0, NQ/2) and [NQ/2, NQ)).  A half is refilled with the next tile's data as soon as its
    // MFMAs have been issued, so loads are always in flight behind the matrix work while only one tile's worth of operands (NQ floats x 2)
    // is live next to the accumulators -- a whole second tile (the first version) spilled 1200 registers at K = N = 128.
    constexpr int HQ = NQ / 2, HL = NL / 2;
    float hb[2][2 * HQ];
    // rows of a tile: 32 consecutive a-indices (PA >= 32), or -- first leg of the tensor, PA < 32 -- all PA a-indices of 32 / PA consecutive
    // b-indices (the lanes then sit 1 KiB apart and each streams its own contiguous fiber over the k-steps: 16-byte pieces instead of
    // 256-byte runs per instruction, which the matrix work per byte of these shapes hides)
    const bool rows_b = PA < 32;
    const int RB = rows_b ? 32 / (int)PA : 1;
    const long long kst = (long long)D * PA;                                                   // stride of the contracted index (elements)
    const long long lane_in = rows_b ? (long long)D * (ln % (int)PA) + kst * K * (ln / (int)PA) : (long long)D * ln;
    const long long lane_out = rows_b ? (long long)D * (ln % (int)PA) + kst * No * (ln / (int)PA) : (long long)D * ln;
    auto base_in = [&](int t) { return rows_b ? kst * K * RB * t : (long long)D * 32 * (t % it.nta) + kst * K * (t / it.nta); };
    auto base_out = [&](int t) { return rows_b ? kst * No * RB * t : (long long)D * 32 * (t % it.nta) + kst * No * (t / it.nta); };
    auto issue_half = [&](int t, int hf) {
        const cf* p = in + base_in(t) + lane_in + kst * (h + 2 * HL * hf);
        if (D == 1) {
#pragma unroll
            for (int j = 0; j < HL; ++j) { const v2f v = ldg2(p + kst * 2 * j); hb[hf][2 * j] = v[0]; hb[hf][2 * j + 1] = v[1]; }
        } else {
#pragma unroll
            for (int j = 0; j < HL; ++j) { const v4f v = ldg4(p + kst * 2 * j);
                                           hb[hf][4 * j] = v[0]; hb[hf][4 * j + 1] = v[1]; hb[hf][4 * j + 2] = v[2]; hb[hf][4 * j + 3] = v[3]; }
        }
    };
    double nrm = 0;
    int t = t_begin + w;
    if (t < t_end) { issue_half(t, 0); issue_half(t, 1); }
    for (; t < t_end; t += 4) {
        CAcc32<M3> C[NB];                                                        // three accumulators per block with the three-multiplication product
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
            for (int q = 0; q < HQ; ++q) {
                const float br = hb[hf][2 * q], bi = hb[hf][2 * q + 1];          // in[row][kk(q + HQ hf, h)]
                const float bd = M3 ? bi - br : bi, bs = M3 ? br + bi : -bi;     // B-side combinations, shared by the NB blocks
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const v2f a = Xl[((q + HQ * hf) * NB + nb) * 64 + lane];      // X[kk][32 nb + ln]
                    if (hf == 0 && q == 0) C[nb].template mac_bpre<true>(a[0], a[1], br, bd, bs);
                    else C[nb].template mac_bpre<false>(a[0], a[1], br, bd, bs);
                }
            }
            __builtin_amdgcn_sched_barrier(0);                                   // the refill must not be hoisted above the MFMAs that read the half
            if (t + 4 < t_end) issue_half(t + 4, hf);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) C[nb].finish();
        float nf = 0.f;
        if (D == 1) {
            cf* p = out + base_out(t) + lane_out;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = 32 * nb + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (n < No) { cf v; v.re = C[nb].a[r]; v.im = C[nb].b[r]; stgc(p + PA * n, v); nf += v.re * v.re + v.im * v.im; }
                }
        } else {
            cf* p = out + base_out(t) + lane_out;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int nn = 32 * nb + (r & 3) + 8 * (r >> 2) + 4 * h;  // even: s' = 0 of n = nn / 2; register r + 1 is s' = 1
                    const int n = nn >> 1;
                    if (n < No) {
                        v4f v = {C[nb].a[r], C[nb].b[r], C[nb].a[r + 1], C[nb].b[r + 1]};
                        stg4(p + 2 * PA * n, v);
                        nf += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                    }
                }
        }
        nrm += (double)nf;
    }
    if (it.want_norm) {
        nrm = wave_sum_d(nrm);
        if (lane == 0) sh_red[w] = nrm;
        __syncthreads();
        if (tid == 0) norm_partials[gw] = sh_red[0] + sh_red[1] + sh_red[2] + sh_red[3];
    }
}
// shapes (K = bond dimension of the contracted leg, D = 1: mode product, D = 2: gate epilogue with the site index folded in):
//   chi = 64: <2,2,1> (KK = NN = 64), <4,4,2> (KK = NN = 128);   chi = 32: <1,1,1> (KK = NN = 32), <2,2,2> (KK = NN = 64)
static int rowgemm_variant(const FiberItem& it) {
    if (it.D != it.Do || (it.D != 1 && it.D != 2) || it.No < 1 || it.No > it.K) return -1;
    if (it.K == 64) return it.D == 1 ? 0 : 1;
    if (it.K == 32) return it.D == 1 ? 2 : 3;
    return -1;
}
bool rowgemm_covers(const FiberItem& it) {
    if (it.PA >= 32) { if (it.PA % 32 != 0) return false; }
    else if (it.PA < 1 || 32 % it.PA != 0 || it.PB % (32 / it.PA) != 0) return false;
    return rowgemm_variant(it) >= 0;
}
void rowgemm_tiles(FiberItem& it) {       // tile grid of an item rowgemm_covers() accepted
    it.TA = 32; it.TB = 1;
    if (it.PA >= 32) { it.nta = it.PA / 32; it.ntb = it.PB; } else { it.nta = 1; it.ntb = it.PB / (32 / it.PA); }
}
template <int KB, int NB, int D> static void launch_rowgemm_t(hipStream_t s, const FiberItem* d_items, int nitems, int total_wgs, double* d_norm_partials) {
    const size_t lds = (size_t)(16 * KB) * NB * 64 * sizeof(v2f);
    // three-multiplication product except for K = N = 128: its 16 blocks x 3 accumulators do not fit next to the operand registers (21 spills)
    constexpr bool fits3 = KB * NB <= 4;
    if (fits3) {
        set_max_dynamic_lds((const void*)mfma_rowgemm_kernel<KB, NB, D, fits3>, lds);
        hipLaunchKernelGGL((mfma_rowgemm_kernel<KB, NB, D, fits3>), dim3(total_wgs), dim3(256), lds, s, d_items, nitems, d_norm_partials);
    } else {
        set_max_dynamic_lds((const void*)mfma_rowgemm_kernel<KB, NB, D, false>, lds);
        hipLaunchKernelGGL((mfma_rowgemm_kernel<KB, NB, D, false>), dim3(total_wgs), dim3(256), lds, s, d_items, nitems, d_norm_partials);
    }
    TNQS_CHECK_LAUNCH();
}
// all items of one launch must share K and D (the caller groups them)
void launch_mfma_rowgemm(hipStream_t s, const FiberItem* d_items, int nitems, int total_wgs, int D, int K, double* d_norm_partials) {
    if (total_wgs <= 0) return;
    if (D * K == 64 && mfma_use_x3()) { launch_x3_rowgemm64(s, d_items, nitems, total_wgs, D, d_norm_partials); return; }      // <2, 2, 1> and <2, 2, 2> on the bf16 matrix cores
    if (K == 64 && D == 1) launch_rowgemm_t<2, 2, 1>(s, d_items, nitems, total_wgs, d_norm_partials);
    else if (K == 64 && D == 2) launch_rowgemm_t<4, 4, 2>(s, d_items, nitems, total_wgs, d_norm_partials);
    else if (K == 32 && D == 1) launch_rowgemm_t<1, 1, 1>(s, d_items, nitems, total_wgs, d_norm_partials);
    else if (K == 32 && D == 2) launch_rowgemm_t<2, 2, 2>(s, d_items, nitems, total_wgs, d_norm_partials);
    else throw std::runtime_error("launch_mfma_rowgemm: shape not covered");
}

// ------------------------------------------------------------------------------------------------------------
// Gram with f64 accumulation on v_mfma_f64_16x16x4_f64 for 64 < KK = D*K <= 128 (gate path at chi = 64: G = psi~^dagger psi~ over the
// outer legs, 128 x 128).  Same scheme as mfma_gram64_f64_kernel (kernels_mfma.hip): ComplexF32 tiles of 64 fibers, double-buffered in
// LDS as [kk][row], converted on the fly (f32 products are exact in f64); G is Hermitian, so only the 16 x 16 blocks (I <= J) are
// computed -- 36 for KK = 128, dealt round-robin to the four waves (9 each) -- and mirrored when the partial is written.
// One partial per chunk.  f64 MFMA layout: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15], C[row = (l >> 4) + 4 r][col = l & 15].
// ------------------------------------------------------------------------------------------------------------
// SHARED (every item of the launch has KK = 128: eight panels, 36 upper blocks): twelve sets of three blocks that share a panel
// (gram_f64_shared, mfma_common.hpp), three sets per wave --
//   wave w = 0, 1, 2:  {(w,w),(w,w+1),(w,w+2)},  {(w+3,w+3),(w+3,w+4),(w+3,w+5)},  {(w,w+3),(w,w+4),(w,w+5)}      (rows from one panel)
//   wave 3          :  {(0,7),(1,7),(3,7)},  {(4,7),(6,7),(7,7)},  {(0,6),(3,6),(6,6)}                              (columns from one panel)
// set s of a wave owns accumulator slots 3 s .. 3 s + 2
template <bool ROW, int DIAGJ, bool M3>
__device__ __forceinline__ void gram128_set(const float* __restrict__ Xr, const float* __restrict__ Xi, int TRP, int l15, int kq, int P, int q0, int q1, int q2,
                                            v4d& r0, v4d& r1, v4d& r2, v4d& i0, v4d& i1, v4d& i2, v4d& c0, v4d& c1, v4d& c2) {
    const int Q[3] = {q0, q1, q2};
    v4d r[3] = {r0, r1, r2}, i[3] = {i0, i1, i2}, c[3] = {c0, c1, c2};
    gram_f64_shared<3, ROW, DIAGJ, M3>(Xr, Xi, TRP, l15, kq, P, Q, r, i, c);
    r0 = r[0]; r1 = r[1]; r2 = r[2]; i0 = i[0]; i1 = i[1]; i2 = i[2]; c0 = c[0]; c1 = c[1]; c2 = c[2];
}
template <bool M3, bool SHARED>
__global__ __launch_bounds__(256) void mfma_gram128_f64_kernel(const GramItem* __restrict__ items, int nitems) {
    constexpr int TR = 64, TRP = TR + 4, NU = 16, KKP = 128, NBW = 9;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const Xbuf = reinterpret_cast<float*>(smem);          // [buf][re|im][KKP * TRP]
    const int tid = threadIdx.x;
    int lo = 0, hi = nitems - 1;
    const int gc = blockIdx.x;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (items[mid].chunk_begin <= gc) lo = mid; else hi = mid - 1; }
    const GramItem it = items[lo];
    const int lc = gc - it.chunk_begin;
    const int D = it.D, K = it.K, TA = it.TA, TB = it.TB, KK = D * K;
    const long long PA = it.PA;
    const cf* __restrict__ Xg = reinterpret_cast<const cf*>(it.X);
    const int ntiles = it.nta * it.ntb;
    const int t_begin = lc * it.tiles_per_chunk;
    const int t_end = min(ntiles, t_begin + it.tiles_per_chunk);
    const int lane = tid & 63, w = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    const int nb = (KK + 15) >> 4, nblk = nb * (nb + 1) / 2;
    int bI[NBW], bJ[NBW]; bool bOn[NBW];
#pragma unroll
    for (int q = 0; q < NBW; ++q) {
        int idx = w + 4 * q; bOn[q] = idx < nblk;
        int I = 0, rem = bOn[q] ? idx : 0; while (rem >= nb - I) { rem -= nb - I; ++I; }
        bI[q] = I; bJ[q] = I + rem;
    }
    if (SHARED) {
        const int rI[9] = {0, 1, 3, 4, 6, 7, 0, 3, 6}, rJ[9] = {7, 7, 7, 7, 7, 7, 6, 6, 6};            // wave 3
#pragma unroll
        for (int q = 0; q < NBW; ++q) {
            bOn[q] = true;
            if (w < 3) { const int st = q / 3, k = q % 3; bI[q] = st == 1 ? w + 3 : w; bJ[q] = st == 0 ? w + k : (st == 1 ? w + 3 + k : w + 3 + k); }
            else { bI[q] = rI[q]; bJ[q] = rJ[q]; }
        }
    }
    v4d Cr[NBW], Ci[NBW], Cc[NBW];                             // M3: sum (ar+ai) br,  sum ai (br-bi),  sum ar (bi+br)  (CAcc32::mac_conj in f64)
#pragma unroll
    for (int q = 0; q < NBW; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) { Cr[q][r] = 0.0; Ci[q][r] = 0.0; Cc[q][r] = 0.0; }
    for (int e = tid; e < 4 * KKP * TRP; e += 256) Xbuf[e] = 0.f;
    const TileMap m = make_map(tid, D, TA, TB, PA, K);
    const long long kstride = (long long)D * PA;
    const bool fast = m.U <= 256 && (K + m.KP - 1) / m.KP <= NU;
    v4f px[NU];
    auto tile_origin = [&](int t, int& a0, int& b0, int& na, int& nbb) {
        int ta = t % it.nta, tb = t / it.nta;
        a0 = ta * TA; b0 = tb * TB; na = min(TA, it.PA - a0); nbb = min(TB, it.PB - b0);
    };
    const bool straight = fast && m.active && m.vec == 2 && K == m.KP * NU;      // see mfma_gram64_kernel: guarded loads are serialised
    auto issue_loads = [&](int t) {
        int a0, b0, na, nbb; tile_origin(t, a0, b0, na, nbb);
        const long long org = (long long)D * (a0 + PA * (long long)K * b0);
        if (straight && na == TA && nbb == TB) {
            const cf* p0 = Xg + org + m.off + kstride * m.kp; const long long st = kstride * m.KP;
#pragma unroll
            for (int j = 0; j < NU; ++j) px[j] = ldg4(p0 + st * j);
            return;
        }
        const bool v0 = m.active && m.al < na && m.bl < nbb, v1 = v0 && m.al1 < na;
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            int k = m.kp + m.KP * j;
            v4f vx; vx[0] = vx[1] = vx[2] = vx[3] = 0.f;
            if (k < K && v0) {
                const long long o = org + m.off + kstride * k;
                if (m.vec == 2 && v1) vx = ldg4(Xg + o);
                else { cf x = ldgc(Xg + o); vx[0] = x.re; vx[1] = x.im; }
            }
            px[j] = vx;
        }
    };
    auto commit_loads = [&](int buf) {
        if (!m.active) return;
        float* Xr = Xbuf + buf * (2 * KKP * TRP); float* Xi = Xr + KKP * TRP;
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            int k = m.kp + m.KP * j;
            if (k < K) {
                int o0 = (m.c0 + D * k) * TRP + m.row0;
                Xr[o0] = px[j][0]; Xi[o0] = px[j][1];
                if (m.vec == 2) { int o1 = (m.c1 + D * k) * TRP + m.row1; Xr[o1] = px[j][2]; Xi[o1] = px[j][3]; }
            }
        }
    };
    auto fill_slow = [&](int t, int buf) {
        float* Xr = Xbuf + buf * (2 * KKP * TRP); float* Xi = Xr + KKP * TRP;
        int a0, b0, na, nbb; tile_origin(t, a0, b0, na, nbb);
        const int ntile_el = D * TA * K * TB;
        for (int e = tid; e < ntile_el; e += 256) {
            int s = e % D; int r1 = e / D; int al = r1 % TA; int r2 = r1 / TA; int k = r2 % K; int bl = r2 / K;
            cf vx; vx.re = vx.im = 0.f;
            if (al < na && bl < nbb) vx = Xg[s + D * ((long long)(a0 + al) + PA * ((long long)k + (long long)K * (b0 + bl)))];
            int o = (s + D * k) * TRP + (al + TA * bl);
            Xr[o] = vx.re; Xi[o] = vx.im;
        }
    };
    lds_barrier();                                               // zero fill done
    if (t_begin < t_end) { if (fast) { issue_loads(t_begin); commit_loads(0); if (t_begin + 1 < t_end) issue_loads(t_begin + 1); } else fill_slow(t_begin, 0); }
    lds_barrier();
    for (int t = t_begin; t < t_end; ++t) {
        const int cur = (t - t_begin) & 1;
        if (t + 1 < t_end) { if (fast) { commit_loads(cur ^ 1); if (t + 2 < t_end) issue_loads(t + 2); } else fill_slow(t + 1, cur ^ 1); }
        const float* Xr = Xbuf + cur * (2 * KKP * TRP); const float* Xi = Xr + KKP * TRP;
        if (SHARED) {
#define TNQS_SET(ROW, DJ, S, P, A, B, C) gram128_set<ROW, DJ, M3>(Xr, Xi, TRP, l15, kq, P, A, B, C, Cr[3 * S], Cr[3 * S + 1], Cr[3 * S + 2], \
                                                                  Ci[3 * S], Ci[3 * S + 1], Ci[3 * S + 2], Cc[3 * S], Cc[3 * S + 1], Cc[3 * S + 2])
            if (w < 3) {
                TNQS_SET(true, 0, 0, w, w, w + 1, w + 2);
                TNQS_SET(true, 0, 1, w + 3, w + 3, w + 4, w + 5);
                TNQS_SET(true, -1, 2, w, w + 3, w + 4, w + 5);
            } else {
                TNQS_SET(false, -1, 0, 7, 0, 1, 3);
                TNQS_SET(false, 2, 1, 7, 4, 6, 7);
                TNQS_SET(false, 2, 2, 6, 0, 3, 6);
            }
#undef TNQS_SET
        } else
#pragma unroll
        for (int q = 0; q < NBW; ++q) {
            if (!bOn[q]) continue;                               // wave-uniform
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int ro = (16 * bI[q] + l15) * TRP + 16 * kq + 8 * half;
                const int rb = (16 * bJ[q] + l15) * TRP + 16 * kq + 8 * half;
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    const v4f t0 = *reinterpret_cast<const v4f*>(Xr + ro + 4 * qq), t1 = *reinterpret_cast<const v4f*>(Xi + ro + 4 * qq);
                    const v4f u0 = *reinterpret_cast<const v4f*>(Xr + rb + 4 * qq), u1 = *reinterpret_cast<const v4f*>(Xi + rb + 4 * qq);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const double ar = (double)t0[c], ai = (double)t1[c], br = (double)u0[c], bi = (double)u1[c];
                        if (M3) {                                                                      // out[i][j] += x[i] conj(x[j])
                            Cr[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar + ai, br, Cr[q], 0, 0, 0);
                            Ci[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, br - bi, Ci[q], 0, 0, 0);
                            Cc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, bi + br, Cc[q], 0, 0, 0);
                        } else {
                            Cr[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, Cr[q], 0, 0, 0);
                            Ci[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, br, Ci[q], 0, 0, 0);
                            Cr[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, bi, Cr[q], 0, 0, 0);
                            Ci[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ar, bi, Ci[q], 0, 0, 0);
                        }
                    }
                }
            }
        }
        lds_barrier();                                           // tile t consumed by everybody, tile t+1 committed by everybody
    }
    struct alignas(16) cd { double re, im; };
    cd* __restrict__ part = reinterpret_cast<cd*>(it.partial) + (size_t)lc * KK * KK;
#pragma unroll
    for (int q = 0; q < NBW; ++q) {
        if (!bOn[q]) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * bI[q] + kq + 4 * r, j = 16 * bJ[q] + l15;
            if (i < KK && j < KK) {
                cd v; v.re = M3 ? Cr[q][r] - Ci[q][r] : Cr[q][r]; v.im = M3 ? Cr[q][r] - Cc[q][r] : Ci[q][r]; part[i + (size_t)KK * j] = v;
                if (bI[q] != bJ[q]) { cd c; c.re = v.re; c.im = -v.im; part[j + (size_t)KK * i] = c; }
            }
        }
    }
}
bool launch_mfma_gram128_f64(hipStream_t s, const GramItem* d_items, int nitems, int total_chunks, int KKmax, bool all_kk128) {
    if (KKmax > 128) return false;
    if (total_chunks <= 0) return true;
    const size_t lds = (size_t)4 * 128 * 68 * sizeof(float);
#define TNQS_G128(M3, SH) { set_max_dynamic_lds((const void*)mfma_gram128_f64_kernel<M3, SH>, lds); hipLaunchKernelGGL((mfma_gram128_f64_kernel<M3, SH>), dim3(total_chunks), dim3(256), lds, s, d_items, nitems); }
    if (all_kk128) TNQS_G128(true, true) else TNQS_G128(true, false)
#undef TNQS_G128
    TNQS_CHECK_LAUNCH();
    return true;
}

}  // namespace tnqs
