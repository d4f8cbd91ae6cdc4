// kernels_mfma.hip -- CDNA4 (gfx950) matrix-core fast paths for ComplexF32 data.
//
// Complex GEMMs are run as 4 real v_mfma_f32_32x32x2_f32 products on split re/im planes staged in LDS
// (exact f32: an fmaf chain, so results agree with the generic kernels to rounding-order differences).
// MFMA operand layout used throughout (wave64, h = lane>>5, ln = lane&31):
//      A[i = ln][k = h]   B[k = h][j = ln]   C[row = (r&3) + 8*(r>>2) + 4*h][col = ln],  r = 0..15
// Because the k index only has to be consistent between A and B, k-step t of a 32-deep block is mapped to
// k = t + 16*h, so every lane reads 16 CONSECUTIVE floats of its operand row from LDS (4 x ds_read_b128).
#include <hip/hip_runtime.h>
#include "kernels.hpp"

namespace tnqs {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
struct alignas(8) cf { float re, im; };

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------------------------------------------------
// fiber GEMM:  out[(s',n),(a,b)] = sum_{(s,k)} in[(s,k),(a,b)] X[(s,k),(s',n)]   with D*K <= 32*KB, Do*No <= 32*NB
// one workgroup = one tile of TR fibers; zero padding in LDS makes any K, N legal.
// ------------------------------------------------------------------------------------------------------------
template <int KB, int NB, int TR>
__global__ __launch_bounds__(256) void mfma_fiber_gemm_kernel(const FiberItem* __restrict__ items, int nitems,
                                                              double* __restrict__ norm_partials) {
    constexpr int KKP = 32 * KB, NNP = 32 * NB;
    constexpr int CP = (KKP > NNP ? KKP : NNP);
    constexpr int PA_ = CP + 4;        // pitch (floats) of a tile row: 16-B aligned, conflict-free b128 reads
    constexpr int PX = KKP + 4;
    constexpr int RB = TR / 32;        // row blocks per tile
    static_assert(RB * NB == 4, "one (row block, column block) unit per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* At_re = reinterpret_cast<float*>(smem);
    float* At_im = At_re + TR * PA_;
    float* Xt_re = At_im + TR * PA_;
    float* Xt_im = Xt_re + NNP * PX;
    __shared__ double sh_red[4];
    const int tid = threadIdx.x;
    int lo = 0, hi = nitems - 1;
    const int gt = blockIdx.x;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (items[mid].tile_begin <= gt) lo = mid; else hi = mid - 1; }
    const FiberItem it = items[lo];
    const int lt = gt - it.tile_begin;
    const int ta = lt % it.nta, tb = lt / it.nta;
    const int a0 = ta * it.TA, b0 = tb * it.TB;
    const int na = min(it.TA, it.PA - a0), nb = min(it.TB, it.PB - b0);
    const int D = it.D, K = it.K, TA = it.TA, TB = it.TB, KK = D * K;
    const int Do = it.Do, No = it.No, NN = Do * No;
    const size_t PA = it.PA;
    const cf* __restrict__ in = reinterpret_cast<const cf*>(it.in);
    const cf* __restrict__ X = reinterpret_cast<const cf*>(it.X);
    // ---- stage X^T (zero padded) ---------------------------------------------------------------------------
    for (int e = tid; e < NNP * KKP; e += 256) {
        int kk = e % KKP, nn = e / KKP;
        cf v; v.re = 0.f; v.im = 0.f;
        if (kk < KK && nn < NN) v = X[kk + (size_t)KK * nn];
        Xt_re[nn * PX + kk] = v.re; Xt_im[nn * PX + kk] = v.im;
    }
    // ---- stage the input tile: rows = fibers, columns = (s,k) ---------------------------------------------------
    const int ntile_el = D * TA * K * TB;
    for (int e = tid; e < ntile_el; e += 256) {
        int s = e % D; int r1 = e / D; int al = r1 % TA; int r2 = r1 / TA; int k = r2 % K; int bl = r2 / K;
        cf v; v.re = 0.f; v.im = 0.f;
        if (al < na && bl < nb) v = in[s + D * ((size_t)(a0 + al) + PA * ((size_t)k + (size_t)K * (b0 + bl)))];
        int row = al + TA * bl;
        At_re[row * PA_ + s + D * k] = v.re; At_im[row * PA_ + s + D * k] = v.im;
    }
    if (KK < KKP) for (int e = tid; e < TR * (KKP - KK); e += 256) {
        int row = e / (KKP - KK), c = KK + e % (KKP - KK);
        At_re[row * PA_ + c] = 0.f; At_im[row * PA_ + c] = 0.f;
    }
    if (TA * TB < TR) for (int e = tid; e < (TR - TA * TB) * KKP; e += 256) {
        int row = TA * TB + e / KKP, c = e % KKP;
        At_re[row * PA_ + c] = 0.f; At_im[row * PA_ + c] = 0.f;
    }
    __syncthreads();
    // ---- MFMA: wave w owns (row block rb, column block cb) ----------------------------------------------------
    const int lane = tid & 63, w = tid >> 6, ln = lane & 31, h = lane >> 5;
    const int rb = w % RB, cb = w / RB;
    v16f Cr, Ci;
#pragma unroll
    for (int r = 0; r < 16; ++r) { Cr[r] = 0.f; Ci[r] = 0.f; }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        float ar[16], ai[16], br[16], bi[16];
        const float* pa_r = At_re + (32 * rb + ln) * PA_ + 32 * kb + 16 * h;
        const float* pa_i = At_im + (32 * rb + ln) * PA_ + 32 * kb + 16 * h;
        const float* pb_r = Xt_re + (32 * cb + ln) * PX + 32 * kb + 16 * h;
        const float* pb_i = Xt_im + (32 * cb + ln) * PX + 32 * kb + 16 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v4f t0 = *reinterpret_cast<const v4f*>(pa_r + 4 * q), t1 = *reinterpret_cast<const v4f*>(pa_i + 4 * q);
            v4f t2 = *reinterpret_cast<const v4f*>(pb_r + 4 * q), t3 = *reinterpret_cast<const v4f*>(pb_i + 4 * q);
#pragma unroll
            for (int c = 0; c < 4; ++c) { ar[4 * q + c] = t0[c]; ai[4 * q + c] = t1[c]; br[4 * q + c] = t2[c]; bi[4 * q + c] = t3[c]; }
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            Cr = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[t], br[t], Cr, 0, 0, 0);
            Cr = __builtin_amdgcn_mfma_f32_32x32x2f32(-ai[t], bi[t], Cr, 0, 0, 0);
            Ci = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[t], bi[t], Ci, 0, 0, 0);
            Ci = __builtin_amdgcn_mfma_f32_32x32x2f32(ai[t], br[t], Ci, 0, 0, 0);
        }
    }
    __syncthreads();       // every wave has read its operands: the tile rows can now take the results
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int row = 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * h;
        At_re[row * PA_ + 32 * cb + ln] = Cr[r];
        At_im[row * PA_ + 32 * cb + ln] = Ci[r];
    }
    __syncthreads();
    // ---- coalesced store of the output tile -------------------------------------------------------------------
    cf* __restrict__ out = reinterpret_cast<cf*>(it.out);
    double nrm = 0;
    const int nout_el = Do * TA * No * TB;
    for (int e = tid; e < nout_el; e += 256) {
        int sp = e % Do; int r1 = e / Do; int al = r1 % TA; int r2 = r1 / TA; int n = r2 % No; int bl = r2 / No;
        if (al < na && bl < nb) {
            int row = al + TA * bl, nn = sp + Do * n;
            cf v; v.re = At_re[row * PA_ + nn]; v.im = At_im[row * PA_ + nn];
            out[sp + Do * ((size_t)(a0 + al) + PA * ((size_t)n + (size_t)No * (b0 + bl)))] = v;
            nrm += (double)v.re * v.re + (double)v.im * v.im;
        }
    }
    if (it.want_norm) {
        nrm = wave_sum_d(nrm);
        if (lane == 0) sh_red[w] = nrm;
        __syncthreads();
        if (tid == 0) norm_partials[gt] = sh_red[0] + sh_red[1] + sh_red[2] + sh_red[3];
    }
}

template <int KB, int NB, int TR> static size_t fiber_lds() {
    constexpr int KKP = 32 * KB, NNP = 32 * NB; constexpr int CP = (KKP > NNP ? KKP : NNP);
    return (size_t)(2 * TR * (CP + 4) + 2 * NNP * (KKP + 4)) * sizeof(float);
}
int mfma_fiber_tile_rows(int KK, int NN) {
    if (KK <= 32 && NN <= 32) return 128;
    if (KK <= 64 && NN <= 64) return 64;
    return 0;
}
// returns false if the shape is not covered (caller falls back to the generic kernel)
bool launch_mfma_fiber_gemm(hipStream_t s, const FiberItem* d_items, int nitems, int total_tiles, int KKmax, int NNmax,
                            double* d_norm_partials) {
    if (total_tiles <= 0) return true;
    if (KKmax <= 32 && NNmax <= 32) {
        const size_t lds = fiber_lds<1, 1, 128>();
        hipLaunchKernelGGL((mfma_fiber_gemm_kernel<1, 1, 128>), dim3(total_tiles), dim3(256), lds, s, d_items, nitems, d_norm_partials);
        return true;
    }
    if (KKmax <= 64 && NNmax <= 64) {
        const size_t lds = fiber_lds<2, 2, 64>();
        static bool attr = false;
        if (!attr) { (void)hipFuncSetAttribute((const void*)mfma_fiber_gemm_kernel<2, 2, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
        hipLaunchKernelGGL((mfma_fiber_gemm_kernel<2, 2, 64>), dim3(total_tiles), dim3(256), lds, s, d_items, nitems, d_norm_partials);
        return true;
    }
    return false;
}

// ------------------------------------------------------------------------------------------------------------
// Gram (f32 accumulate):  partial[4*c + w][i + KK*j] = sum_{rows of chunk c handled by wave w} X[i,row] conj(Y[j,row])
// KK = D*K <= 32.  Tiles of 64 fibers; wave w takes 16 of them per tile.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mfma_gram32_kernel(const GramItem* __restrict__ items, int nitems) {
    constexpr int TR = 64, TRP = TR + 4;
    __shared__ __attribute__((aligned(16))) float Xr[32 * TRP];
    __shared__ __attribute__((aligned(16))) float Xi[32 * TRP];
    __shared__ __attribute__((aligned(16))) float Yr[32 * TRP];
    __shared__ __attribute__((aligned(16))) float Yi[32 * TRP];
    const int tid = threadIdx.x;
    int lo = 0, hi = nitems - 1;
    const int gc = blockIdx.x;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (items[mid].chunk_begin <= gc) lo = mid; else hi = mid - 1; }
    const GramItem it = items[lo];
    const int lc = gc - it.chunk_begin;
    const int D = it.D, K = it.K, TA = it.TA, TB = it.TB, KK = D * K;
    const size_t PA = it.PA;
    const bool same = (it.X == it.Y);
    const cf* __restrict__ Xg = reinterpret_cast<const cf*>(it.X);
    const cf* __restrict__ Yg = reinterpret_cast<const cf*>(it.Y);
    const int ntiles = it.nta * it.ntb;
    const int t_begin = lc * it.tiles_per_chunk;
    const int t_end = min(ntiles, t_begin + it.tiles_per_chunk);
    const int lane = tid & 63, w = tid >> 6, ln = lane & 31, h = lane >> 5;
    v16f Cr, Ci;
#pragma unroll
    for (int r = 0; r < 16; ++r) { Cr[r] = 0.f; Ci[r] = 0.f; }
    // zero the padding rows (kk >= KK) once
    for (int e = tid; e < (32 - KK) * TRP; e += 256) { int o = KK * TRP + e; Xr[o] = 0.f; Xi[o] = 0.f; Yr[o] = 0.f; Yi[o] = 0.f; }
    const int ntile_el = D * TA * K * TB;
    for (int t = t_begin; t < t_end; ++t) {
        const int ta = t % it.nta, tb = t / it.nta;
        const int a0 = ta * TA, b0 = tb * TB;
        const int na = min(TA, it.PA - a0), nb = min(TB, it.PB - b0);
        __syncthreads();
        for (int e = tid; e < ntile_el; e += 256) {
            int s = e % D; int r1 = e / D; int al = r1 % TA; int r2 = r1 / TA; int k = r2 % K; int bl = r2 / K;
            cf vx, vy; vx.re = vx.im = vy.re = vy.im = 0.f;
            if (al < na && bl < nb) {
                size_t off = s + D * ((size_t)(a0 + al) + PA * ((size_t)k + (size_t)K * (b0 + bl)));
                vx = Xg[off];
                vy = same ? vx : Yg[off];
            }
            int o = (s + D * k) * TRP + (al + TA * bl);
            Xr[o] = vx.re; Xi[o] = vx.im; Yr[o] = vy.re; Yi[o] = vy.im;
        }
        if (TA * TB < TR) for (int e = tid; e < 32 * (TR - TA * TB); e += 256) {
            int kk = e / (TR - TA * TB), row = TA * TB + e % (TR - TA * TB); int o = kk * TRP + row;
            Xr[o] = 0.f; Xi[o] = 0.f; Yr[o] = 0.f; Yi[o] = 0.f;
        }
        __syncthreads();
        // wave w: rows 16w .. 16w+15; lane half h takes rows 16w + 8h + t
        float xr[8], xi[8], yr[8], yi[8];
        const int ro = ln * TRP + 16 * w + 8 * h;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            v4f t0 = *reinterpret_cast<const v4f*>(Xr + ro + 4 * q), t1 = *reinterpret_cast<const v4f*>(Xi + ro + 4 * q);
            v4f t2 = *reinterpret_cast<const v4f*>(Yr + ro + 4 * q), t3 = *reinterpret_cast<const v4f*>(Yi + ro + 4 * q);
#pragma unroll
            for (int c = 0; c < 4; ++c) { xr[4 * q + c] = t0[c]; xi[4 * q + c] = t1[c]; yr[4 * q + c] = t2[c]; yi[4 * q + c] = t3[c]; }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            // out[i][j] += x[i] * conj(y[j])
            Cr = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[q], yr[q], Cr, 0, 0, 0);
            Cr = __builtin_amdgcn_mfma_f32_32x32x2f32(xi[q], yi[q], Cr, 0, 0, 0);
            Ci = __builtin_amdgcn_mfma_f32_32x32x2f32(xi[q], yr[q], Ci, 0, 0, 0);
            Ci = __builtin_amdgcn_mfma_f32_32x32x2f32(-xr[q], yi[q], Ci, 0, 0, 0);
        }
    }
    cf* __restrict__ part = reinterpret_cast<cf*>(it.partial) + (size_t)(4 * lc + w) * KK * KK;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * h, j = ln;
        if (i < KK && j < KK) { cf v; v.re = Cr[r]; v.im = Ci[r]; part[i + (size_t)KK * j] = v; }
    }
}
bool launch_mfma_gram32(hipStream_t s, const GramItem* d_items, int nitems, int total_chunks, int KKmax) {
    if (KKmax > 32) return false;
    if (total_chunks <= 0) return true;
    hipLaunchKernelGGL(mfma_gram32_kernel, dim3(total_chunks), dim3(256), 0, s, d_items, nitems);
    return true;
}

}  // namespace tnqs
