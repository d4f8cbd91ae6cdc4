// kernels_mfma.hip -- CDNA4 (gfx950) matrix-core fast paths for ComplexF32 data.
//
// Complex GEMMs are run as 4 real v_mfma_f32_32x32x2_f32 products on split re/im planes staged in LDS
// (exact f32: an fmaf chain, so results agree with the generic kernels to rounding-order differences).
// MFMA operand layout used throughout (wave64, h = lane>>5, ln = lane&31):
//      A[i = ln][k = h]   B[k = h][j = ln]   C[row = (r&3) + 8*(r>>2) + 4*h][col = ln],  r = 0..15
// Because the k index only has to be consistent between A and B, k-step t of a 32-deep block is mapped to
// k = t + 16*h, so every lane reads 16 CONSECUTIVE floats of its operand row from LDS (4 x ds_read_b128).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdexcept>
#include <string>
#define TNQS_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) throw std::runtime_error(std::string("HIP kernel launch failed (") + __func__ + "): " + hipGetErrorString(e_)); } while (0)
#include "kernels.hpp"
#include "mfma_common.hpp"
#include <type_traits>
#include "launch_util.hpp"

namespace tnqs {

// ------------------------------------------------------------------------------------------------------------
// fiber GEMM:  out[(s',n),(a,b)] = sum_{(s,k)} in[(s,k),(a,b)] X[(s,k),(s',n)]   with D*K <= 32*KB, Do*No <= 32*NB.
// One workgroup stages X^T once and walks `tpw` consecutive tiles of TR fibers; the next tile's global loads are
// issued before the MFMA block of the current one.  Zero padding in LDS makes any K, N legal.
// ------------------------------------------------------------------------------------------------------------
template <int KB, int NB, int TR>
__global__ __launch_bounds__(256) void mfma_fiber_gemm_kernel(const FiberItem* __restrict__ items, int nitems,
                                                              double* __restrict__ norm_partials) {
    constexpr int KKP = 32 * KB, NNP = 32 * NB;
    constexpr int CP = (KKP > NNP ? KKP : NNP);
    constexpr int PT = CP + 1;         // odd pitch: conflict-free row-strided writes and ds_read_b32 operand reads
    constexpr int PX = KKP + 1;
    constexpr int RB = TR / 32;
    constexpr int NU = 8;              // max units per thread per tile
    static_assert(RB * NB == 4, "one (row block, column block) unit per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* At_re = reinterpret_cast<float*>(smem);
    float* At_im = At_re + TR * PT;
    float* Xt_re = At_im + TR * PT;
    float* Xt_im = Xt_re + NNP * PX;
    __shared__ double sh_red[4];
    const int tid = threadIdx.x;
    int lo = 0, hi = nitems - 1;
    const int gw = blockIdx.x;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (items[mid].tile_begin <= gw) lo = mid; else hi = mid - 1; }
    const FiberItem it = items[lo];
    const int D = it.D, K = it.K, TA = it.TA, TB = it.TB, KK = D * K;
    const int Do = it.Do, No = it.No, NN = Do * No;
    const long long PA = it.PA;
    const int ntiles = it.nta * it.ntb;
    const int t_begin = (gw - it.tile_begin) * it.tpw;
    const int t_end = min(ntiles, t_begin + it.tpw);
    const cf* __restrict__ in = reinterpret_cast<const cf*>(it.in);
    const cf* __restrict__ X = reinterpret_cast<const cf*>(it.X);
    cf* __restrict__ out = reinterpret_cast<cf*>(it.out);
    // ---- stage X^T (zero padded) once ---------------------------------------------------------------------------
    for (int e = tid; e < NNP * KKP; e += 256) {
        int kk = e & (KKP - 1), nn = e / KKP;
        cf v; v.re = 0.f; v.im = 0.f;
        if (kk < KK && nn < NN) v = X[kk + (size_t)KK * nn];
        Xt_re[nn * PX + kk] = v.re; Xt_im[nn * PX + kk] = v.im;
    }
    // zero the whole A tile once (padding columns / rows stay zero: tiles only overwrite valid cells)
    for (int e = tid; e < TR * PT; e += 256) { At_re[e] = 0.f; At_im[e] = 0.f; }
    const TileMap mi = make_map(tid, D, TA, TB, PA, K);
    const TileMap mo = make_map(tid, Do, TA, TB, PA, No);
    const long long kstride_in = (long long)D * PA, kstride_out = (long long)Do * PA;
    const bool fast = mi.U <= 256 && mo.U <= 256 && (K + mi.KP - 1) / mi.KP <= NU;
    const int lane = tid & 63, w = tid >> 6, ln = lane & 31, h = lane >> 5;
    const int rb = w % RB, cb = w / RB;
    v4f pre[NU];
    auto tile_origin = [&](int t, int& a0, int& b0, int& na, int& nb) {
        int ta = t % it.nta, tb = t / it.nta;
        a0 = ta * TA; b0 = tb * TB; na = min(TA, it.PA - a0); nb = min(TB, it.PB - b0);
    };
    auto issue_loads = [&](int t) {
        int a0, b0, na, nb; tile_origin(t, a0, b0, na, nb);
        const long long org = (long long)D * (a0 + PA * (long long)K * b0);
        const bool v0 = mi.active && mi.al < na && mi.bl < nb, v1 = v0 && mi.al1 < na;
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            int k = mi.kp + mi.KP * j;
            v4f v; v[0] = v[1] = v[2] = v[3] = 0.f;
            if (k < K && v0) {
                const cf* p = in + org + mi.off + kstride_in * k;
                if (mi.vec == 2 && v1) v = ldg4(p);
                else { cf x = ldgc(p); v[0] = x.re; v[1] = x.im; }
            }
            pre[j] = v;
        }
    };
    auto commit_loads = [&]() {
        if (!mi.active) return;
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            int k = mi.kp + mi.KP * j;
            if (k < K) {
                int kk0 = mi.c0 + D * k;
                At_re[mi.row0 * PT + kk0] = pre[j][0]; At_im[mi.row0 * PT + kk0] = pre[j][1];
                if (mi.vec == 2) { int kk1 = mi.c1 + D * k; At_re[mi.row1 * PT + kk1] = pre[j][2]; At_im[mi.row1 * PT + kk1] = pre[j][3]; }
            }
        }
    };
    double nrm = 0;
    if (fast && t_begin < t_end) issue_loads(t_begin);
    for (int t = t_begin; t < t_end; ++t) {
        int a0, b0, na, nb; tile_origin(t, a0, b0, na, nb);
        lds_barrier();                       // previous tile's output has left the LDS tile
        if (fast) commit_loads();
        else {
            const int ntile_el = D * TA * K * TB;
            for (int e = tid; e < ntile_el; e += 256) {
                int s = e % D; int r1 = e / D; int al = r1 % TA; int r2 = r1 / TA; int k = r2 % K; int bl = r2 / K;
                cf v; v.re = 0.f; v.im = 0.f;
                if (al < na && bl < nb) v = in[s + D * ((long long)(a0 + al) + PA * ((long long)k + (long long)K * (b0 + bl)))];
                int row = al + TA * bl;
                At_re[row * PT + s + D * k] = v.re; At_im[row * PT + s + D * k] = v.im;
            }
        }
        lds_barrier();
        if (fast && t + 1 < t_end) issue_loads(t + 1);       // in flight while the matrix cores work
        v16f Cr, Ci;
#pragma unroll
        for (int r = 0; r < 16; ++r) { Cr[r] = 0.f; Ci[r] = 0.f; }
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            float ar[16], ai[16], br[16], bi[16];
            const float* pa_r = At_re + (32 * rb + ln) * PT + 32 * kb + 16 * h;
            const float* pa_i = At_im + (32 * rb + ln) * PT + 32 * kb + 16 * h;
            const float* pb_r = Xt_re + (32 * cb + ln) * PX + 32 * kb + 16 * h;
            const float* pb_i = Xt_im + (32 * cb + ln) * PX + 32 * kb + 16 * h;
#pragma unroll
            for (int q = 0; q < 16; ++q) { ar[q] = pa_r[q]; ai[q] = pa_i[q]; br[q] = pb_r[q]; bi[q] = pb_i[q]; }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                Cr = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[q], br[q], Cr, 0, 0, 0);
                Cr = __builtin_amdgcn_mfma_f32_32x32x2f32(-ai[q], bi[q], Cr, 0, 0, 0);
                Ci = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[q], bi[q], Ci, 0, 0, 0);
                Ci = __builtin_amdgcn_mfma_f32_32x32x2f32(ai[q], br[q], Ci, 0, 0, 0);
            }
        }
        lds_barrier();       // every wave has read its operands: the tile rows can now take the results
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * h;
            At_re[row * PT + 32 * cb + ln] = Cr[r];
            At_im[row * PT + 32 * cb + ln] = Ci[r];
        }
        lds_barrier();
        // ---- coalesced store of the output tile ---------------------------------------------------------------
        if (fast) {
            const long long org = (long long)Do * (a0 + PA * (long long)No * b0);
            const bool v0 = mo.active && mo.al < na && mo.bl < nb, v1 = v0 && mo.al1 < na;
            if (v0) {
                for (int n = mo.kp; n < No; n += mo.KP) {
                    int nn0 = mo.c0 + Do * n;
                    v4f v;
                    v[0] = At_re[mo.row0 * PT + nn0]; v[1] = At_im[mo.row0 * PT + nn0];
                    cf* p = out + org + mo.off + kstride_out * n;
                    nrm += (double)v[0] * v[0] + (double)v[1] * v[1];
                    if (mo.vec == 2 && v1) {
                        int nn1 = mo.c1 + Do * n;
                        v[2] = At_re[mo.row1 * PT + nn1]; v[3] = At_im[mo.row1 * PT + nn1];
                        nrm += (double)v[2] * v[2] + (double)v[3] * v[3];
                        stg4(p, v);
                    } else { cf x; x.re = v[0]; x.im = v[1]; stgc(p, x); }
                }
            }
        } else {
            const int nout_el = Do * TA * No * TB;
            for (int e = tid; e < nout_el; e += 256) {
                int sp = e % Do; int r1 = e / Do; int al = r1 % TA; int r2 = r1 / TA; int n = r2 % No; int bl = r2 / No;
                if (al < na && bl < nb) {
                    int row = al + TA * bl, nn = sp + Do * n;
                    cf v; v.re = At_re[row * PT + nn]; v.im = At_im[row * PT + nn];
                    out[sp + Do * ((long long)(a0 + al) + PA * ((long long)n + (long long)No * (b0 + bl)))] = v;
                    nrm += (double)v.re * v.re + (double)v.im * v.im;
                }
            }
        }
        // the result cells written beyond the valid KK columns must be cleared again for the next tile's operands
        if (NN > KK || KK < KKP) {
            lds_barrier();
            for (int e = tid; e < TR * (CP - KK); e += 256) { int row = e / (CP - KK), c = KK + e % (CP - KK); At_re[row * PT + c] = 0.f; At_im[row * PT + c] = 0.f; }
        }
    }
    if (it.want_norm) {
        nrm = wave_sum_d(nrm);
        if (lane == 0) sh_red[w] = nrm;
        __syncthreads();
        if (tid == 0) norm_partials[gw] = sh_red[0] + sh_red[1] + sh_red[2] + sh_red[3];
    }
}

// ------------------------------------------------------------------------------------------------------------
// wave-private variant: every wave owns tiles of 32 fibers and its own LDS slab, so there is NO workgroup barrier in
// the tile loop -- the 3-4 waves resident on a SIMD drift apart and one wave's MFMA block overlaps the others'
// global / LDS phases.  This is the production kernel; the workgroup-tile kernel above is kept for A/B.
// ------------------------------------------------------------------------------------------------------------
template <int KB, int NB, int NU>
__global__ __launch_bounds__(256) void mfma_fiber_gemm_w_kernel(const FiberItem* __restrict__ items, int nitems,
                                                                double* __restrict__ norm_partials) {
    constexpr int KKP = 32 * KB, NNP = 32 * NB;
    constexpr int CP = (KKP > NNP ? KKP : NNP);
    constexpr int PT = CP + 1, PX = KKP + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Xt_re = reinterpret_cast<float*>(smem);
    float* Xt_im = Xt_re + NNP * PX;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, ln = lane & 31, h = lane >> 5;
    float* At_re = Xt_im + NNP * PX + w * (2 * 32 * PT);
    float* At_im = At_re + 32 * PT;
    __shared__ double sh_red[4];
    int lo = 0, hi = nitems - 1;
    const int gw = blockIdx.x;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (items[mid].tile_begin <= gw) lo = mid; else hi = mid - 1; }
    const FiberItem it = items[lo];
    const int D = it.D, K = it.K, TA = it.TA, TB = it.TB, KK = D * K;
    const int Do = it.Do, No = it.No, NN = Do * No;
    const long long PA = it.PA;
    const int ntiles = it.nta * it.ntb;
    const int t_begin = (gw - it.tile_begin) * it.tpw;
    const int t_end = min(ntiles, t_begin + it.tpw);
    const cf* __restrict__ in = reinterpret_cast<const cf*>(it.in);
    const cf* __restrict__ X = reinterpret_cast<const cf*>(it.X);
    cf* __restrict__ out = reinterpret_cast<cf*>(it.out);
    for (int e = tid; e < NNP * KKP; e += 256) {
        int kk = e & (KKP - 1), nn = e / KKP;
        cf v; v.re = 0.f; v.im = 0.f;
        if (kk < KK && nn < NN) v = X[kk + (size_t)KK * nn];
        Xt_re[nn * PX + kk] = v.re; Xt_im[nn * PX + kk] = v.im;
    }
    for (int e = lane; e < 32 * PT; e += 64) { At_re[e] = 0.f; At_im[e] = 0.f; }
    __syncthreads();                                   // the only workgroup barrier: X^T is staged
    const TileMap mi = make_map_wave(lane, D, TA, TB, PA, K);
    const TileMap mo = make_map_wave(lane, Do, TA, TB, PA, No);
    const long long kstride_in = (long long)D * PA, kstride_out = (long long)Do * PA;
    const bool fast = mi.U <= 64 && mo.U <= 64 && (K + mi.KP - 1) / mi.KP <= NU;
    v4f pre[NU];
    auto tile_origin = [&](int t, int& a0, int& b0, int& na, int& nb) {
        int ta = t % it.nta, tb = t / it.nta;
        a0 = ta * TA; b0 = tb * TB; na = min(TA, it.PA - a0); nb = min(TB, it.PB - b0);
    };
    auto issue_loads = [&](int t) {
        int a0, b0, na, nb; tile_origin(t, a0, b0, na, nb);
        const long long org = (long long)D * (a0 + PA * (long long)K * b0);
        const bool v0 = mi.active && mi.al < na && mi.bl < nb, v1 = v0 && mi.al1 < na;
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            int k = mi.kp + mi.KP * j;
            v4f v; v[0] = v[1] = v[2] = v[3] = 0.f;
            if (k < K && v0) {
                const cf* p = in + org + mi.off + kstride_in * k;
                if (mi.vec == 2 && v1) v = ldg4(p);
                else { cf x = ldgc(p); v[0] = x.re; v[1] = x.im; }
            }
            pre[j] = v;
        }
    };
    auto commit_loads = [&]() {
        if (!mi.active) return;
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            int k = mi.kp + mi.KP * j;
            if (k < K) {
                int kk0 = mi.c0 + D * k;
                At_re[mi.row0 * PT + kk0] = pre[j][0]; At_im[mi.row0 * PT + kk0] = pre[j][1];
                if (mi.vec == 2) { int kk1 = mi.c1 + D * k; At_re[mi.row1 * PT + kk1] = pre[j][2]; At_im[mi.row1 * PT + kk1] = pre[j][3]; }
            }
        }
    };
    double nrm = 0;
    const int t_first = t_begin + w;
    if (fast && t_first < t_end) issue_loads(t_first);
    for (int t = t_first; t < t_end; t += 4) {
        int a0, b0, na, nb; tile_origin(t, a0, b0, na, nb);
        if (fast) commit_loads();
        else {
            const int ntile_el = D * TA * K * TB;
            for (int e = lane; e < ntile_el; e += 64) {
                int s = e % D; int r1 = e / D; int al = r1 % TA; int r2 = r1 / TA; int k = r2 % K; int bl = r2 / K;
                cf v; v.re = 0.f; v.im = 0.f;
                if (al < na && bl < nb) v = in[s + D * ((long long)(a0 + al) + PA * ((long long)k + (long long)K * (b0 + bl)))];
                int row = al + TA * bl;
                At_re[row * PT + s + D * k] = v.re; At_im[row * PT + s + D * k] = v.im;
            }
        }
        __builtin_amdgcn_wave_barrier();                 // LDS is in-order per wave; only the compiler must not reorder
        if (fast && t + 4 < t_end) issue_loads(t + 4);
        v16f Cr[NB], Ci[NB];
#pragma unroll
        for (int c = 0; c < NB; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) { Cr[c][r] = 0.f; Ci[c][r] = 0.f; }
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            float ar[16], ai[16];
            const float* pa_r = At_re + ln * PT + 32 * kb + 16 * h;
            const float* pa_i = At_im + ln * PT + 32 * kb + 16 * h;
#pragma unroll
            for (int q = 0; q < 16; ++q) { ar[q] = pa_r[q]; ai[q] = pa_i[q]; }
#pragma unroll
            for (int c = 0; c < NB; ++c) {
                float br[16], bi[16];
                const float* pb_r = Xt_re + (32 * c + ln) * PX + 32 * kb + 16 * h;
                const float* pb_i = Xt_im + (32 * c + ln) * PX + 32 * kb + 16 * h;
#pragma unroll
                for (int q = 0; q < 16; ++q) { br[q] = pb_r[q]; bi[q] = pb_i[q]; }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    Cr[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[q], br[q], Cr[c], 0, 0, 0);
                    Cr[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(-ai[q], bi[q], Cr[c], 0, 0, 0);
                    Ci[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[q], bi[q], Ci[c], 0, 0, 0);
                    Ci[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ai[q], br[q], Ci[c], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < NB; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                At_re[row * PT + 32 * c + ln] = Cr[c][r];
                At_im[row * PT + 32 * c + ln] = Ci[c][r];
            }
        __builtin_amdgcn_wave_barrier();
        if (fast) {
            const long long org = (long long)Do * (a0 + PA * (long long)No * b0);
            const bool v0 = mo.active && mo.al < na && mo.bl < nb, v1 = v0 && mo.al1 < na;
            if (v0) {
                for (int n = mo.kp; n < No; n += mo.KP) {
                    int nn0 = mo.c0 + Do * n;
                    v4f v;
                    v[0] = At_re[mo.row0 * PT + nn0]; v[1] = At_im[mo.row0 * PT + nn0];
                    cf* p = out + org + mo.off + kstride_out * n;
                    nrm += (double)v[0] * v[0] + (double)v[1] * v[1];
                    if (mo.vec == 2 && v1) {
                        int nn1 = mo.c1 + Do * n;
                        v[2] = At_re[mo.row1 * PT + nn1]; v[3] = At_im[mo.row1 * PT + nn1];
                        nrm += (double)v[2] * v[2] + (double)v[3] * v[3];
                        stg4(p, v);
                    } else { cf x; x.re = v[0]; x.im = v[1]; stgc(p, x); }
                }
            }
        } else {
            const int nout_el = Do * TA * No * TB;
            for (int e = lane; e < nout_el; e += 64) {
                int sp = e % Do; int r1 = e / Do; int al = r1 % TA; int r2 = r1 / TA; int n = r2 % No; int bl = r2 / No;
                if (al < na && bl < nb) {
                    int row = al + TA * bl, nn = sp + Do * n;
                    cf v; v.re = At_re[row * PT + nn]; v.im = At_im[row * PT + nn];
                    out[sp + Do * ((long long)(a0 + al) + PA * ((long long)n + (long long)No * (b0 + bl)))] = v;
                    nrm += (double)v.re * v.re + (double)v.im * v.im;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (NN > KK || KK < KKP) {          // results spilled into the operand padding columns: clear them again
            for (int e = lane; e < 32 * (CP - KK); e += 64) { int row = e / (CP - KK), c = KK + e % (CP - KK); At_re[row * PT + c] = 0.f; At_im[row * PT + c] = 0.f; }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (it.want_norm) {
        nrm = wave_sum_d(nrm);
        if (lane == 0) sh_red[w] = nrm;
        __syncthreads();
        if (tid == 0) norm_partials[gw] = sh_red[0] + sh_red[1] + sh_red[2] + sh_red[3];
    }
}

template <int KB, int NB, int TR> static size_t fiber_lds() {
    constexpr int KKP = 32 * KB, NNP = 32 * NB; constexpr int CP = (KKP > NNP ? KKP : NNP);
    return (size_t)(2 * TR * (CP + 1) + 2 * NNP * (KKP + 1)) * sizeof(float);
}
static int g_wave_private = -1;
static bool wave_private() {
    if (g_wave_private < 0) {
        g_wave_private = 1;
#ifdef TNQS_EXPERIMENTS
        const char* e = getenv("TNQS_MFMA_WG_TILES"); if (e && e[0] == '1') g_wave_private = 0;
#endif
    }
    return g_wave_private == 1;
}
int mfma_fiber_tile_rows(int KK, int NN) {
    if (wave_private()) return (KK <= 64 && NN <= 64) ? 32 : 0;
    if (KK <= 32 && NN <= 32) return 128;
    if (KK <= 64 && NN <= 64) return 64;
    return 0;
}
// returns false if the shape is not covered (caller falls back to the generic kernel)
bool launch_mfma_fiber_gemm(hipStream_t s, const FiberItem* d_items, int nitems, int total_tiles, int KKmax, int NNmax,
                            double* d_norm_partials) {
    if (total_tiles <= 0) return true;
    if (wave_private()) {
        if (KKmax <= 32 && NNmax <= 32) {
            const size_t lds = fiber_lds<1, 1, 128>();
            hipLaunchKernelGGL((mfma_fiber_gemm_w_kernel<1, 1, 8>), dim3(total_tiles), dim3(256), lds, s, d_items, nitems, d_norm_partials); TNQS_CHECK_LAUNCH();
            return true;
        }
        if (KKmax <= 64 && NNmax <= 64) {
            const size_t lds = fiber_lds<2, 2, 128>();
            set_max_dynamic_lds((const void*)mfma_fiber_gemm_w_kernel<2, 2, 16>, (size_t)lds);
            hipLaunchKernelGGL((mfma_fiber_gemm_w_kernel<2, 2, 16>), dim3(total_tiles), dim3(256), lds, s, d_items, nitems, d_norm_partials); TNQS_CHECK_LAUNCH();
            return true;
        }
        return false;
    }
    if (KKmax <= 32 && NNmax <= 32) {
        const size_t lds = fiber_lds<1, 1, 128>();
        hipLaunchKernelGGL((mfma_fiber_gemm_kernel<1, 1, 128>), dim3(total_tiles), dim3(256), lds, s, d_items, nitems, d_norm_partials); TNQS_CHECK_LAUNCH();
        return true;
    }
    if (KKmax <= 64 && NNmax <= 64) {
        const size_t lds = fiber_lds<2, 2, 64>();
        set_max_dynamic_lds((const void*)mfma_fiber_gemm_kernel<2, 2, 64>, (size_t)lds);
        hipLaunchKernelGGL((mfma_fiber_gemm_kernel<2, 2, 64>), dim3(total_tiles), dim3(256), lds, s, d_items, nitems, d_norm_partials); TNQS_CHECK_LAUNCH();
        return true;
    }
    return false;
}

// ------------------------------------------------------------------------------------------------------------
// Gram (f32 accumulate):  partial[4*c + w][i + KK*j] = sum_{rows of chunk c handled by wave w} X[i,row] conj(Y[j,row])
// KK = D*K <= 32.  Tiles of 64 fibers, LDS layout [kk][row] (rows contiguous = memory order); wave w takes 16 rows of
// every tile; the next tile's loads are in flight during the MFMA block.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mfma_gram32_kernel(const GramItem* __restrict__ items, int nitems) {
    constexpr int TR = 64, TRP = TR + 4, NU = 8;
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
    const long long PA = it.PA;
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
    for (int e = tid; e < 32 * TRP; e += 256) { Xr[e] = 0.f; Xi[e] = 0.f; Yr[e] = 0.f; Yi[e] = 0.f; }
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
                if (m.vec == 2 && v1) { vx = ldg4(Xg + o); vy = same ? vx : ldg4(Yg + o); }
                else { cf x = ldgc(Xg + o); vx[0] = x.re; vx[1] = x.im; if (same) vy = vx; else { cf y = ldgc(Yg + o); vy[0] = y.re; vy[1] = y.im; } }
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
// ------------------------------------------------------------------------------------------------------------
// fused last mode product + Gram (BP message epilogue):
//      out[b,b'] = sum_{s,j,rest} ( sum_i X[b,(s,i),rest] M[i,j] ) conj(Y[b',(s,j),rest])
// A wave tile is 64 fibers = (s: 2) x (i: 32) of the first row leg, so both GEMMs of a plane chain in registers:
//   step 1  C1[j][b] = sum_i M[i][j] X[b][s+2i]        (A = M^T held in registers, B = X from LDS)
//   step 2  out[b][b'] += sum_j C1[j][b] conj Y[b'][s+2j]   (A = C1's accumulator registers as they are: the C layout
//           row (r&3)+8(r>>2)+4h is exactly the k index instruction r consumes; B = Y from LDS at those rows)
// One LDS slab per wave is used for X, then for Y; the next tensor's global loads fly during each MFMA block.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mfma_gram32_fused_kernel(const GramItem* __restrict__ items, int nitems) {
    constexpr int PITCH = 65, NU = 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, ln = lane & 31, h = lane >> 5;
    float* Lr = reinterpret_cast<float*>(smem) + w * (2 * 32 * PITCH);
    float* Li = Lr + 32 * PITCH;
    int lo = 0, hi = nitems - 1;
    const int gc = blockIdx.x;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (items[mid].chunk_begin <= gc) lo = mid; else hi = mid - 1; }
    const GramItem it = items[lo];
    const int lc = gc - it.chunk_begin;
    const int K = it.K, TA = it.TA, TB = it.TB;            // D == 1, TA*TB == 64, K <= 32
    const long long PA = it.PA;
    const cf* __restrict__ Xg = reinterpret_cast<const cf*>(it.X);
    const cf* __restrict__ Yg = reinterpret_cast<const cf*>(it.Y);
    const cf* __restrict__ Mg = reinterpret_cast<const cf*>(it.M);
    const int ntiles = it.nta * it.ntb;
    const int t_begin = lc * it.tiles_per_chunk;
    const int t_end = min(ntiles, t_begin + it.tiles_per_chunk);
    // A operand of step 1: M^T, lane (j = ln, h), k-step q -> i = q + 16h
    float mr[16], mi[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) { cf v = Mg[(q + 16 * h) + 32 * ln]; mr[q] = v.re; mi[q] = v.im; }
    for (int e = lane; e < 32 * PITCH; e += 64) { Lr[e] = 0.f; Li[e] = 0.f; }      // rows kk >= K stay zero
    v16f Or, Oi;
#pragma unroll
    for (int r = 0; r < 16; ++r) { Or[r] = 0.f; Oi[r] = 0.f; }
    const TileMap m = make_map_wave(lane, 1, TA, TB, PA, K);
    const long long kstride = PA;
    v4f pre[NU];
    auto issue = [&](const cf* __restrict__ G, int t) {
        int ta = t % it.nta, tb = t / it.nta;
        int a0 = ta * TA, b0 = tb * TB;
        const long long org = a0 + PA * (long long)K * b0;
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            int k = m.kp + m.KP * j;
            v4f v; v[0] = v[1] = v[2] = v[3] = 0.f;
            if (k < K) v = ldg4(G + org + m.off + kstride * k);
            pre[j] = v;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            int k = m.kp + m.KP * j;
            if (k < K) { int o0 = k * PITCH + m.row0; Lr[o0] = pre[j][0]; Li[o0] = pre[j][1]; Lr[o0 + 1] = pre[j][2]; Li[o0 + 1] = pre[j][3]; }
        }
    };
    const int t_first = t_begin + w;
    if (t_first < t_end) issue(Xg, t_first);
    for (int t = t_first; t < t_end; t += 4) {
        commit();                                           // X tile -> LDS
        __builtin_amdgcn_wave_barrier();
        issue(Yg, t);                                       // Y tile in flight during step 1
        v16f C1r[2], C1i[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float xr[16], xi[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) { int o = ln * PITCH + s + 2 * (q + 16 * h); xr[q] = Lr[o]; xi[q] = Li[o]; }
#pragma unroll
            for (int r = 0; r < 16; ++r) { C1r[s][r] = 0.f; C1i[s][r] = 0.f; }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                C1r[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(mr[q], xr[q], C1r[s], 0, 0, 0);
                C1r[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(-mi[q], xi[q], C1r[s], 0, 0, 0);
                C1i[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(mr[q], xi[q], C1i[s], 0, 0, 0);
                C1i[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(mi[q], xr[q], C1i[s], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
        commit();                                           // Y tile -> the same LDS slab (X operands are in registers)
        __builtin_amdgcn_wave_barrier();
        if (t + 4 < t_end) issue(Xg, t + 4);                // next X tile in flight during step 2
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float yr[16], yi[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { int j = (r & 3) + 8 * (r >> 2) + 4 * h; int o = ln * PITCH + s + 2 * j; yr[r] = Lr[o]; yi[r] = Li[o]; }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                Or = __builtin_amdgcn_mfma_f32_32x32x2f32(C1r[s][r], yr[r], Or, 0, 0, 0);
                Or = __builtin_amdgcn_mfma_f32_32x32x2f32(C1i[s][r], yi[r], Or, 0, 0, 0);
                Oi = __builtin_amdgcn_mfma_f32_32x32x2f32(C1i[s][r], yr[r], Oi, 0, 0, 0);
                Oi = __builtin_amdgcn_mfma_f32_32x32x2f32(-C1r[s][r], yi[r], Oi, 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    cf* __restrict__ part = reinterpret_cast<cf*>(it.partial) + (size_t)(4 * lc + w) * K * K;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * h, j = ln;
        if (i < K && j < K) { cf v; v.re = Or[r]; v.im = Oi[r]; part[i + (size_t)K * j] = v; }
    }
}
void launch_mfma_gram32_fused(hipStream_t s, const GramItem* d_items, int nitems, int total_chunks) {
    if (total_chunks <= 0) return;
    const size_t lds = (size_t)4 * 2 * 32 * 65 * sizeof(float);
    set_max_dynamic_lds((const void*)mfma_gram32_fused_kernel, lds);
    hipLaunchKernelGGL(mfma_gram32_fused_kernel, dim3(total_chunks), dim3(256), lds, s, d_items, nitems); TNQS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// pair of mode products on two legs x, y (dimension 32 each) in ONE pass:
//      out[c, jx, jy] = sum_{ix,iy} in[c, ix, iy] Mx[ix, jx] My[iy, jy]          for every companion index c
// A slice is 16 companions (PairGeom: 128-byte runs) x the 32 x 32 plane of the two legs; a workgroup walks HALF slices (8 companions,
// one plane per wave; workgroups lw and lw + 8 of a group of 16 sit on the same XCD and take the two 64-byte halves of every line at the
// same time).  Both GEMMs of a plane are chained in registers:
//   step 1  Y[iy][jx] = sum_ix S[ix][iy] Mx[ix][jx]      (A = S^T from LDS, four updates ahead of their use; B = Mx in registers)
//   step 2  S'[jx][jy] = sum_iy Y[iy][jx] My[iy][jy]      (A = Y's accumulator registers as they are, B = My in registers)
// with Gauss' three-multiplication complex product (CAcc32).  The planes are double-buffered in LDS (2 x 8 x 8.3 KiB): a phase computes
// its plane in place, and the NEXT phase moves the result out (LDS -> global, one 16-byte store every six MFMAs of its step 2) while the
// loads of the phase after it arrive (one every six MFMAs of step 1) and are committed over the locations the same thread just stored
// from -- so one barrier per phase orders everything.
// ------------------------------------------------------------------------------------------------------------
// XCD-aware workgroup order: consecutive workgroup ids go round-robin over the 8 XCDs, so XCD x sees ids x, x+8, ...  Remapping id ->
// (id % 8) * (n/8) + id / 8 gives every XCD one contiguous range of the work list (neighbouring slices share DRAM pages and L2 sets).
__device__ __forceinline__ int xcd_remap(int id, int n, int mode) {
    if (!mode) return id;
    const int n8 = (n >> 3) << 3;
    return id < n8 ? (id & 7) * (n8 >> 3) + (id >> 3) : id;
}
__device__ __forceinline__ long long pair_slice_base(const PairGeom& g, int sl) {
    int a0 = sl % g.n0; int r1 = sl / g.n0; int a1 = r1 % g.n1; int a2 = r1 / g.n1;
    return (long long)a0 * g.t0 + (long long)a1 * g.t1 + (long long)a2 * g.t2;
}
template <bool M3>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void mfma_pair_kernel(const PairItem* __restrict__ items, int nitems) {
    constexpr int PS = 32 * 33 + 4;            // plane stride in complex elements, pitch 33 (rows and columns both conflict-free)
    constexpr int BUF = 8 * PS;
    #define TNQS_PIN() __builtin_amdgcn_sched_barrier(0x2 | 0x4)      // VALU / SALU may cross; LDS, global and matrix instructions keep their order
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2f* L = reinterpret_cast<v2f*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, ln = lane & 31, h = lane >> 5;
    int lo = 0, hi_ = nitems - 1;
    const int gw = blockIdx.x;
    while (lo < hi_) { int mid = (lo + hi_ + 1) >> 1; if (items[mid].slice_begin <= gw) lo = mid; else hi_ = mid - 1; }
    const PairItem it = items[lo];
    const PairGeom g = it.g;
    const int nslices = g.n0 * g.n1 * g.n2;
    const int lw = gw - it.slice_begin;                     // slice_begin is a multiple of 16
    const int half = (lw >> 3) & 1, pw = ((lw >> 4) << 3) | (lw & 7);
    const int s_begin = pw * it.spw, s_end = min(nslices, s_begin + it.spw);
    const cf* __restrict__ in = reinterpret_cast<const cf*>(it.in);
    cf* __restrict__ out = reinterpret_cast<cf*>(it.out);
    // B operands: Mx[k = q+16h][j = ln] (step 1), My[k = kappa(r,h)][j = ln] (step 2); element (i,j) at i + 32 j
    float mxr[16], mxi[16], myr[16], myi[16];
    {
        const cf* __restrict__ Mx = reinterpret_cast<const cf*>(it.Mx) + 32 * ln; const cf* __restrict__ My = reinterpret_cast<const cf*>(it.My) + 32 * ln;
#pragma unroll
        for (int q = 0; q < 16; q += 2) { const v4f v = ldg4(Mx + q + 16 * h); mxr[q] = v[0]; mxi[q] = v[1]; mxr[q + 1] = v[2]; mxi[q + 1] = v[3]; }
#pragma unroll
        for (int q = 0; q < 16; q += 4) {                   // kappa(q..q+3, h) = 2 q + 4 h + (0..3): four consecutive elements
            const v4f u = ldg4(My + 2 * q + 4 * h), v = ldg4(My + 2 * q + 4 * h + 2);
            myr[q] = u[0]; myi[q] = u[1]; myr[q + 1] = u[2]; myi[q + 1] = u[3]; myr[q + 2] = v[0]; myi[q + 2] = v[1]; myr[q + 3] = v[2]; myi[q + 3] = v[3];
        }
    }
    // mover: thread -> (companion pair f4 = 0..3 of the half, first segment sg0 = 0..127); segment j: ix = sg0 & 31, iy = (sg0 >> 5) + 4 j
    const int f4 = tid & 3, sg0 = tid >> 2;
    const int ix0 = sg0 & 31, iy0 = sg0 >> 5;
    const long long toff = (long long)(4 * half + f4) * g.cstr + g.sx * ix0 + g.sy * iy0, tstr = 4 * g.sy;
    v2f* const lbase = L + (2 * f4) * PS + iy0 * 33 + ix0;           // element (ix, iy) of a plane at [iy][ix]; iy advances by 4 per j (132 elements)
    v4f pre[8];
    auto commit = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v2f* p0 = lbase + buf * BUF + 132 * j;
            v2f a; a[0] = pre[j][0]; a[1] = pre[j][1]; v2f b; b[0] = pre[j][2]; b[1] = pre[j][3];
            p0[0] = a; p0[PS] = b;
        }
    };
    auto store1 = [&](int buf, long long ob, int j) {            // one 16-byte piece of a finished phase: LDS -> global
        const v2f* p0 = lbase + buf * BUF + 132 * j;
        const v2f a = p0[0], c = p0[PS];
        v4f v; v[0] = a[0]; v[1] = a[1]; v[2] = c[0]; v[3] = c[1];
        stg4(out + ob + tstr * j, v);
    };
    if (s_begin < s_end) {
        const long long b = pair_slice_base(g, s_begin) + toff;
#pragma unroll
        for (int j = 0; j < 8; ++j) pre[j] = ldg4(in + b + tstr * j);
        commit(0);
    }
    lds_barrier();
    for (int sl = s_begin; sl < s_end; ++sl) {
        const int buf = (sl - s_begin) & 1;
        const bool more = sl + 1 < s_end, prev = sl > s_begin;
        const long long nb = pair_slice_base(g, more ? sl + 1 : sl) + toff;      // (the last phase re-reads its own slice: no branch in the stream)
        const long long ob = pair_slice_base(g, prev ? sl - 1 : sl) + toff;
        v2f* P = L + buf * BUF + w * PS;
        v2f xc[4], xn[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) xc[i] = P[ln * 33 + i + 16 * h];            // A[i = iy = ln][k = ix]
        // ---- step 1 ---------------------------------------------------------------------------------------------------------------
        CAcc32<M3> Y;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            if (c4 < 3) {
#pragma unroll
                for (int i = 0; i < 4; ++i) xn[i] = P[ln * 33 + 4 * (c4 + 1) + i + 16 * h];
            }
            TNQS_PIN();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = 4 * c4 + i;
                const float br = mxr[q], bi = mxi[q];
                if (q == 0) Y.template mac_bpre<true>(xc[i][0], xc[i][1], br, M3 ? bi - br : bi, M3 ? br + bi : -bi);
                else Y.template mac_bpre<false>(xc[i][0], xc[i][1], br, M3 ? bi - br : bi, M3 ? br + bi : -bi);
                if ((q & 1) == 0) { pre[q >> 1] = ldg4(in + nb + tstr * (q >> 1)); TNQS_PIN(); }      // next phase's loads, one every two updates
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) xc[i] = xn[i];
        }
        // ---- step 2 (and the previous phase's plane on its way out) -----------------------------------------------------------------
        CAcc32<M3> S;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float br = myr[r], bi = myi[r];
            if (r == 0) S.template mac_bpre<true>(Y.re(r), Y.im(r), br, M3 ? bi - br : bi, M3 ? br + bi : -bi);
            else S.template mac_bpre<false>(Y.re(r), Y.im(r), br, M3 ? bi - br : bi, M3 ? br + bi : -bi);
            if ((r & 1) == 0) { if (prev) store1(buf ^ 1, ob, r >> 1); TNQS_PIN(); }
        }
        S.finish();
        // S'[jx = kappa(r,h)][jy = ln] -> LDS [jy][jx] (the plane is private to this wave; its reads are complete)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const int jx = (r & 3) + 8 * (r >> 2) + 4 * h; v2f v; v[0] = S.a[r]; v[1] = S.b[r]; P[ln * 33 + jx] = v; }
        if (more) commit(buf ^ 1);                // over the locations this thread stored from above
        lds_barrier();
    }
    if (s_begin < s_end) {                         // the last phase's plane
        const int buf = (s_end - 1 - s_begin) & 1;
        const long long ob = pair_slice_base(g, s_end - 1) + toff;
#pragma unroll
        for (int j = 0; j < 8; ++j) store1(buf, ob, j);
    }
    #undef TNQS_PIN
}
void launch_mfma_pair(hipStream_t s, const PairItem* d_items, int nitems, int total_wgs) {
    if (total_wgs <= 0) return;
    if (mfma_use_x3()) { launch_x3_pair(s, d_items, nitems, total_wgs); return; }
    const size_t lds = (size_t)16 * (32 * 33 + 4) * 2 * sizeof(float);
    set_max_dynamic_lds((const void*)mfma_pair_kernel<true>, lds); hipLaunchKernelGGL(mfma_pair_kernel<true>, dim3(total_wgs), dim3(512), lds, s, d_items, nitems);
    TNQS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// last absorption + Gram on a 32 x 32 plane of two ARBITRARY legs (BP message epilogue reading a cached pair product X):
//      out[b,b'] = sum_{c, jx} ( sum_ix X[c, ix, b] M[ix, jx] ) conj(Y[c, jx, b'])        x = absorbed leg, y = kept leg
// Same workgroup shape and mover as the pair kernel (16 companions x the plane, one LDS slab); the slab holds the X planes
// for step 1 and is then refilled with the Y planes for step 2, the intermediate stays in the accumulator registers:
//   step 1  C1[jx][b] = sum_ix M[ix][jx] X[ix][b]            (A = M^T in registers, B = X plane from LDS)
//   step 2  out[b][b'] += sum_jx C1[jx][b] conj Y[jx][b']    (A = C1's accumulator registers as they are, B = Y plane)
// Each wave owns two planes and one 32 x 32 accumulator; it writes one partial per workgroup and wave.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void mfma_pair_gram_kernel(const PairGramItem* __restrict__ items, int nitems, int xcd) {
    constexpr int PS = 32 * 33 + 1;            // plane stride in complex elements ((re, im) pairs, ds_*_b64 accesses)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2f* L = reinterpret_cast<v2f*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, ln = lane & 31, h = lane >> 5;
    int lo = 0, hi_ = nitems - 1;
    const int gw = xcd_remap(blockIdx.x, gridDim.x, xcd);
    while (lo < hi_) { int mid = (lo + hi_ + 1) >> 1; if (items[mid].wg_begin <= gw) lo = mid; else hi_ = mid - 1; }
    const PairGramItem it = items[lo];
    const PairGeom g = it.g;
    const int nslices = g.n0 * g.n1 * g.n2;
    const int lw = gw - it.wg_begin;
    const int s_begin = lw * it.spw, s_end = min(nslices, s_begin + it.spw);
    const cf* __restrict__ Xg = reinterpret_cast<const cf*>(it.X);
    const cf* __restrict__ Yg = reinterpret_cast<const cf*>(it.Y);
    const cf* __restrict__ Mg = reinterpret_cast<const cf*>(it.M);
    float mr[16], mi[16];                      // A operand of step 1: M^T, lane (jx = ln, h), k-step q -> ix = q + 16h
#pragma unroll
    for (int q = 0; q < 16; ++q) { cf v = Mg[(q + 16 * h) + 32 * ln]; mr[q] = v.re; mi[q] = v.im; }
    v16f Or, Oi;
#pragma unroll
    for (int r = 0; r < 16; ++r) { Or[r] = 0.f; Oi[r] = 0.f; }
    const int f = tid & 7, sg0 = tid >> 3;
    const long long sx = g.sx, sy = g.sy, fo = (long long)f * g.cstr;
    v4f pre[16];
    const int ix0 = sg0 & 31, iy0 = sg0 >> 5;            // segment j: ix = ix0, iy = iy0 + 2 j
    const long long toff = fo + sx * ix0 + sy * iy0, tstr = 2 * sy;
    v2f* const lbase = L + (2 * f) * PS + iy0 * 33 + ix0;
    auto issue = [&](const cf* __restrict__ G, int sl) {
        const cf* p = G + pair_slice_base(g, sl) + toff;
#pragma unroll
        for (int j = 0; j < 16; ++j) pre[j] = ldg4(p + tstr * j);
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            v2f* p0 = lbase + 66 * j;                            // plane 2f: element (ix, iy) stored at [iy][ix]; plane 2f+1 one stride further
            v2f a; a[0] = pre[j][0]; a[1] = pre[j][1]; v2f b; b[0] = pre[j][2]; b[1] = pre[j][3];
            p0[0] = a; p0[PS] = b;
        }
    };

    if (s_begin < s_end) issue(Xg, s_begin);
    for (int sl = s_begin; sl < s_end; ++sl) {
        lds_barrier();                                          // the previous slice's Y planes have been consumed
        commit();                                               // X planes
        lds_barrier();
        issue(Yg, sl);                                          // Y in flight during step 1
        v16f C1r[2], C1i[2];
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const v2f* P = L + (w + 8 * pp) * PS;
            float xr[16], xi[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) { v2f v = P[ln * 33 + q + 16 * h]; xr[q] = v[0]; xi[q] = v[1]; }     // B[k=ix][j=b=ln]
#pragma unroll
            for (int r = 0; r < 16; ++r) { C1r[pp][r] = 0.f; C1i[pp][r] = 0.f; }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                C1r[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(mr[q], xr[q], C1r[pp], 0, 0, 0);
                C1r[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(-mi[q], xi[q], C1r[pp], 0, 0, 0);
                C1i[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(mr[q], xi[q], C1i[pp], 0, 0, 0);
                C1i[pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(mi[q], xr[q], C1i[pp], 0, 0, 0);
            }
        }
        lds_barrier();                                          // every wave has read its X operands
        commit();                                               // Y planes into the same slab
        lds_barrier();
        if (sl + 1 < s_end) issue(Xg, sl + 1);                  // next X in flight during step 2
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const v2f* P = L + (w + 8 * pp) * PS;
            float yr[16], yi[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { int jx = (r & 3) + 8 * (r >> 2) + 4 * h; v2f v = P[ln * 33 + jx]; yr[r] = v[0]; yi[r] = v[1]; }   // B[k=jx][j=b'=ln]
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                Or = __builtin_amdgcn_mfma_f32_32x32x2f32(C1r[pp][r], yr[r], Or, 0, 0, 0);
                Or = __builtin_amdgcn_mfma_f32_32x32x2f32(C1i[pp][r], yi[r], Or, 0, 0, 0);
                Oi = __builtin_amdgcn_mfma_f32_32x32x2f32(C1i[pp][r], yr[r], Oi, 0, 0, 0);
                Oi = __builtin_amdgcn_mfma_f32_32x32x2f32(-C1r[pp][r], yi[r], Oi, 0, 0, 0);
            }
        }
    }
    // one partial per workgroup (see mfma_pair_gram2_kernel)
    cf* __restrict__ part = reinterpret_cast<cf*>(it.partial) + (size_t)lw * 1024;
    v2f* const R = L;
    lds_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        v2f v; v[0] = Or[r]; v[1] = Oi[r];
        R[w * (32 * 33) + ln * 33 + i] = v;
    }
    lds_barrier();
    for (int e = threadIdx.x; e < 1024; e += 512) {
        const int i = e & 31, j = e >> 5;
        float sr = 0.f, si = 0.f;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) { const v2f v = R[ww * (32 * 33) + j * 33 + i]; sr += v[0]; si += v[1]; }
        cf o; o.re = sr; o.im = si; part[e] = o;
    }
}
void launch_mfma_pair_gram(hipStream_t s, const PairGramItem* d_items, int nitems, int total_wgs) {
    if (total_wgs <= 0) return;
    const size_t lds = (size_t)16 * (32 * 33 + 1) * 2 * sizeof(float);
    set_max_dynamic_lds((const void*)mfma_pair_gram_kernel, lds);
    static int xcd = -1;
    if (xcd < 0) {
        xcd = 0;
#ifdef TNQS_EXPERIMENTS
        if (const char* e = std::getenv("TNQS_XCD_REMAP")) xcd = std::atoi(e);
#endif
    }
    hipLaunchKernelGGL(mfma_pair_gram_kernel, dim3(total_wgs), dim3(512), lds, s, d_items, nitems, xcd); TNQS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// BOTH messages a degree-4 site sends into a linear forest, from one pass over the shared pair product X and psi = Y:
//      out_y[b,b'] = sum_{c,jx} ( sum_ix X[c,ix,b] Mx[ix,jx] ) conj Y[c,jx,b']      (kept leg y, leg x absorbed)
//      out_x[d,d'] = sum_{c,jy} ( sum_iy X[c,d,iy] My[iy,jy] ) conj Y[c,d',jy]      (kept leg x, leg y absorbed)
// 64 algorithmic flop per byte: matrix-core bound.  QUARTER slices: a workgroup takes 4 of a slice's 16 companions per phase (workgroups
// lw, lw+8, lw+16, lw+24 of a group of 32 sit on the same XCD and walk the same slices at the same time, so the other three 32-byte
// quarters of every line are L2 hits), and its eight waves are (companion c = w & 3) x (message m = w >> 2): a wave carries ONE output
// accumulator, which leaves the registers for Gauss' three-multiplication product (CAcc32<true>: 3 accumulators per complex tile) on both
// steps,   step 1  C[j'][kept] = sum_k M[k][j'] X[k][kept]     step 2  O[kept][kept'] += sum_j' C[j'][kept] conj Y[j'][kept']
// with C's accumulator registers as step 2's A operand.  A phase's planes (4 X + 4 Y = 66 KiB) are double-buffered in LDS: the next
// phase's planes are loaded to registers while the matrix cores work and written into the other buffer at the end of the phase -- one
// barrier per phase, no dead time between barriers.  The wave's matrix (Mx or My, by its message) lives in REGISTERS as the A operands
// of step 1 (m_r + m_i, m_r, m_i: 48 registers, no LDS traffic and no VALU work for them); the plane operands are read from LDS four
// updates ahead of their use (explicit register double buffer, pinned by scheduling barriers -- left to itself the compiler issues each
// ds_read directly in front of the MFMA that needs it); the eight global loads of the next phase are spread over the first half of the
// phase, one every six MFMAs (a burst of eight 32-lines-per-instruction loads at the phase start stalled the issue: 129 -> 143 TFLOP/s).
// Measured on 100 device-resident sites (profiles/plane_bench.py): 147 TFLOP/s algorithmic (8 flop per complex multiply-add), against
// 129 for the same kernel with LDS-resident matrices and compiler-scheduled reads and 114 for the round-2 half-slice kernel it replaces.
// M3 = false (TNQS_NO_3M=1): the same schedule with the four-multiplication product.
// ------------------------------------------------------------------------------------------------------------
template <bool M3>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void mfma_pair_gram2_kernel(const PairGram2Item* __restrict__ items, int nitems) {
    constexpr int PS = 32 * 33 + 8;            // plane stride in complex elements
    constexpr int BUF = 8 * PS;                // one phase: X planes of companions 0..3, then their Y planes
    // scheduling barrier: VALU / SALU may cross, LDS, global-memory and matrix instructions keep their program order
    #define TNQS_PIN() __builtin_amdgcn_sched_barrier(0x2 | 0x4)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v2f* L = reinterpret_cast<v2f*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, ln = lane & 31, h = lane >> 5;
    int lo = 0, hi_ = nitems - 1;
    const int gw = blockIdx.x;
    while (lo < hi_) { int mid = (lo + hi_ + 1) >> 1; if (items[mid].wg_begin <= gw) lo = mid; else hi_ = mid - 1; }
    const PairGram2Item it = items[lo];
    const PairGeom g = it.g;
    const int nslices = g.n0 * g.n1 * g.n2;
    const int lw = gw - it.wg_begin;                        // wg_begin is a multiple of 32
    const int quarter = (lw >> 3) & 3, pw = ((lw >> 5) << 3) | (lw & 7);
    const int s_begin = pw * it.spw, s_end = min(nslices, s_begin + it.spw);
    const cf* __restrict__ Xg = reinterpret_cast<const cf*>(it.X);
    const cf* __restrict__ Yg = reinterpret_cast<const cf*>(it.Y);
    const int comp = w & 3, msg = w >> 2;
    // A operands of step 1: A[i = ln][k = q + 16 h] = M[(q + 16 h) + 32 ln], 16 consecutive complex numbers per lane
    float m0[16], mr[16], mi[16];                            // m0: the A-side combination of CAcc32::mac_pre
    {
        const cf* __restrict__ M = reinterpret_cast<const cf*>(msg ? it.My : it.Mx) + 16 * h + 32 * ln;
#pragma unroll
        for (int q = 0; q < 16; q += 2) { const v4f v = ldg4(M + q); mr[q] = v[0]; mi[q] = v[1]; mr[q + 1] = v[2]; mi[q + 1] = v[3]; }
#pragma unroll
        for (int q = 0; q < 16; ++q) m0[q] = M3 ? mr[q] + mi[q] : -mi[q];
    }
    // mover: thread -> (companion pair f2 = 0..1 of the quarter, first segment sg0 = 0..255); segment j: ix = sg0 & 31, iy = (sg0 >> 5) + 8 j
    const int f2 = tid & 1, sg0 = tid >> 1;
    const int ix0 = sg0 & 31, iy0 = sg0 >> 5;
    const long long toff = (long long)(2 * quarter + f2) * g.cstr + g.sx * ix0 + g.sy * iy0, tstr = 8 * g.sy;
    v2f* const lbase = L + (2 * f2) * PS + iy0 * 33 + ix0;
    v4f px[4], py[4];
    auto commit = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v2f* p0 = lbase + buf * BUF + 264 * j;                // element (ix, iy) at [iy][ix]; iy advances by 8 per j
            v2f a; a[0] = px[j][0]; a[1] = px[j][1]; v2f b; b[0] = px[j][2]; b[1] = px[j][3];
            p0[0] = a; p0[PS] = b;
            v2f c; c[0] = py[j][0]; c[1] = py[j][1]; v2f d; d[0] = py[j][2]; d[1] = py[j][3];
            p0[4 * PS] = c; p0[5 * PS] = d;
        }
    };
    CAcc32<M3> O; O.zero();
    if (s_begin < s_end) {
        const long long b = pair_slice_base(g, s_begin) + toff;
#pragma unroll
        for (int j = 0; j < 4; ++j) { px[j] = ldg4(Xg + b + tstr * j); py[j] = ldg4(Yg + b + tstr * j); }
        commit(0);
    }
    lds_barrier();
    // the phase loop, once per message (the plane operands are read as rows or as columns): waves 0..3 and 4..7 run different copies
    auto run = [&](auto msg_c) {
    constexpr int MSG = decltype(msg_c)::value;
    for (int sl = s_begin; sl < s_end; ++sl) {
        const int buf = (sl - s_begin) & 1;
        const bool more = sl + 1 < s_end;
        const v2f* PX = L + buf * BUF + comp * PS; const v2f* PY = PX + 4 * PS;
        const long long nb = pair_slice_base(g, more ? sl + 1 : sl) + toff;         // (the last phase re-reads its own slice: no branch in the stream)
        // plane operand k of the kept index ln: message 0 reads rows ([ln][k]), message 1 the same planes transposed ([k][ln])
        auto xat = [&](int k) { return MSG ? PX[k * 33 + ln] : PX[ln * 33 + k]; };
        auto yat = [&](int k) { return MSG ? PY[k * 33 + ln] : PY[ln * 33 + k]; };
        v2f xc[4], xn[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) xc[i] = xat(i + 16 * h);
        // ---- step 1: C[j'][kept] = sum_k M[k][j'] X[k][kept] ---------------------------------------------------------------
        CAcc32<M3> C;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (c4 < 3) xn[i] = xat(4 * (c4 + 1) + i + 16 * h);
                else { const int r = i, k = (r & 3) + 8 * (r >> 2) + 4 * h; xn[i] = yat(k); }      // first four operands of step 2
            }
            TNQS_PIN();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = 4 * c4 + i;
                if (q == 0) C.template mac_pre<true>(m0[q], mr[q], mi[q], xc[i][0], xc[i][1]);
                else C.template mac_pre<false>(m0[q], mr[q], mi[q], xc[i][0], xc[i][1]);
                if ((q & 1) == 0) {                                   // one global load of the next phase every two updates
                    const int j = q >> 2;
                    if ((q & 2) == 0) px[j] = ldg4(Xg + nb + tstr * j); else py[j] = ldg4(Yg + nb + tstr * j);
                    TNQS_PIN();
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) xc[i] = xn[i];
        }
        // ---- step 2: O[kept][kept'] += sum_j' C[j'][kept] conj Y[j'][kept'] ----------------------------------------------------
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            if (c4 < 3) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { const int r = 4 * (c4 + 1) + i, k = (r & 3) + 8 * (r >> 2) + 4 * h; xn[i] = yat(k); }
            }
            TNQS_PIN();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * c4 + i;
                O.mac_conj(C.re(r), C.im(r), xc[i][0], xc[i][1]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) xc[i] = xn[i];
        }
        if (more) commit(buf ^ 1);                                      // the other buffer was consumed before the previous barrier
        lds_barrier();
    }
    };
    if (msg == 0) run(std::integral_constant<int, 0>{}); else run(std::integral_constant<int, 1>{});
    // one partial per workgroup and message: the four waves of a message are summed through the (now free) buffers
    O.finish_conj();
    cf* __restrict__ p1 = reinterpret_cast<cf*>(it.partial_y) + (size_t)lw * 1024;
    cf* __restrict__ p2 = reinterpret_cast<cf*>(it.partial_x) + (size_t)lw * 1024;
    v2f* const R = L;                                           // 8 blocks of 32 x 33
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        v2f v; v[0] = O.a[r]; v[1] = O.b[r];
        R[w * (32 * 33) + ln * 33 + i] = v;                     // element (i, j = ln)
    }
    lds_barrier();
    for (int e = tid; e < 2048; e += 512) {
        const int mm = e >> 10, ee = e & 1023, i = ee & 31, j = ee >> 5;
        float sr = 0.f, si = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) { const v2f v = R[(4 * mm + ww) * (32 * 33) + j * 33 + i]; sr += v[0]; si += v[1]; }
        cf o; o.re = sr; o.im = si; stgc((mm ? p2 : p1) + ee, o);
    }
    #undef TNQS_PIN
}
int x3_pair_gram2_group();
int pair_gram2_group() { return mfma_use_x3() ? x3_pair_gram2_group() : 32; }
void launch_mfma_pair_gram2(hipStream_t s, const PairGram2Item* d_items, int nitems, int total_wgs) {
    if (total_wgs <= 0) return;
    if (mfma_use_x3()) { launch_x3_pair_gram2(s, d_items, nitems, total_wgs); return; }
    const size_t lds = (size_t)16 * (32 * 33 + 8) * 2 * sizeof(float);
    set_max_dynamic_lds((const void*)mfma_pair_gram2_kernel<true>, lds);
    hipLaunchKernelGGL(mfma_pair_gram2_kernel<true>, dim3(total_wgs), dim3(512), lds, s, d_items, nitems);
    TNQS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// right singular vectors of the theta SVD, recovered: V = theta0^dagger (U Sigma) Sigma^-2  -- an n x n x m complex GEMM
// per gate (m, n <= 256).  A wave owns one 32 x 32 tile of V; operands are read straight from L2 (both matrices are <= 512 KiB).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void recover_v_mfma_kernel(const RecoverItem* __restrict__ items) {
    const RecoverItem it = items[blockIdx.x];
    if (it.pre && theta_pre_takes(it.dyn, it.dm, it.dn, it.pre == 2)) return;      // V already written by theta_svd_pre_kernel
    const cf* __restrict__ A0 = reinterpret_cast<const cf*>(it.A0);
    const cf* __restrict__ A = reinterpret_cast<const cf*>(it.A);
    cf* __restrict__ V = reinterpret_cast<cf*>(it.V);
    int m_ = it.m, n_ = it.n, nu_ = it.nu;
    if (it.dyn) theta_dims(it.dyn, it.dm, it.dn, m_, n_, nu_);      // dimensions found on the device (RecoverItem::dyn)
    const int m = m_, n = n_, nu = nu_;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, ln = lane & 31, h = lane >> 5;
    const int nt = (n + 31) >> 5;
    const int tile = blockIdx.y * 4 + w;
    if (tile >= nt * nt || n < 1 || nu < 1) return;
    const int c0 = 32 * (tile % nt), u0 = 32 * (tile / nt);
    const int col = c0 + ln, u = u0 + ln;
    const cf* pa = A0 + (size_t)m * min(col, n - 1);
    const cf* pb = A + (size_t)m * min(u, nu - 1);
    if (u0 >= nu) return;
    const bool okc = col < n, oku = u < nu;
    v16f Cr, Ci;
#pragma unroll
    for (int r = 0; r < 16; ++r) { Cr[r] = 0.f; Ci[r] = 0.f; }
    double s2 = 0.0;                                          // |a_u|^2 (rows split over the two half waves); double: squares of small f32 columns underflow
    for (int k0 = 0; k0 < m; k0 += 2) {
        const int row = k0 + h;
        cf a = {0.f, 0.f}, b = {0.f, 0.f};
        if (row < m) { if (okc) a = pa[row]; if (oku) b = pb[row]; }
        s2 += (double)b.re * b.re + (double)b.im * b.im;
        // C[col][u] += conj(a) * b
        Cr = __builtin_amdgcn_mfma_f32_32x32x2f32(a.re, b.re, Cr, 0, 0, 0);
        Cr = __builtin_amdgcn_mfma_f32_32x32x2f32(a.im, b.im, Cr, 0, 0, 0);
        Ci = __builtin_amdgcn_mfma_f32_32x32x2f32(a.re, b.im, Ci, 0, 0, 0);
        Ci = __builtin_amdgcn_mfma_f32_32x32x2f32(-a.im, b.re, Ci, 0, 0, 0);
    }
    s2 += __shfl_xor(s2, 32, 64);                             // lane ln (and ln+32) now hold |a_{u0+ln}|^2
    const double inv = s2 > 0.0 ? 1.0 / s2 : 0.0;
    if (oku) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * h;      // C[row = col index][col = u = ln]
            if (c < n) { cf v; v.re = (float)(Cr[r] * inv); v.im = (float)(Ci[r] * inv); V[c + (size_t)n * u] = v; }
        }
    }
}
void launch_recover_v_mfma(hipStream_t s, const RecoverItem* d_items, int nitems, int nmax) {
    if (nitems <= 0) return;
    const int nt = (nmax + 31) / 32;
    hipLaunchKernelGGL(recover_v_mfma_kernel, dim3(nitems, (nt * nt + 3) / 4), dim3(256), 0, s, d_items); TNQS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// Gram with f64 accumulation on v_mfma_f64_16x16x4_f64 (gate path: G = X X^dagger over the fibers, KK = D*K <= 64,
// ComplexF32 input converted on the fly; the f32 products are exact in f64, so G is the exact Gram of the rounded
// tensor -- what the eigen factorisation replacing the thin QR needs).  f64 MFMA layout: A[i=l&15][k=l>>4],
// B[k=l>>4][j=l&15], C[row=(l>>4)+4r][col=l&15].  G is Hermitian: only the 16 x 16 blocks (I, J >= I) are computed (ten for a
// 64 x 64 G) and mirrored when the partials are written.  The blocks are dealt round-robin to the four waves, forwards on even
// tiles and backwards on odd tiles, so every wave (= SIMD) does 5 blocks per two tiles instead of 8; the two tile parities
// accumulate into separate partials (2 per chunk) because a block changes wave with the parity.
// ------------------------------------------------------------------------------------------------------------
template <bool M3, bool SHARED>         // SHARED: every item of the launch has KK = 64 (four panels): blocks dealt in panel-sharing sets
__global__ __launch_bounds__(256, 2) void mfma_gram64_f64_kernel(const GramItem* __restrict__ items, int nitems, int dbg_skip) {
    constexpr int TR = 64, TRP = TR + 4, NU = 8;
    // two tile buffers (re, im planes each): the next tile is committed while other waves still multiply the current one
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const Xbuf = reinterpret_cast<float*>(smem);          // [buf][re|im][64 * TRP]
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
    // upper-triangle blocks in row-major order, dealt to the waves: slot q of parity p holds block  p ? nblk-1-(w+4q) ... see below
    const int nb = (KK + 15) >> 4, nblk = nb * (nb + 1) / 2;
    // waves 0, 1 own three blocks on even tiles and two on odd tiles, waves 2, 3 the other way round: slot sets A (3) and B (2)
    const int parA = (w < 2) ? 0 : 1;                          // tile parity on which this wave uses set A
    int aI[3], aJ[3], bI[2], bJ[2]; bool aOn[3], bOn[2];
    auto block_of = [&](int idx, int& I, int& J) { I = 0; int rem = idx; while (rem >= nb - I) { rem -= nb - I; ++I; } J = I + rem; };
#pragma unroll
    for (int q = 0; q < 3; ++q) { const int wv = parA ? 3 - w : w; int idx = wv + 4 * q; aOn[q] = idx < nblk; block_of(aOn[q] ? idx : 0, aI[q], aJ[q]); }
#pragma unroll
    for (int q = 0; q < 2; ++q) { const int wv = parA ? w : 3 - w; int idx = wv + 4 * q; bOn[q] = idx < nblk; block_of(bOn[q] ? idx : 0, bI[q], bJ[q]); }
    // KK = 64 (four panels, ten blocks): sets that share a panel (gram_f64_shared) instead of the round-robin deal --
    //   even waves: A = {(0,0),(0,1),(0,2)} (rows from panel 0), B = {(1,1),(1,2)} (rows from panel 1)
    //   odd waves : A = {(0,3),(1,3),(2,3)} (columns from panel 3), B = {(2,2),(3,3)}
    constexpr bool shared_sets = SHARED;
    if (shared_sets) {
#pragma unroll
        for (int q = 0; q < 3; ++q) { aOn[q] = true; aI[q] = (w & 1) ? q : 0; aJ[q] = (w & 1) ? 3 : q; }
        bOn[0] = bOn[1] = true;
        bI[0] = (w & 1) ? 2 : 1; bJ[0] = (w & 1) ? 2 : 1; bI[1] = (w & 1) ? 3 : 1; bJ[1] = (w & 1) ? 3 : 2;
    }
    // M3: Gauss' three-multiplication product in f64 (mfma_common.hpp, CAcc32::mac_conj): per block  sum (ar+ai) br,  sum ai (br-bi),  sum ar (bi+br)
    v4d CAr[3], CAi[3], CBr[2], CBi[2], CAc[3], CBc[2];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { CAr[j][r] = 0.0; CAi[j][r] = 0.0; CAc[j][r] = 0.0; }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { CBr[j][r] = 0.0; CBi[j][r] = 0.0; CBc[j][r] = 0.0; }
    for (int e = tid; e < 4 * 64 * TRP; e += 256) Xbuf[e] = 0.f;
    const TileMap m = make_map(tid, D, TA, TB, PA, K);
    const long long kstride = (long long)D * PA;
    const bool fast = m.U <= 256 && (K + m.KP - 1) / m.KP <= NU;
    v4f px[NU];
    auto tile_origin = [&](int t, int& a0, int& b0, int& na, int& nb) {
        int ta = t % it.nta, tb = t / it.nta;
        a0 = ta * TA; b0 = tb * TB; na = min(TA, it.PA - a0); nb = min(TB, it.PB - b0);
    };
    // full tiles with NU two-element units per thread: straight-line loads (the guarded loop compiles into one branch per load with an
    // s_waitcnt vmcnt(0) before the next, which serialises the NU loads of a tile -- kernels_chi64.hip, mfma_gram64_kernel)
    const bool straight = fast && m.active && m.vec == 2 && K == m.KP * NU;
    auto issue_loads = [&](int t) {
        int a0, b0, na, nb; tile_origin(t, a0, b0, na, nb);
        const long long org = (long long)D * (a0 + PA * (long long)K * b0);
        if (straight && na == TA && nb == TB) {
            const cf* p0 = Xg + org + m.off + kstride * m.kp; const long long st = kstride * m.KP;
#pragma unroll
            for (int j = 0; j < NU; ++j) px[j] = ldg4(p0 + st * j);
            return;
        }
        const bool v0 = m.active && m.al < na && m.bl < nb, v1 = v0 && m.al1 < na;
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
        float* Xr = Xbuf + buf * (2 * 64 * TRP); float* Xi = Xr + 64 * TRP;
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
        float* Xr = Xbuf + buf * (2 * 64 * TRP); float* Xi = Xr + 64 * TRP;
        int a0, b0, na, nb; tile_origin(t, a0, b0, na, nb);
        const int ntile_el = D * TA * K * TB;
        for (int e = tid; e < ntile_el; e += 256) {
            int s = e % D; int r1 = e / D; int al = r1 % TA; int r2 = r1 / TA; int k = r2 % K; int bl = r2 / K;
            cf vx; vx.re = vx.im = 0.f;
            if (al < na && bl < nb) vx = Xg[s + D * ((long long)(a0 + al) + PA * ((long long)k + (long long)K * (b0 + bl)))];
            int o = (s + D * k) * TRP + (al + TA * bl);
            Xr[o] = vx.re; Xi[o] = vx.im;
        }
    };
    // software pipeline over the tiles of this chunk: loads run two tiles ahead (registers), the LDS commit one tile ahead
    lds_barrier();                                               // zero fill done
    if (t_begin < t_end) { if (fast) { issue_loads(t_begin); commit_loads(0); if (t_begin + 1 < t_end) issue_loads(t_begin + 1); } else fill_slow(t_begin, 0); }
    lds_barrier();
    for (int t = t_begin; t < t_end; ++t) {
        const int cur = (t - t_begin) & 1;
        if (t + 1 < t_end) {                                     // buffer cur^1 was last read at tile t-1: every wave passed the barrier since
            if (fast) { commit_loads(cur ^ 1); if (t + 2 < t_end) issue_loads(t + 2); } else fill_slow(t + 1, cur ^ 1);
        }
        const float* Xr = Xbuf + cur * (2 * 64 * TRP); const float* Xi = Xr + 64 * TRP;
        // rows of the tile: lane quarter kq takes rows 16*kq + tt, tt = 0..15, in two halves of 8 to bound registers
        auto block_pass = [&](int I, int J, v4d& cr, v4d& ci, v4d& cc) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int ro = (16 * I + l15) * TRP + 16 * kq + 8 * half;
                const int rb = (16 * J + l15) * TRP + 16 * kq + 8 * half;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const v4f t0 = *reinterpret_cast<const v4f*>(Xr + ro + 4 * q), t1 = *reinterpret_cast<const v4f*>(Xi + ro + 4 * q);
                    const v4f u0 = *reinterpret_cast<const v4f*>(Xr + rb + 4 * q), u1 = *reinterpret_cast<const v4f*>(Xi + rb + 4 * q);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const double ar = (double)t0[c], ai = (double)t1[c], br = (double)u0[c], bi = (double)u1[c];
                        // out[i][j] += x[i] * conj(x[j])
                        if (M3) {
                            cr = __builtin_amdgcn_mfma_f64_16x16x4f64(ar + ai, br, cr, 0, 0, 0);
                            ci = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, br - bi, ci, 0, 0, 0);
                            cc = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, bi + br, cc, 0, 0, 0);
                        } else {
                            cr = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, cr, 0, 0, 0);
                            ci = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, br, ci, 0, 0, 0);
                            cr = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, bi, cr, 0, 0, 0);
                            ci = __builtin_amdgcn_mfma_f64_16x16x4f64(-ar, bi, ci, 0, 0, 0);
                        }
                    }
                }
            }
        };
        if (dbg_skip != 1 && shared_sets) {
            if (((t - t_begin) & 1) == parA) {                    // wave-uniform
                const int q3[3] = {0, 1, 2};
                if (w & 1) gram_f64_shared<3, false, -1, M3>(Xr, Xi, TRP, l15, kq, 3, q3, CAr, CAi, CAc);
                else       gram_f64_shared<3, true, 0, M3>(Xr, Xi, TRP, l15, kq, 0, q3, CAr, CAi, CAc);
            } else if (w & 1) {
                const int q2[1] = {2}, q3b[1] = {3};
#pragma unroll
                for (int b = 0; b < 2; ++b) {                      // two diagonal blocks, each on its own panel (register copies, no address taken)
                    v4d r1[1] = {CBr[b]}, i1[1] = {CBi[b]}, c1[1] = {CBc[b]};
                    if (b == 0) gram_f64_shared<1, true, 0, M3>(Xr, Xi, TRP, l15, kq, 2, q2, r1, i1, c1);
                    else        gram_f64_shared<1, true, 0, M3>(Xr, Xi, TRP, l15, kq, 3, q3b, r1, i1, c1);
                    CBr[b] = r1[0]; CBi[b] = i1[0]; CBc[b] = c1[0];
                }
            } else {
                const int q12[2] = {1, 2};
                gram_f64_shared<2, true, 0, M3>(Xr, Xi, TRP, l15, kq, 1, q12, CBr, CBi, CBc);
            }
        } else if (dbg_skip != 1) {
            if (((t - t_begin) & 1) == parA) {                    // wave-uniform
#pragma unroll
                for (int q = 0; q < 3; ++q) if (aOn[q]) block_pass(aI[q], aJ[q], CAr[q], CAi[q], CAc[q]);
            } else {
#pragma unroll
                for (int q = 0; q < 2; ++q) if (bOn[q]) block_pass(bI[q], bJ[q], CBr[q], CBi[q], CBc[q]);
            }
        }
        lds_barrier();                                           // tile t consumed by everybody, tile t+1 committed by everybody
    }
    struct alignas(16) cd { double re, im; };
    auto write_block = [&](cd* __restrict__ part, int I, int J, const v4d& cr, const v4d& ci, const v4d& cc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int i = 16 * I + kq + 4 * r, j = 16 * J + l15;
            if (i < KK && j < KK) {
                cd v; v.re = M3 ? cr[r] - ci[r] : cr[r]; v.im = M3 ? cr[r] - cc[r] : ci[r]; part[i + (size_t)KK * j] = v;
                if (I != J) { cd c; c.re = v.re; c.im = -v.im; part[j + (size_t)KK * i] = c; }     // G[j][i] = conj(G[i][j])
            }
        }
    };
    // partial 2*lc + p collects the blocks accumulated on tiles of parity p (each block exactly once per parity)
    cd* __restrict__ partA = reinterpret_cast<cd*>(it.partial) + (size_t)(2 * lc + parA) * KK * KK;
    cd* __restrict__ partB = reinterpret_cast<cd*>(it.partial) + (size_t)(2 * lc + (parA ^ 1)) * KK * KK;
#pragma unroll
    for (int q = 0; q < 3; ++q) if (aOn[q]) write_block(partA, aI[q], aJ[q], CAr[q], CAi[q], CAc[q]);
#pragma unroll
    for (int q = 0; q < 2; ++q) if (bOn[q]) write_block(partB, bI[q], bJ[q], CBr[q], CBi[q], CBc[q]);
}
bool launch_mfma_gram64_f64(hipStream_t s, const GramItem* d_items, int nitems, int total_chunks, int KKmax, bool all_kk64) {
    if (KKmax > 64) return false;
    if (total_chunks <= 0) return true;
    const size_t lds = (size_t)4 * 64 * 68 * sizeof(float);
    static int skip = -1;
    if (skip < 0) {
        skip = 0;
#ifdef TNQS_EXPERIMENTS
        if (const char* e = std::getenv("TNQS_DBG_GRAM_SKIP")) skip = std::atoi(e);
#endif
    }
#define TNQS_G64(M3, SH) { set_max_dynamic_lds((const void*)mfma_gram64_f64_kernel<M3, SH>, lds); hipLaunchKernelGGL((mfma_gram64_f64_kernel<M3, SH>), dim3(total_chunks), dim3(256), lds, s, d_items, nitems, skip); }
    if (all_kk64) TNQS_G64(true, true) else TNQS_G64(true, false)
#undef TNQS_G64
    TNQS_CHECK_LAUNCH();
    return true;
}

bool launch_mfma_gram32(hipStream_t s, const GramItem* d_items, int nitems, int total_chunks, int KKmax) {
    if (KKmax > 32) return false;
    if (total_chunks <= 0) return true;
    hipLaunchKernelGGL(mfma_gram32_kernel, dim3(total_chunks), dim3(256), 0, s, d_items, nitems); TNQS_CHECK_LAUNCH();
    return true;
}

}  // namespace tnqs
