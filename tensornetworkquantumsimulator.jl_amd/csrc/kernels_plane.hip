// kernels_plane.hip -- CDNA4 (gfx950) plane kernels for bond dimension 16 (ComplexF32): the per-site shape of BASELINE configs[3]
// (periodic cubic lattice, degree 6, chi = 16: site tensors of 2 x 16^6 elements = 268 MB) and of the heavy-hex lattice.
//
// At chi = 16 one mode product has an arithmetic intensity of 8 flop/B (ridge of the machine: 19.7), so everything here is about HBM
// passes: two legs per pass (mfma_pair16_kernel / mfma_pair16w_kernel, 16 flop/B) and both messages of a forest from one pass over
// (T, psi) (mfma_pair_gram2x16_kernel, 32 flop/B).  The kernels are WAVE PRIVATE: a wave owns an LDS slab, moves the companion planes of
// its slice itself (register prefetch of the next one) and never meets a workgroup barrier inside its loop, so the waves of a CU drift
// apart and one wave's global / LDS phases hide behind the others' MFMAs.  What a wave owns differs per kernel, and measurably matters
// (profiles/plane16_bench.py):
//   mfma_pair16_kernel        8 waves x half slices (8 companions, the other half of each line belongs to the neighbouring wave): planes that
//                             contain leg 0, where a wave's 16 lanes read 256 contiguous bytes anyway (5.0 - 5.3 TB/s)
//   mfma_pair16w_kernel       4 waves x whole slices: all other planes, whose 16 companions are ONE 128-byte line (4.0 - 5.1 TB/s; the
//                             half-slice kernel ran them at 2.5 - 3.9)
//   mfma_pair_gram2x16_kernel 8 waves x half slices, four companions resident at a time: two waves per SIMD keep the matrix pipe fed
// Matrix instruction: v_mfma_f32_16x16x4_f32 (one 16 x 16 plane = one tile).  Lane l = (c = l & 15, g = l >> 4):
//      A[i = c][k = g]   B[k = g][j = c]   C[row = 4 g + r][col = c],  r = 0..3
// k only has to be consistent between A and B: k-step t of group g is mapped to index 4 g + t, so a lane reads FOUR CONSECUTIVE complex
// numbers of its operand row from LDS (2 x ds_read_b128).  Chaining as in the chi = 32 kernels: accumulator register r of a product
// holds row 4 g + r, which is exactly the k index instruction r of the next product consumes -- the intermediate never leaves registers.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <type_traits>
#define TNQS_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) throw std::runtime_error(std::string("HIP kernel launch failed (") + __func__ + "): " + hipGetErrorString(e_)); } while (0)
#include "kernels.hpp"
#include "launch_util.hpp"
#include "mfma_common.hpp"

namespace tnqs {

__device__ __forceinline__ long long plane_slice_base(const PlaneGeom& g, int sl) {
    int a0 = sl % g.n0; int r1 = sl / g.n0; int a1 = r1 % g.n1; int r2 = r1 / g.n1; int a2 = r2 % g.n2; int a3 = r2 / g.n2;
    return (long long)a0 * g.t0 + (long long)a1 * g.t1 + (long long)a2 * g.t2 + (long long)a3 * g.t3;
}

// LDS plane of one companion: element (ix, iy) at [iy][ix], pitch 18 complex (16-byte aligned rows for ds_read_b128, and the 16 rows
// a b128 operand read touches fall on 16 different 4-bank groups); plane stride 292 (the four companion pairs of a commit start 16 banks apart)
constexpr int P16 = 18, PS16 = 16 * P16 + 4;

// ------------------------------------------------------------------------------------------------------------
// pair of mode products on two 16-dimensional legs x, y in ONE pass:
//      out[c, jx, jy] = sum_{ix,iy} in[c, ix, iy] Mx[ix, jx] My[iy, jy]          for every companion index c
//   step 1  Y[iy][jx] = sum_ix S[ix][iy] Mx[ix][jx]       (A = S^T from LDS, B = Mx in registers)
//   step 2  S'[jx][jy] = sum_iy Y[iy][jx] My[iy][jy]      (A = Y's accumulator registers as they are, B = My in registers)
// Wave w of a workgroup takes half (w & 1) of slices s0 + (w >> 1), s0 + (w >> 1) + 4, ...; the next unit's 16 KiB are prefetched into
// registers before the matrix work of the current one.
// ------------------------------------------------------------------------------------------------------------
template <bool M3>          // M3: Gauss' three-multiplication complex product (mfma_common.hpp), 24 instead of 32 matrix instructions per plane
__global__ __launch_bounds__(512) void mfma_pair16_kernel(const Pair16Item* __restrict__ items, int nitems) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c16 = lane & 15, g4 = lane >> 4;
    v2f* const L = reinterpret_cast<v2f*>(smem) + w * (8 * PS16);
    int lo = 0, hi_ = nitems - 1;
    const int gw = blockIdx.x;
    while (lo < hi_) { int mid = (lo + hi_ + 1) >> 1; if (items[mid].wg_begin <= gw) lo = mid; else hi_ = mid - 1; }
    const Pair16Item it = items[lo];
    const PlaneGeom g = it.g;
    const int nslices = g.n0 * g.n1 * g.n2 * g.n3;
    const int s_begin = (gw - it.wg_begin) * it.spw, s_end = min(nslices, s_begin + it.spw);
    const cf* __restrict__ in = reinterpret_cast<const cf*>(it.in);
    cf* __restrict__ out = reinterpret_cast<cf*>(it.out);
    const cf* __restrict__ Mx = reinterpret_cast<const cf*>(it.Mx);
    const cf* __restrict__ My = reinterpret_cast<const cf*>(it.My);
    // B operands: Mx[k = 4 g + t][j = c] (step 1), My[k = 4 g + r][j = c] (step 2); element (i, j) at i + 16 j
    // M3: (mxi, myi) hold bi - br and (mxs, mys) br + bi
    float mxr[4], mxi[4], myr[4], myi[4], mxs[4], mys[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        cf a = Mx[(4 * g4 + t) + 16 * c16]; mxr[t] = a.re; mxi[t] = M3 ? a.im - a.re : a.im; mxs[t] = a.re + a.im;
        cf b = My[(4 * g4 + t) + 16 * c16]; myr[t] = b.re; myi[t] = M3 ? b.im - b.re : b.im; mys[t] = b.re + b.im;
    }
    // mover: lane -> (companion pair f = lane & 3: 16 bytes, ix = lane >> 2); load j covers iy = j
    const int f = lane & 3, ix0 = lane >> 2, half = w & 1;
    const long long toff = (long long)(4 * half + f) * g.cstr + g.sx * ix0;
    v2f* const lbase = L + (2 * f) * PS16 + ix0;
    v4f pre[16];
    auto issue = [&](int sl) {
        const cf* p = in + plane_slice_base(g, sl) + toff;
#pragma unroll
        for (int j = 0; j < 16; ++j) pre[j] = ldg4(p + g.sy * j);
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            v2f a; a[0] = pre[j][0]; a[1] = pre[j][1]; v2f b; b[0] = pre[j][2]; b[1] = pre[j][3];
            lbase[P16 * j] = a; lbase[P16 * j + PS16] = b;
        }
    };
    int sl = s_begin + (w >> 1);
    if (sl < s_end) issue(sl);
    for (; sl < s_end; sl += 4) {
        commit();
        __builtin_amdgcn_wave_barrier();                 // LDS is in order per wave; only the compiler must not reorder
        if (sl + 4 < s_end) issue(sl + 4);
#pragma unroll 2
        for (int c = 0; c < 8; ++c) {
            v2f* const P = L + c * PS16;
            const v4f a01 = *reinterpret_cast<const v4f*>(P + c16 * P16 + 4 * g4);          // A[i = iy = c16][k = ix = 4 g + t]
            const v4f a23 = *reinterpret_cast<const v4f*>(P + c16 * P16 + 4 * g4 + 2);
            const float ar[4] = {a01[0], a01[2], a23[0], a23[2]}, ai[4] = {a01[1], a01[3], a23[1], a23[3]};
            v4f Yr = {0.f, 0.f, 0.f, 0.f}, Yi = {0.f, 0.f, 0.f, 0.f};
            v4f Sr = {0.f, 0.f, 0.f, 0.f}, Si = {0.f, 0.f, 0.f, 0.f};
            if (M3) {
                v4f k1 = Yr, k2 = Yr, k3 = Yr;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    k1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[t] + ai[t], mxr[t], k1, 0, 0, 0);
                    k2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[t], mxi[t], k2, 0, 0, 0);
                    k3 = __builtin_amdgcn_mfma_f32_16x16x4f32(ai[t], mxs[t], k3, 0, 0, 0);
                }
                Yr = k1 - k3; Yi = k1 + k2;
                v4f q1 = Sr, q2 = Sr, q3 = Sr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {                                                 // Y reg r: row iy = 4 g + r, col jx = c16
                    q1 = __builtin_amdgcn_mfma_f32_16x16x4f32(Yr[r] + Yi[r], myr[r], q1, 0, 0, 0);
                    q2 = __builtin_amdgcn_mfma_f32_16x16x4f32(Yr[r], myi[r], q2, 0, 0, 0);
                    q3 = __builtin_amdgcn_mfma_f32_16x16x4f32(Yi[r], mys[r], q3, 0, 0, 0);
                }
                Sr = q1 - q3; Si = q1 + q2;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    Yr = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[t], mxr[t], Yr, 0, 0, 0);
                    Yi = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[t], mxi[t], Yi, 0, 0, 0);
                    Yr = __builtin_amdgcn_mfma_f32_16x16x4f32(-ai[t], mxi[t], Yr, 0, 0, 0);
                    Yi = __builtin_amdgcn_mfma_f32_16x16x4f32(ai[t], mxr[t], Yi, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {                                                 // Y reg r: row iy = 4 g + r, col jx = c16
                    Sr = __builtin_amdgcn_mfma_f32_16x16x4f32(Yr[r], myr[r], Sr, 0, 0, 0);
                    Si = __builtin_amdgcn_mfma_f32_16x16x4f32(Yr[r], myi[r], Si, 0, 0, 0);
                    Sr = __builtin_amdgcn_mfma_f32_16x16x4f32(-Yi[r], myi[r], Sr, 0, 0, 0);
                    Si = __builtin_amdgcn_mfma_f32_16x16x4f32(Yi[r], myr[r], Si, 0, 0, 0);
                }
            }
            // S'[jx = 4 g + r][jy = c16] -> LDS [jy][jx]: four consecutive complex numbers
            v4f o01 = {Sr[0], Si[0], Sr[1], Si[1]}, o23 = {Sr[2], Si[2], Sr[3], Si[3]};
            *reinterpret_cast<v4f*>(P + c16 * P16 + 4 * g4) = o01;
            *reinterpret_cast<v4f*>(P + c16 * P16 + 4 * g4 + 2) = o23;
        }
        __builtin_amdgcn_wave_barrier();
        {
            cf* p = out + plane_slice_base(g, sl) + toff;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const v2f a = lbase[P16 * j], b = lbase[P16 * j + PS16];
                v4f v = {a[0], a[1], b[0], b[1]};
                stg4(p + g.sy * j, v);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}
// ------------------------------------------------------------------------------------------------------------
// The same pass with WHOLE 128-byte lines per wave.  Above, the two waves of a pair read the two 64-byte halves of every line; measured
// per leg pair on 12 degree-6 sites (profiles/plane16_bench.py): 3.2 - 3.9 TB/s when the 16 companions of a line are 16 consecutive
// elements (planes that do not contain leg 0), 5.0 - 5.3 TB/s when the plane contains leg 0 and a wave's 16 lanes read 256 contiguous
// bytes.  Here a wave owns all 16 companions of a slice: 32 KiB in registers (prefetch of the next slice), 16 planes in its LDS slab
// (36.5 KiB; four waves per workgroup, one per SIMD), the same matrix work per plane.  The kernel is far from matrix-core bound (27 ms
// against 50 ms of memory time on the 3 x 3 x 3 torus), so one wave per SIMD is enough to cover the compute phase with the prefetch.
// ------------------------------------------------------------------------------------------------------------
template <bool M3>
__global__ __launch_bounds__(256) void mfma_pair16w_kernel(const Pair16Item* __restrict__ items, int nitems) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c16 = lane & 15, g4 = lane >> 4;
    v2f* const L = reinterpret_cast<v2f*>(smem) + w * (16 * PS16);
    int lo = 0, hi_ = nitems - 1;
    const int gw = blockIdx.x;
    while (lo < hi_) { int mid = (lo + hi_ + 1) >> 1; if (items[mid].wg_begin <= gw) lo = mid; else hi_ = mid - 1; }
    const Pair16Item it = items[lo];
    const PlaneGeom g = it.g;
    const int nslices = g.n0 * g.n1 * g.n2 * g.n3;
    const int s_begin = (gw - it.wg_begin) * it.spw, s_end = min(nslices, s_begin + it.spw);
    const cf* __restrict__ in = reinterpret_cast<const cf*>(it.in);
    cf* __restrict__ out = reinterpret_cast<cf*>(it.out);
    const cf* __restrict__ Mx = reinterpret_cast<const cf*>(it.Mx);
    const cf* __restrict__ My = reinterpret_cast<const cf*>(it.My);
    float mxr[4], mxi[4], myr[4], myi[4], mxs[4], mys[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        cf a = Mx[(4 * g4 + t) + 16 * c16]; mxr[t] = a.re; mxi[t] = M3 ? a.im - a.re : a.im; mxs[t] = a.re + a.im;
        cf b = My[(4 * g4 + t) + 16 * c16]; myr[t] = b.re; myi[t] = M3 ? b.im - b.re : b.im; mys[t] = b.re + b.im;
    }
    // mover: lane -> (companion pair f = lane & 7: 16 bytes, 8 lanes = one line; ix = (lane >> 3) + 8 h); load (h, j) covers iy = j
    const int f = lane & 7, ix0 = lane >> 3;
    const long long toff = (long long)f * g.cstr + g.sx * ix0;
    v2f* const lbase = L + (2 * f) * PS16 + ix0;
    v4f pre[32];
    auto issue = [&](int sl) {
        const cf* p = in + plane_slice_base(g, sl) + toff;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 16; ++j) pre[16 * h + j] = ldg4(p + 8 * g.sx * h + g.sy * j);
    };
    auto commit = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const v4f v = pre[16 * h + j];
                v2f a; a[0] = v[0]; a[1] = v[1]; v2f b; b[0] = v[2]; b[1] = v[3];
                lbase[P16 * j + 8 * h] = a; lbase[P16 * j + 8 * h + PS16] = b;
            }
    };
    int sl = s_begin + w;
    if (sl < s_end) issue(sl);
    for (; sl < s_end; sl += 4) {
        commit();
        __builtin_amdgcn_wave_barrier();                 // LDS is in order per wave; only the compiler must not reorder
        if (sl + 4 < s_end) issue(sl + 4);
#pragma unroll 2
        for (int c = 0; c < 16; ++c) {
            v2f* const P = L + c * PS16;
            const v4f a01 = *reinterpret_cast<const v4f*>(P + c16 * P16 + 4 * g4);          // A[i = iy = c16][k = ix = 4 g + t]
            const v4f a23 = *reinterpret_cast<const v4f*>(P + c16 * P16 + 4 * g4 + 2);
            const float ar[4] = {a01[0], a01[2], a23[0], a23[2]}, ai[4] = {a01[1], a01[3], a23[1], a23[3]};
            v4f Yr = {0.f, 0.f, 0.f, 0.f}, Yi = Yr, Sr = Yr, Si = Yr;
            if (M3) {
                v4f k1 = Yr, k2 = Yr, k3 = Yr;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    k1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[t] + ai[t], mxr[t], k1, 0, 0, 0);
                    k2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[t], mxi[t], k2, 0, 0, 0);
                    k3 = __builtin_amdgcn_mfma_f32_16x16x4f32(ai[t], mxs[t], k3, 0, 0, 0);
                }
                Yr = k1 - k3; Yi = k1 + k2;
                v4f q1 = Sr, q2 = Sr, q3 = Sr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    q1 = __builtin_amdgcn_mfma_f32_16x16x4f32(Yr[r] + Yi[r], myr[r], q1, 0, 0, 0);
                    q2 = __builtin_amdgcn_mfma_f32_16x16x4f32(Yr[r], myi[r], q2, 0, 0, 0);
                    q3 = __builtin_amdgcn_mfma_f32_16x16x4f32(Yi[r], mys[r], q3, 0, 0, 0);
                }
                Sr = q1 - q3; Si = q1 + q2;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    Yr = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[t], mxr[t], Yr, 0, 0, 0);
                    Yi = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[t], mxi[t], Yi, 0, 0, 0);
                    Yr = __builtin_amdgcn_mfma_f32_16x16x4f32(-ai[t], mxi[t], Yr, 0, 0, 0);
                    Yi = __builtin_amdgcn_mfma_f32_16x16x4f32(ai[t], mxr[t], Yi, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    Sr = __builtin_amdgcn_mfma_f32_16x16x4f32(Yr[r], myr[r], Sr, 0, 0, 0);
                    Si = __builtin_amdgcn_mfma_f32_16x16x4f32(Yr[r], myi[r], Si, 0, 0, 0);
                    Sr = __builtin_amdgcn_mfma_f32_16x16x4f32(-Yi[r], myi[r], Sr, 0, 0, 0);
                    Si = __builtin_amdgcn_mfma_f32_16x16x4f32(Yi[r], myr[r], Si, 0, 0, 0);
                }
            }
            v4f o01 = {Sr[0], Si[0], Sr[1], Si[1]}, o23 = {Sr[2], Si[2], Sr[3], Si[3]};      // S'[jx = 4 g + r][jy = c16] -> LDS [jy][jx]
            *reinterpret_cast<v4f*>(P + c16 * P16 + 4 * g4) = o01;
            *reinterpret_cast<v4f*>(P + c16 * P16 + 4 * g4 + 2) = o23;
        }
        __builtin_amdgcn_wave_barrier();
        {
            cf* p = out + plane_slice_base(g, sl) + toff;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const v2f a = lbase[P16 * j + 8 * h], b = lbase[P16 * j + 8 * h + PS16];
                    v4f v = {a[0], a[1], b[0], b[1]};
                    stg4(p + 8 * g.sx * h + g.sy * j, v);
                }
        }
        __builtin_amdgcn_wave_barrier();
    }
}
// whole_lines: every item has its 16 companions in one 128-byte line (PlaneGeom::cstr == 2) and TNQS_PAIR16_HALF is not set: see pair16_whole_lines()
bool pair16_whole_lines(const PlaneGeom& g) {
#ifdef TNQS_EXPERIMENTS
    static const bool off = [] { const char* e = std::getenv("TNQS_PAIR16_HALF"); return e && e[0] == '1'; }();
    if (off) return false;
#endif
    return g.cstr == 2;
}
void launch_mfma_pair16(hipStream_t s, const Pair16Item* d_items, int nitems, int total_wgs, bool whole_lines) {
    if (total_wgs > 0 && whole_lines) {
        const size_t lds = (size_t)4 * 16 * PS16 * sizeof(v2f);
        set_max_dynamic_lds((const void*)mfma_pair16w_kernel<true>, lds); hipLaunchKernelGGL(mfma_pair16w_kernel<true>, dim3(total_wgs), dim3(256), lds, s, d_items, nitems);
        TNQS_CHECK_LAUNCH();
        return;
    }
    if (total_wgs <= 0) return;
    const size_t lds = (size_t)8 * 8 * PS16 * sizeof(v2f);
    set_max_dynamic_lds((const void*)mfma_pair16_kernel<true>, lds); hipLaunchKernelGGL(mfma_pair16_kernel<true>, dim3(total_wgs), dim3(512), lds, s, d_items, nitems);
    TNQS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// BOTH messages a site sends into a linear forest, from one pass over the shared partial product X and psi = Y (16 x 16 planes):
//      out_y[b,b'] = sum_{c,jx} ( sum_ix X[c,ix,b] Mx[ix,jx] ) conj Y[c,jx,b']      (kept leg y, leg x absorbed)
//      out_x[d,d'] = sum_{c,jy} ( sum_iy X[c,d,iy] My[iy,jy] ) conj Y[c,d',jy]      (kept leg x, leg y absorbed)
// 32 flop/B: co-bound.  Both 16 x 16 messages are accumulated in registers over the workgroup's slices.
// ------------------------------------------------------------------------------------------------------------
// Eight waves per workgroup, two per SIMD.  A wave moves half slices (8 companions, 64-byte runs) through its registers but keeps only FOUR
// companions' planes resident (18 KiB): commit companions 0..3 (the lanes that hold them) -> matrix work -> commit 4..7 -> prefetch the next
// half slice -> matrix work.  The matrix phase of one wave runs under the memory phases of the other wave of its SIMD: 1.61 - 1.87 ms per
// 12 sites against 1.97 - 2.15 with four waves of 8 resident companions (one per SIMD), which was matrix-core bound in practice: 1.78 ms
// even without its global loads.
template <bool M3>          // M3: three-multiplication complex products, 48 instead of 64 matrix instructions per companion (both messages)
__global__ __launch_bounds__(512) void mfma_pair_gram2x16_kernel(const PairGram2x16Item* __restrict__ items, int nitems) {
    constexpr int NW = 8, NC = 4;                                        // waves per workgroup, companions resident per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c16 = lane & 15, g4 = lane >> 4;
    v2f* const L = reinterpret_cast<v2f*>(smem) + w * (2 * NC * PS16);   // planes 0..NC-1: X of the resident companions, planes NC..2NC-1: Y
    int lo = 0, hi_ = nitems - 1;
    const int gw = blockIdx.x;
    while (lo < hi_) { int mid = (lo + hi_ + 1) >> 1; if (items[mid].wg_begin <= gw) lo = mid; else hi_ = mid - 1; }
    const PairGram2x16Item it = items[lo];
    const PlaneGeom g = it.g;
    const int nslices = g.n0 * g.n1 * g.n2 * g.n3;
    const int lw = gw - it.wg_begin;
    const int s_begin = lw * it.spw, s_end = min(nslices, s_begin + it.spw);
    const cf* __restrict__ Xg = reinterpret_cast<const cf*>(it.X);
    const cf* __restrict__ Yg = reinterpret_cast<const cf*>(it.Y);
    // A operands of the first steps: Mx^T / My^T, lane (i = c16, g): k-step t -> M[4 g + t][c16]
    float mxr[4], mxi[4], myr[4], myi[4], mxs[4], mys[4];      // M3: (mxs, mys) = re + im, the A-side sum of Gauss' product
    {
        const cf* __restrict__ Mx = reinterpret_cast<const cf*>(it.Mx); const cf* __restrict__ My = reinterpret_cast<const cf*>(it.My);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            cf a = Mx[(4 * g4 + t) + 16 * c16]; mxr[t] = a.re; mxi[t] = a.im; mxs[t] = a.re + a.im;
            cf b = {0.f, 0.f}; if (My) b = My[(4 * g4 + t) + 16 * c16]; myr[t] = b.re; myi[t] = b.im; mys[t] = b.re + b.im;
        }
    }
    const bool both = it.My != nullptr;                 // My == null: only the message through ly is wanted (a site that sends one message in this level)
    // M3: (O?r, O?i, O?c) accumulate sum (ar + ai) br, sum ai (br - bi), sum ar (bi + br) of out += C conj(Y): re = r - i, im = r - c
    v4f O1r = {0.f, 0.f, 0.f, 0.f}, O1i = O1r, O2r = O1r, O2i = O1r, O1c = O1r, O2c = O1r;
    const int f = lane & 3, ix0 = lane >> 2, half = w & 1;
    const long long toff = (long long)(4 * half + f) * g.cstr + g.sx * ix0;
    v2f* const lbase = L + (2 * (f & 1)) * PS16 + ix0;
    v4f px[16], py[16];
    auto issue = [&](int sl) {
        const long long b = plane_slice_base(g, sl) + toff;
#pragma unroll
        for (int j = 0; j < 16; ++j) { px[j] = ldg4(Xg + b + g.sy * j); py[j] = ldg4(Yg + b + g.sy * j); }
    };
    auto commit = [&](int sub) {                        // the lanes whose companion pair belongs to the group `sub` of four
        if ((f >> 1) != sub) return;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            v2f a = {px[j][0], px[j][1]}, b = {px[j][2], px[j][3]}, c = {py[j][0], py[j][1]}, d = {py[j][2], py[j][3]};
            lbase[P16 * j] = a; lbase[P16 * j + PS16] = b; lbase[P16 * j + NC * PS16] = c; lbase[P16 * j + (NC + 1) * PS16] = d;
        }
    };
    // waves (2 m, 2 m + 1) take the two halves of slices s0 + m, s0 + m + 4, ...  (m = 0..3)
    // (the loop is instantiated for both / one message: a wave-uniform branch inside it would split the companion loop into basic blocks and
    // keep the scheduler from overlapping one companion's LDS reads with the previous one's matrix instructions)
    auto run = [&](auto both_c) {
    constexpr bool BOTH = decltype(both_c)::value;
    int sl = s_begin + (w >> 1);
    if (sl < s_end) issue(sl);
    for (; sl < s_end; sl += NW / 2) {
        commit(0);
        __builtin_amdgcn_wave_barrier();
        // the companion loop, software pipelined: the operands of companion c + 1 are read from LDS before the matrix work of companion c, and the
        // two messages' products advance together (first products of both, then the second ones), so that the LDS latency and the result
        // latency of the first products are covered by the other chain's instructions -- a wave has its SIMD to itself here
        struct Ops { v4f x01, x23, y01, y23; v2f xt[4], yt[4]; };
        auto load_ops = [&](int c, Ops& o) {
            const v2f* const PX = L + c * PS16; const v2f* const PY = L + (NC + c) * PS16;
            o.x01 = *reinterpret_cast<const v4f*>(PX + c16 * P16 + 4 * g4);                  // B[k = ix = 4 g + t][j = b = iy = c16]
            o.x23 = *reinterpret_cast<const v4f*>(PX + c16 * P16 + 4 * g4 + 2);
            o.y01 = *reinterpret_cast<const v4f*>(PY + c16 * P16 + 4 * g4);                  // B[k = jx = 4 g + r][j = b' = iy = c16]
            o.y23 = *reinterpret_cast<const v4f*>(PY + c16 * P16 + 4 * g4 + 2);
            if constexpr (BOTH) {
#pragma unroll
                for (int t = 0; t < 4; ++t) { o.xt[t] = PX[(4 * g4 + t) * P16 + c16]; o.yt[t] = PY[(4 * g4 + t) * P16 + c16]; }   // the same planes, read transposed
            }
        };
        auto compute = [&](const Ops& o) {
            const float xr[4] = {o.x01[0], o.x01[2], o.x23[0], o.x23[2]}, xi[4] = {o.x01[1], o.x01[3], o.x23[1], o.x23[3]};
            const float yr[4] = {o.y01[0], o.y01[2], o.y23[0], o.y23[2]}, yi[4] = {o.y01[1], o.y01[3], o.y23[1], o.y23[3]};
            const v4f z4 = {0.f, 0.f, 0.f, 0.f};
            // ---- first products: C1[jx = 4 g + r][b = c16] = Mx^T X (message through ly), C2[jy = 4 g + r][d = c16] = My^T X^T (through lx)
            v4f C1r = z4, C1i = z4, C2r = z4, C2i = z4;
            if (M3) {
                v4f k1 = z4, k2 = z4, k3 = z4, q1 = z4, q2 = z4, q3 = z4;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    k1 = __builtin_amdgcn_mfma_f32_16x16x4f32(mxs[t], xr[t], k1, 0, 0, 0);
                    k2 = __builtin_amdgcn_mfma_f32_16x16x4f32(mxr[t], xi[t] - xr[t], k2, 0, 0, 0);
                    k3 = __builtin_amdgcn_mfma_f32_16x16x4f32(mxi[t], xr[t] + xi[t], k3, 0, 0, 0);
                    if constexpr (BOTH) {
                        q1 = __builtin_amdgcn_mfma_f32_16x16x4f32(mys[t], o.xt[t][0], q1, 0, 0, 0);
                        q2 = __builtin_amdgcn_mfma_f32_16x16x4f32(myr[t], o.xt[t][1] - o.xt[t][0], q2, 0, 0, 0);
                        q3 = __builtin_amdgcn_mfma_f32_16x16x4f32(myi[t], o.xt[t][0] + o.xt[t][1], q3, 0, 0, 0);
                    }
                }
                C1r = k1 - k3; C1i = k1 + k2; C2r = q1 - q3; C2i = q1 + q2;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    C1r = __builtin_amdgcn_mfma_f32_16x16x4f32(mxr[t], xr[t], C1r, 0, 0, 0);
                    C1i = __builtin_amdgcn_mfma_f32_16x16x4f32(mxr[t], xi[t], C1i, 0, 0, 0);
                    C1r = __builtin_amdgcn_mfma_f32_16x16x4f32(-mxi[t], xi[t], C1r, 0, 0, 0);
                    C1i = __builtin_amdgcn_mfma_f32_16x16x4f32(mxi[t], xr[t], C1i, 0, 0, 0);
                    if constexpr (BOTH) {
                        C2r = __builtin_amdgcn_mfma_f32_16x16x4f32(myr[t], o.xt[t][0], C2r, 0, 0, 0);
                        C2i = __builtin_amdgcn_mfma_f32_16x16x4f32(myr[t], o.xt[t][1], C2i, 0, 0, 0);
                        C2r = __builtin_amdgcn_mfma_f32_16x16x4f32(-myi[t], o.xt[t][1], C2r, 0, 0, 0);
                        C2i = __builtin_amdgcn_mfma_f32_16x16x4f32(myi[t], o.xt[t][0], C2i, 0, 0, 0);
                    }
                }
            }
            // ---- second products: out1[b][b'] += C1[jx][b] conj Y[jx][b'],  out2[d][d'] += C2[jy][d] conj Y^T[jy][d'] ------------------
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (M3) {
                    O1r = __builtin_amdgcn_mfma_f32_16x16x4f32(C1r[r] + C1i[r], yr[r], O1r, 0, 0, 0);
                    O1c = __builtin_amdgcn_mfma_f32_16x16x4f32(C1r[r], yi[r] + yr[r], O1c, 0, 0, 0);
                    O1i = __builtin_amdgcn_mfma_f32_16x16x4f32(C1i[r], yr[r] - yi[r], O1i, 0, 0, 0);
                    if constexpr (BOTH) {
                        O2r = __builtin_amdgcn_mfma_f32_16x16x4f32(C2r[r] + C2i[r], o.yt[r][0], O2r, 0, 0, 0);
                        O2c = __builtin_amdgcn_mfma_f32_16x16x4f32(C2r[r], o.yt[r][1] + o.yt[r][0], O2c, 0, 0, 0);
                        O2i = __builtin_amdgcn_mfma_f32_16x16x4f32(C2i[r], o.yt[r][0] - o.yt[r][1], O2i, 0, 0, 0);
                    }
                } else {
                    O1r = __builtin_amdgcn_mfma_f32_16x16x4f32(C1r[r], yr[r], O1r, 0, 0, 0);
                    O1i = __builtin_amdgcn_mfma_f32_16x16x4f32(C1i[r], yr[r], O1i, 0, 0, 0);
                    O1r = __builtin_amdgcn_mfma_f32_16x16x4f32(C1i[r], yi[r], O1r, 0, 0, 0);
                    O1i = __builtin_amdgcn_mfma_f32_16x16x4f32(-C1r[r], yi[r], O1i, 0, 0, 0);
                    if constexpr (BOTH) {
                        O2r = __builtin_amdgcn_mfma_f32_16x16x4f32(C2r[r], o.yt[r][0], O2r, 0, 0, 0);
                        O2i = __builtin_amdgcn_mfma_f32_16x16x4f32(C2i[r], o.yt[r][0], O2i, 0, 0, 0);
                        O2r = __builtin_amdgcn_mfma_f32_16x16x4f32(C2i[r], o.yt[r][1], O2r, 0, 0, 0);
                        O2i = __builtin_amdgcn_mfma_f32_16x16x4f32(-C2r[r], o.yt[r][1], O2i, 0, 0, 0);
                    }
                }
            }
        };
        auto resident = [&]() {
            Ops oa, ob;
            load_ops(0, oa);
#pragma unroll
            for (int c = 0; c < NC; c += 2) {
                load_ops(c + 1, ob);
                compute(oa);
                if (c + 2 < NC) load_ops(c + 2, oa);
                compute(ob);
            }
        };
        resident();
        __builtin_amdgcn_wave_barrier();                 // companions 0..3 consumed
        commit(1);
        __builtin_amdgcn_wave_barrier();
        if (sl + 4 < s_end) issue(sl + 4);               // every lane's registers are free now
        resident();
        __builtin_amdgcn_wave_barrier();                 // the planes have been consumed: the next commit may overwrite them
    }
    };
    if (both) run(std::true_type{}); else run(std::false_type{});
    // one partial per workgroup and message: the four waves' accumulators are summed through LDS, in wave order
    cf* __restrict__ p1 = reinterpret_cast<cf*>(it.partial_y) + (size_t)lw * 256;
    cf* __restrict__ p2 = reinterpret_cast<cf*>(it.partial_x) + (size_t)lw * 256;
    if (M3) { const v4f a1 = O1r, a2 = O2r; O1r = a1 - O1i; O1i = a1 - O1c; O2r = a2 - O2i; O2i = a2 - O2c; }
    __syncthreads();                                     // every wave is done with its slab
    v2f* const R = reinterpret_cast<v2f*>(smem);        // [message 2][wave NW][16 x 17]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g4 + r;                        // element (i, j = c16)
        v2f a = {O1r[r], O1i[r]}, b = {O2r[r], O2i[r]};
        R[(0 * NW + w) * 272 + c16 * 17 + i] = a; R[(1 * NW + w) * 272 + c16 * 17 + i] = b;
    }
    __syncthreads();
    for (int e = tid; e < 512; e += 64 * NW) {
        const int msg = e >> 8, q = e & 255, i = q & 15, j = q >> 4;
        float sr = 0.f, si = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) { const v2f v = R[(msg * NW + ww) * 272 + j * 17 + i]; sr += v[0]; si += v[1]; }
        cf o; o.re = sr; o.im = si; if (msg == 0) p1[q] = o; else if (both) p2[q] = o;
    }
}
// slices a workgroup walks at a time (eight waves x half slices): PairGram2x16Item::spw must be a multiple
int pair_gram2x16_slices_at_a_time() { return 4; }
void launch_mfma_pair_gram2x16(hipStream_t s, const PairGram2x16Item* d_items, int nitems, int total_wgs) {
    if (total_wgs <= 0) return;
    const size_t lds = (size_t)8 * 8 * PS16 * sizeof(v2f);
    set_max_dynamic_lds((const void*)mfma_pair_gram2x16_kernel<true>, lds); hipLaunchKernelGGL(mfma_pair_gram2x16_kernel<true>, dim3(total_wgs), dim3(512), lds, s, d_items, nitems);
    TNQS_CHECK_LAUNCH();
}

}  // namespace tnqs
