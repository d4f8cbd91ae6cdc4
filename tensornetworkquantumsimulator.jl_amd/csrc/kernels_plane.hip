// kernels_plane.hip -- CDNA4 (gfx950) plane kernels for bond dimension 16 (ComplexF32): the per-site shape of BASELINE configs[3]
// (periodic cubic lattice, degree 6, chi = 16: site tensors of 2 x 16^6 elements = 268 MB) and of the heavy-hex lattice.
//
// At chi = 16 one mode product has an arithmetic intensity of 8 flop/B (ridge of the machine: 19.7), so everything here is about HBM
// passes: two legs per pass (mfma_pair16_kernel, 16 flop/B) and both messages of a forest from one pass over (T, psi)
// (mfma_pair_gram2x16_kernel, 32 flop/B).  The kernels are WAVE PRIVATE: a wave owns an LDS slab, moves its own half slice (8 companion
// elements = 64-byte runs; the other half of every 128-byte line belongs to the neighbouring wave of the same workgroup, which walks
// the same slices at the same time) and never meets a workgroup barrier inside its loop, so the waves of a CU drift apart and one
// wave's global / LDS phases hide behind the others' MFMAs -- the structure that reached 80 % of the streaming rate for the single
// mode product (mfma_fiber_gemm_w_kernel), where the workgroup-synchronous chi = 32 plane kernels stop at 2.9 TB/s.
//
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
    float mxr[4], mxi[4], myr[4], myi[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        cf a = Mx[(4 * g4 + t) + 16 * c16]; mxr[t] = a.re; mxi[t] = a.im;
        cf b = My[(4 * g4 + t) + 16 * c16]; myr[t] = b.re; myi[t] = b.im;
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
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                Yr = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[t], mxr[t], Yr, 0, 0, 0);
                Yi = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[t], mxi[t], Yi, 0, 0, 0);
                Yr = __builtin_amdgcn_mfma_f32_16x16x4f32(-ai[t], mxi[t], Yr, 0, 0, 0);
                Yi = __builtin_amdgcn_mfma_f32_16x16x4f32(ai[t], mxr[t], Yi, 0, 0, 0);
            }
            v4f Sr = {0.f, 0.f, 0.f, 0.f}, Si = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {                                                     // Y reg r: row iy = 4 g + r, col jx = c16
                Sr = __builtin_amdgcn_mfma_f32_16x16x4f32(Yr[r], myr[r], Sr, 0, 0, 0);
                Si = __builtin_amdgcn_mfma_f32_16x16x4f32(Yr[r], myi[r], Si, 0, 0, 0);
                Sr = __builtin_amdgcn_mfma_f32_16x16x4f32(-Yi[r], myi[r], Sr, 0, 0, 0);
                Si = __builtin_amdgcn_mfma_f32_16x16x4f32(Yi[r], myr[r], Si, 0, 0, 0);
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
void launch_mfma_pair16(hipStream_t s, const Pair16Item* d_items, int nitems, int total_wgs) {
    if (total_wgs <= 0) return;
    const size_t lds = (size_t)8 * 8 * PS16 * sizeof(v2f);
    set_max_dynamic_lds((const void*)mfma_pair16_kernel, lds);
    hipLaunchKernelGGL(mfma_pair16_kernel, dim3(total_wgs), dim3(512), lds, s, d_items, nitems); TNQS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// BOTH messages a site sends into a linear forest, from one pass over the shared partial product X and psi = Y (16 x 16 planes):
//      out_y[b,b'] = sum_{c,jx} ( sum_ix X[c,ix,b] Mx[ix,jx] ) conj Y[c,jx,b']      (kept leg y, leg x absorbed)
//      out_x[d,d'] = sum_{c,jy} ( sum_iy X[c,d,iy] My[iy,jy] ) conj Y[c,d',jy]      (kept leg x, leg y absorbed)
// 32 flop/B: co-bound.  A workgroup has 4 waves (one per SIMD); a wave keeps the X and Y planes of its 8 companions resident together
// (2 x 18 KiB), prefetches the next unit into registers and accumulates both 16 x 16 messages in 16 registers.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mfma_pair_gram2x16_kernel(const PairGram2x16Item* __restrict__ items, int nitems) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c16 = lane & 15, g4 = lane >> 4;
    v2f* const L = reinterpret_cast<v2f*>(smem) + w * (16 * PS16);       // planes 0..7: X of companions 0..7, planes 8..15: Y
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
    float mxr[4], mxi[4], myr[4], myi[4];
    {
        const cf* __restrict__ Mx = reinterpret_cast<const cf*>(it.Mx); const cf* __restrict__ My = reinterpret_cast<const cf*>(it.My);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            cf a = Mx[(4 * g4 + t) + 16 * c16]; mxr[t] = a.re; mxi[t] = a.im;
            cf b = {0.f, 0.f}; if (My) b = My[(4 * g4 + t) + 16 * c16]; myr[t] = b.re; myi[t] = b.im;
        }
    }
    const bool both = it.My != nullptr;                 // My == null: only the message through ly is wanted (a site that sends one message in this level)
    v4f O1r = {0.f, 0.f, 0.f, 0.f}, O1i = O1r, O2r = O1r, O2i = O1r;
    const int f = lane & 3, ix0 = lane >> 2, half = w & 1;
    const long long toff = (long long)(4 * half + f) * g.cstr + g.sx * ix0;
    v2f* const lbase = L + (2 * f) * PS16 + ix0;
    v4f px[16], py[16];
    auto issue = [&](int sl) {
        const long long b = plane_slice_base(g, sl) + toff;
#pragma unroll
        for (int j = 0; j < 16; ++j) { px[j] = ldg4(Xg + b + g.sy * j); py[j] = ldg4(Yg + b + g.sy * j); }
    };
    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            v2f a = {px[j][0], px[j][1]}, b = {px[j][2], px[j][3]}, c = {py[j][0], py[j][1]}, d = {py[j][2], py[j][3]};
            lbase[P16 * j] = a; lbase[P16 * j + PS16] = b; lbase[P16 * j + 8 * PS16] = c; lbase[P16 * j + 9 * PS16] = d;
        }
    };
    // waves (0, 1) and (2, 3) take the two halves of slices s0, s0 + 2, ... / s0 + 1, s0 + 3, ...
    int sl = s_begin + (w >> 1);
    if (sl < s_end) issue(sl);
    for (; sl < s_end; sl += 2) {
        commit();
        __builtin_amdgcn_wave_barrier();
        if (sl + 2 < s_end) issue(sl + 2);
#pragma unroll 2
        for (int c = 0; c < 8; ++c) {
            const v2f* const PX = L + c * PS16; const v2f* const PY = L + (8 + c) * PS16;
            // ---- message through ly: absorb lx ------------------------------------------------------------------------
            {
                const v4f x01 = *reinterpret_cast<const v4f*>(PX + c16 * P16 + 4 * g4);      // B[k = ix = 4 g + t][j = b = iy = c16]
                const v4f x23 = *reinterpret_cast<const v4f*>(PX + c16 * P16 + 4 * g4 + 2);
                const float xr[4] = {x01[0], x01[2], x23[0], x23[2]}, xi[4] = {x01[1], x01[3], x23[1], x23[3]};
                v4f Cr = {0.f, 0.f, 0.f, 0.f}, Ci = Cr;                                       // C[jx = 4 g + r][b = c16]
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    Cr = __builtin_amdgcn_mfma_f32_16x16x4f32(mxr[t], xr[t], Cr, 0, 0, 0);
                    Ci = __builtin_amdgcn_mfma_f32_16x16x4f32(mxr[t], xi[t], Ci, 0, 0, 0);
                    Cr = __builtin_amdgcn_mfma_f32_16x16x4f32(-mxi[t], xi[t], Cr, 0, 0, 0);
                    Ci = __builtin_amdgcn_mfma_f32_16x16x4f32(mxi[t], xr[t], Ci, 0, 0, 0);
                }
                const v4f y01 = *reinterpret_cast<const v4f*>(PY + c16 * P16 + 4 * g4);      // B[k = jx = 4 g + r][j = b' = iy = c16]
                const v4f y23 = *reinterpret_cast<const v4f*>(PY + c16 * P16 + 4 * g4 + 2);
                const float yr[4] = {y01[0], y01[2], y23[0], y23[2]}, yi[4] = {y01[1], y01[3], y23[1], y23[3]};
#pragma unroll
                for (int r = 0; r < 4; ++r) {                                                 // out[b][b'] += C[jx][b] conj Y[jx][b']
                    O1r = __builtin_amdgcn_mfma_f32_16x16x4f32(Cr[r], yr[r], O1r, 0, 0, 0);
                    O1i = __builtin_amdgcn_mfma_f32_16x16x4f32(Ci[r], yr[r], O1i, 0, 0, 0);
                    O1r = __builtin_amdgcn_mfma_f32_16x16x4f32(Ci[r], yi[r], O1r, 0, 0, 0);
                    O1i = __builtin_amdgcn_mfma_f32_16x16x4f32(-Cr[r], yi[r], O1i, 0, 0, 0);
                }
            }
            // ---- message through lx: absorb ly (the same planes, read transposed) ------------------------------------------
            if (both) {
                float xr[4], xi[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) { const v2f x = PX[(4 * g4 + t) * P16 + c16]; xr[t] = x[0]; xi[t] = x[1]; }   // B[k = iy = 4 g + t][j = d = ix = c16]
                v4f Cr = {0.f, 0.f, 0.f, 0.f}, Ci = Cr;                                       // C[jy = 4 g + r][d = c16]
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    Cr = __builtin_amdgcn_mfma_f32_16x16x4f32(myr[t], xr[t], Cr, 0, 0, 0);
                    Ci = __builtin_amdgcn_mfma_f32_16x16x4f32(myr[t], xi[t], Ci, 0, 0, 0);
                    Cr = __builtin_amdgcn_mfma_f32_16x16x4f32(-myi[t], xi[t], Cr, 0, 0, 0);
                    Ci = __builtin_amdgcn_mfma_f32_16x16x4f32(myi[t], xr[t], Ci, 0, 0, 0);
                }
                float yr[4], yi[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { const v2f y = PY[(4 * g4 + r) * P16 + c16]; yr[r] = y[0]; yi[r] = y[1]; }   // B[k = jy = 4 g + r][j = d' = c16]
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    O2r = __builtin_amdgcn_mfma_f32_16x16x4f32(Cr[r], yr[r], O2r, 0, 0, 0);
                    O2i = __builtin_amdgcn_mfma_f32_16x16x4f32(Ci[r], yr[r], O2i, 0, 0, 0);
                    O2r = __builtin_amdgcn_mfma_f32_16x16x4f32(Ci[r], yi[r], O2r, 0, 0, 0);
                    O2i = __builtin_amdgcn_mfma_f32_16x16x4f32(-Cr[r], yi[r], O2i, 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();                 // the planes have been consumed: the next commit may overwrite them
    }
    // one partial per workgroup and message: the four waves' accumulators are summed through LDS, in wave order
    cf* __restrict__ p1 = reinterpret_cast<cf*>(it.partial_y) + (size_t)lw * 256;
    cf* __restrict__ p2 = reinterpret_cast<cf*>(it.partial_x) + (size_t)lw * 256;
    __syncthreads();                                     // every wave is done with its slab
    v2f* const R = reinterpret_cast<v2f*>(smem);        // [message 2][wave 4][16 x 17]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * g4 + r;                        // element (i, j = c16)
        v2f a = {O1r[r], O1i[r]}, b = {O2r[r], O2i[r]};
        R[(0 * 4 + w) * 272 + c16 * 17 + i] = a; R[(1 * 4 + w) * 272 + c16 * 17 + i] = b;
    }
    __syncthreads();
    for (int e = tid; e < 512; e += 256) {
        const int msg = e >> 8, q = e & 255, i = q & 15, j = q >> 4;
        float sr = 0.f, si = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) { const v2f v = R[(msg * 4 + ww) * 272 + j * 17 + i]; sr += v[0]; si += v[1]; }
        cf o; o.re = sr; o.im = si; if (msg == 0) p1[q] = o; else if (both) p2[q] = o;
    }
}
void launch_mfma_pair_gram2x16(hipStream_t s, const PairGram2x16Item* d_items, int nitems, int total_wgs) {
    if (total_wgs <= 0) return;
    const size_t lds = (size_t)4 * 16 * PS16 * sizeof(v2f);
    set_max_dynamic_lds((const void*)mfma_pair_gram2x16_kernel, lds);
    hipLaunchKernelGGL(mfma_pair_gram2x16_kernel, dim3(total_wgs), dim3(256), lds, s, d_items, nitems); TNQS_CHECK_LAUNCH();
}

}  // namespace tnqs
