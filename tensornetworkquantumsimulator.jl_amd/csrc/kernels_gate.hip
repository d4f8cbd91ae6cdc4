// kernels_gate.hip -- gate-path kernels of the bulk shape (d = 2, chi = 32): the third gauge leg absorbed INSIDE the f64 Gram.
//
// simple_update gauges a site with the square roots of its three incoming messages and takes a thin QR (src/Apply/simple_update.jl:43-48);
// here the QR is the Cholesky factor of G = psi~^dagger psi~.  Two of the three legs are absorbed by mfma_pair_kernel in
// one pass; the third one used to be a pass of its own (read + write of the whole tensor, HBM-bound) followed by the Gram pass.  A tile of
// the Gram kernel -- 64 fibers x 64 columns (s, b) -- always holds two complete fibers of the FASTEST outer leg r, so that last mode
// product is a (128 x 32) x (32 x 32) complex GEMM inside the tile: it runs on the f32 matrix cores, in place in LDS, before the f64
// Gram of the tile.  psi~ is rounded to f32 exactly where the separate pass rounded it, so G is still the exact Gram matrix of a rounded
// tensor.  One pass over the tensor instead of three.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdexcept>
#include <string>
#define TNQS_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) throw std::runtime_error(std::string("HIP kernel launch failed (") + __func__ + "): " + hipGetErrorString(e_)); } while (0)
#include "kernels.hpp"
#include "mfma_common.hpp"
#include "launch_util.hpp"
#include "x3_common.hpp"

namespace tnqs {

// Layout of a tile in LDS (as in mfma_gram64_f64_kernel): planes Xr / Xi, element (column c = s + 2 k, row u) at c * TRP + u; row
// u = r + 32 j: r = index of the fastest outer leg, j = 0 / 1 the two values of the next one that the tile holds.
// Work split of a workgroup (4 waves, one per SIMD; two workgroups per CU):
//   loads      thread (u = lane, wave w) loads the (s = 0, 1) pair of fiber u at the bond indices k = 8 w .. 8 w + 7 (16 bytes each): wave w
//              owns the 16 columns c = 16 w .. 16 w + 15 of the tile, all 64 rows -- commit and transform are wave private, no barrier between;
//   transform  X'[c][r' + 32 j] = sum_r X[c][r + 32 j] A[r][r']: per wave one 32 x 32 x 32 complex GEMM, rows i = (q = c - 16 w, j), on
//              v_mfma_f32_32x32x2_f32 with Gauss' three products; A operand 16 consecutive floats per lane (k-step t <-> r = t + 16 h),
//              B operand = A^T staged once per workgroup; the result is written over the wave's own rows;
//   Gram       the ten upper 16 x 16 blocks of G on v_mfma_f64_16x16x4_f64 in panel-sharing sets, 3M in f64 (gram_f64_shared), dealt to
//              the waves exactly as in mfma_gram64_f64_kernel<true, true> (same partial layout: 2 per chunk, one per tile parity).
// Pipeline: iteration t commits and transforms tile t + 1 in the other buffer, issues the loads of tile t + 2, then multiplies tile t;
// ONE workgroup barrier per tile.
// X3 (round 5, default): the transform on the bf16 matrix cores -- the tile rows split exactly into three bf16 pieces on the fly, the matrix A split once per
// workgroup into LDS in operand order (12 KiB in place of the 8 KiB f32 copy: 80 KiB per workgroup, two of them fill the 160 KiB of a CU exactly), four real
// products x six piece products: 48 instructions of 32 cycles instead of 48 of 64 (kernels_x3.hip; TNQS_NO_BF16X3=1: the f32 instructions)
template <bool X3>
__global__ __launch_bounds__(256, 2) void mfma_gauge_gram64_kernel(const GramItem* __restrict__ items, int nitems) {
    constexpr int TRP = 68, PLANE = 64 * TRP, NU = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const Xbuf = reinterpret_cast<float*>(smem);          // [buf][re | im][PLANE]
    float* const Mr = Xbuf + 4 * PLANE;                          // A[r][r'] at r * 32 + r' (re), then (im)
    float* const Mi = Mr + 1024;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    int lo = 0, hi = nitems - 1;
    const int gc = blockIdx.x;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (items[mid].chunk_begin <= gc) lo = mid; else hi = mid - 1; }
    const GramItem it = items[lo];
    const int lc = gc - it.chunk_begin;
    const int TA = it.TA, TB = it.TB;
    const long long PA = it.PA;
    const cf* __restrict__ Xg = reinterpret_cast<const cf*>(it.X);
    const cf* __restrict__ Mg = reinterpret_cast<const cf*>(it.M);
    const int ntiles = it.nta * it.ntb;
    const int t_begin = lc * it.tiles_per_chunk;
    const int t_end = min(ntiles, t_begin + it.tiles_per_chunk);
    // the matrix of the absorbed leg, transposed for the B operand: column-major A[r + 32 r'] -> Mr[r * 32 + r']
    u4* const Mp = reinterpret_cast<u4*>(Mr);                    // X3: [n][re h m l, im h m l][lane] x 16 bytes: B[k = r = 16 n + 8 (lane >> 5) + e][j = r' = lane & 31]
    if (X3) {
        if (tid < 128) {
            const int l = tid & 63, n = tid >> 6, rp = l & 31, r0 = 16 * n + 8 * (l >> 5);
            float xr[8], xi[8];
#pragma unroll
            for (int q = 0; q < 8; q += 2) { const v4f v = ldg4(Mg + r0 + q + 32 * rp); xr[q] = v[0]; xi[q] = v[1]; xr[q + 1] = v[2]; xi[q + 1] = v[3]; }
            const P3 pr = split8(xr), pi = split8(xi);
            u4* d = Mp + (size_t)(6 * n) * 64 + l;
            d[0] = pr.h; d[64] = pr.m; d[128] = pr.l; d[192] = pi.h; d[256] = pi.m; d[320] = pi.l;
        }
    } else {
        for (int e = tid; e < 1024; e += 256) { const cf v = ldgc(Mg + e); const int r = e & 31, rp = e >> 5; Mr[r * 32 + rp] = v.re; Mi[r * 32 + rp] = v.im; }
    }
    // thread -> fiber of the tile and its element offset (D = 2, K = 32):  2 (al + PA 32 bl)
    const int al = lane % TA, bl = lane / TA;
    const long long off = 2LL * (al + PA * 32LL * bl) + 2LL * PA * (8 * w);     // + kstride * first bond index of this wave
    const long long kstride = 2LL * PA;
    v4f px[NU];
    auto issue_loads = [&](int t) {
        const int ta = t % it.nta, tb = t / it.nta;
        const cf* p0 = Xg + 2LL * ((long long)ta * TA + PA * 32LL * ((long long)tb * TB)) + off;
#pragma unroll
        for (int j = 0; j < NU; ++j) px[j] = ldg4(p0 + kstride * j);
    };
    auto commit_loads = [&](int buf) {
        float* Xr = Xbuf + buf * (2 * PLANE); float* Xi = Xr + PLANE;
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            const int o0 = (2 * (8 * w + j)) * TRP + lane;
            Xr[o0] = px[j][0]; Xi[o0] = px[j][1]; Xr[o0 + TRP] = px[j][2]; Xi[o0 + TRP] = px[j][3];
        }
    };
    // in-place mode product of the wave's 16 columns (32 GEMM rows) with A
    const int gi = lane & 31, gh = lane >> 5;                   // A-operand row i = (q, j), k half
    const int abase = (16 * w + (gi & 15)) * TRP + 32 * (gi >> 4) + 16 * gh;
    auto transform = [&](int buf) {
        float* Xr = Xbuf + buf * (2 * PLANE); float* Xi = Xr + PLANE;
        if (X3) {
            // A operand: row i = gi, slots k = r = 16 n + 8 gh + e: eight consecutive floats of the row per instruction n
            const int ab3 = (16 * w + (gi & 15)) * TRP + 32 * (gi >> 4) + 8 * gh;
            v16f Re, Im;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                float ar[8], ai[8];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const v4f a = *reinterpret_cast<const v4f*>(Xr + ab3 + 16 * n + 4 * q), b = *reinterpret_cast<const v4f*>(Xi + ab3 + 16 * n + 4 * q);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { ar[4 * q + c] = a[c]; ai[4 * q + c] = b[c]; }
                }
                const P3 pxr = split8(ar), pxi = split8(ai);
                const u4* m = Mp + (size_t)(6 * n) * 64 + lane;
                P3 br, bi; br.h = m[0]; br.m = m[64]; br.l = m[128]; bi.h = m[192]; bi.m = m[256]; bi.l = m[320];
                if (n == 0) mac6x2<true>(Re, pxr, br, Im, pxr, bi); else mac6x2<false>(Re, pxr, br, Im, pxr, bi);
                mac6x2<false>(Re, neg(pxi), bi, Im, pxi, br);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * gh;            // GEMM row (q, j) of this accumulator register; column r' = gi
                const int o = (16 * w + (row & 15)) * TRP + 32 * (row >> 4) + gi;
                Xr[o] = Re[r]; Xi[o] = Im[r];
            }
            return;
        }
        v16f k1, k2, k3;
#pragma unroll
        for (int r = 0; r < 16; ++r) { k1[r] = 0.f; k2[r] = 0.f; k3[r] = 0.f; }
        v4f ar4[4], ai4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { ar4[q] = *reinterpret_cast<const v4f*>(Xr + abase + 4 * q); ai4[q] = *reinterpret_cast<const v4f*>(Xi + abase + 4 * q); }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float ar = ar4[t >> 2][t & 3], ai = ai4[t >> 2][t & 3];
            const float br = Mr[(t + 16 * gh) * 32 + gi], bi = Mi[(t + 16 * gh) * 32 + gi];
            k1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ar + ai, br, k1, 0, 0, 0);
            k2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, bi - br, k2, 0, 0, 0);
            k3 = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, br + bi, k3, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * gh;            // GEMM row (q, j) of this accumulator register; column r' = gi
            const int o = (16 * w + (row & 15)) * TRP + 32 * (row >> 4) + gi;
            Xr[o] = k1[r] - k3[r]; Xi[o] = k1[r] + k2[r];
        }
    };
    // Gram accumulators: set A (three blocks, tile parity parA) and set B (two blocks), see mfma_gram64_f64_kernel
    const int parA = (w < 2) ? 0 : 1;
    v4d CAr[3], CAi[3], CAc[3], CBr[2], CBi[2], CBc[2];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { CAr[j][r] = 0.0; CAi[j][r] = 0.0; CAc[j][r] = 0.0; }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { CBr[j][r] = 0.0; CBi[j][r] = 0.0; CBc[j][r] = 0.0; }
    // the pad columns of the planes are never read (rows 0..63 of a column only); nothing to clear
    lds_barrier();                                                  // A^T staged
    if (t_begin < t_end) { issue_loads(t_begin); commit_loads(0); if (t_begin + 1 < t_end) issue_loads(t_begin + 1); transform(0); }
    lds_barrier();
    for (int t = t_begin; t < t_end; ++t) {
        const int cur = (t - t_begin) & 1;
        if (t + 1 < t_end) {                                        // buffer cur^1 was last read by the Gram of tile t-1: every wave passed the barrier since
            commit_loads(cur ^ 1);
            if (t + 2 < t_end) issue_loads(t + 2);
            transform(cur ^ 1);
        }
        const float* Xr = Xbuf + cur * (2 * PLANE); const float* Xi = Xr + PLANE;
        if (((t - t_begin) & 1) == parA) {                         // wave-uniform
            const int q3[3] = {0, 1, 2};
            if (w & 1) gram_f64_shared<3, false, -1, true>(Xr, Xi, TRP, l15, kq, 3, q3, CAr, CAi, CAc);
            else       gram_f64_shared<3, true, 0, true>(Xr, Xi, TRP, l15, kq, 0, q3, CAr, CAi, CAc);
        } else if (w & 1) {
            const int q2[1] = {2}, q3b[1] = {3};
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                v4d r1[1] = {CBr[b]}, i1[1] = {CBi[b]}, c1[1] = {CBc[b]};
                if (b == 0) gram_f64_shared<1, true, 0, true>(Xr, Xi, TRP, l15, kq, 2, q2, r1, i1, c1);
                else        gram_f64_shared<1, true, 0, true>(Xr, Xi, TRP, l15, kq, 3, q3b, r1, i1, c1);
                CBr[b] = r1[0]; CBi[b] = i1[0]; CBc[b] = c1[0];
            }
        } else {
            const int q12[2] = {1, 2};
            gram_f64_shared<2, true, 0, true>(Xr, Xi, TRP, l15, kq, 1, q12, CBr, CBi, CBc);
        }
        lds_barrier();                                              // tile t consumed by everybody, tile t+1 transformed by everybody
    }
    struct alignas(16) cd { double re, im; };
    int aI[3], aJ[3], bI[2], bJ[2];
#pragma unroll
    for (int q = 0; q < 3; ++q) { aI[q] = (w & 1) ? q : 0; aJ[q] = (w & 1) ? 3 : q; }
    bI[0] = (w & 1) ? 2 : 1; bJ[0] = (w & 1) ? 2 : 1; bI[1] = (w & 1) ? 3 : 1; bJ[1] = (w & 1) ? 3 : 2;
    auto write_block = [&](cd* __restrict__ part, int I, int J, const v4d& cr, const v4d& ci, const v4d& cc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * I + kq + 4 * r, j = 16 * J + l15;
            cd v; v.re = cr[r] - ci[r]; v.im = cr[r] - cc[r]; part[i + (size_t)64 * j] = v;
            if (I != J) { cd c; c.re = v.re; c.im = -v.im; part[j + (size_t)64 * i] = c; }     // G[j][i] = conj(G[i][j])
        }
    };
    cd* __restrict__ partA = reinterpret_cast<cd*>(it.partial) + (size_t)(2 * lc + parA) * 4096;
    cd* __restrict__ partB = reinterpret_cast<cd*>(it.partial) + (size_t)(2 * lc + (parA ^ 1)) * 4096;
#pragma unroll
    for (int q = 0; q < 3; ++q) write_block(partA, aI[q], aJ[q], CAr[q], CAi[q], CAc[q]);
#pragma unroll
    for (int q = 0; q < 2; ++q) write_block(partB, bI[q], bJ[q], CBr[q], CBi[q], CBc[q]);
}

// the shape the kernel covers: d = 2, a 32-dimensional bond leg b and fastest outer leg r (the lowest leg that is not b), complete
// tiles of 64 fibers = (all 32 values of r) x (2 values of the next outer index)
bool gauge_gram64_covers(int d, int z, const int* chi, int bleg, int rleg) {
    if (d != 2 || z < 3 || bleg < 0 || bleg >= z || rleg < 0 || rleg >= z || rleg == bleg) return false;
    if (chi[bleg] != 32 || chi[rleg] != 32) return false;
    if (rleg != (bleg == 0 ? 1 : 0)) return false;
    long long pre = 1; for (int i = 0; i < bleg; ++i) pre *= chi[i];          // PA (site index excluded)
    long long post = 1; for (int i = bleg + 1; i < z; ++i) post *= chi[i];     // PB
    const long long TA = pre < 64 ? pre : 64, TB = 64 / TA;
    if (TA * TB != 64 || pre % TA != 0 || post % TB != 0 || post < TB) return false;
    return true;
}
void launch_mfma_gauge_gram64(hipStream_t s, const GramItem* d_items, int nitems, int total_chunks) {
    if (total_chunks <= 0) return;
    if (mfma_use_x3()) {
        const size_t lds = (size_t)(4 * 64 * 68) * sizeof(float) + 12 * 64 * 16;
        set_max_dynamic_lds((const void*)mfma_gauge_gram64_kernel<true>, lds);
        hipLaunchKernelGGL(mfma_gauge_gram64_kernel<true>, dim3(total_chunks), dim3(256), lds, s, d_items, nitems); TNQS_CHECK_LAUNCH();
        return;
    }
    const size_t lds = (size_t)(4 * 64 * 68 + 2048) * sizeof(float);
    set_max_dynamic_lds((const void*)mfma_gauge_gram64_kernel<false>, lds);
    hipLaunchKernelGGL(mfma_gauge_gram64_kernel<false>, dim3(total_chunks), dim3(256), lds, s, d_items, nitems); TNQS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// The same fusion for bond dimension 16 (BASELINE configs[3]: degree 6, chi = 16, five gauge legs): the last gauge leg r and the f64 Gram
// over 32 columns (s, k) in one pass, after two two-leg passes (mfma_pair16_kernel) instead of two two-leg passes + a single-leg pass + a
// Gram pass.  Here the whole kernel is WAVE PRIVATE: a unit is ONE complete fiber of r = 16 rows x 32 columns (4 KiB); a wave loads it
// (four 16-byte loads per lane, 256-byte runs), commits it to its own LDS slab, multiplies it with A_r in place
//      X'[c][r'] = sum_r X[c][r] A[r][r']          (32 x 16) x (16 x 16), v_mfma_f32_16x16x4_f32, Gauss' three products
// and accumulates the three upper 16 x 16 blocks of G over its own rows on v_mfma_f64_16x16x4_f64 (3M in f64).  No workgroup barrier in
// the loop; the four waves' accumulators are summed through LDS at the end (one partial per workgroup).  Layout: planes Xr / Xi of a
// slab, element (column c = s + 2 k, row r) at c * TRP + r, TRP = 20 (16-byte aligned operand rows; the 16 rows a b128 read of a
// quarter wave touches fall on 16 different 4-bank groups).
// Unit u of a site (K = 16, PA = product of the legs below the bond, site index excluded):
//      bond leg >= 1:  r = leg 0, u = a' + (PA / 16) b', element (s, r, k) at 2 (16 a' + 16 PA b') + s + 2 r + 2 PA k
//      bond leg == 0:  r = leg 1, u = b'',               element (s, r, k) at 512 u + s + 2 k + 32 r      (lanes run along k)
__global__ __launch_bounds__(256, 3) void mfma_gauge_gram32_kernel(const GramItem* __restrict__ items, int nitems) {
    constexpr int TRP = 20, PL = 32 * TRP, SLAB = 2 * PL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, kq = lane >> 4;
    float* const Xr = reinterpret_cast<float*>(smem) + w * SLAB; float* const Xi = Xr + PL;
    int lo = 0, hi = nitems - 1;
    const int gc = blockIdx.x;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (items[mid].chunk_begin <= gc) lo = mid; else hi = mid - 1; }
    const GramItem it = items[lo];
    const int lc = gc - it.chunk_begin;
    const long long PA = it.PA;
    const cf* __restrict__ Xg = reinterpret_cast<const cf*>(it.X);
    const cf* __restrict__ Mg = reinterpret_cast<const cf*>(it.M);
    const int nunits = it.nta * it.ntb;
    const int u_begin = lc * it.tiles_per_chunk, u_end = min(nunits, u_begin + it.tiles_per_chunk);
    const bool along_k = (PA == 1);                               // bond leg 0: the lanes run along k, the four loads along r
    const long long st0 = along_k ? 2 : 2, st1 = along_k ? 32 : 2 * PA;      // element strides of (lane & 15) and of (lane >> 4) + 4 i
    const int na = (int)(PA >> 4);                                // fibers of r per b' (bond leg >= 1)
    // B operand of the transform: A[r = 4 kq + t][r' = l15], element (r, r') at r + 16 r'; the three combinations of Gauss' product
    float br[4], bd[4], bs[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { const cf a = ldgc(Mg + (4 * kq + t) + 16 * l15); br[t] = a.re; bd[t] = a.im - a.re; bs[t] = a.re + a.im; }
    v4f pre[4];
    auto issue = [&](int u) {
        const long long base = along_k ? 512LL * u : 2LL * (16LL * (u % na) + 16LL * PA * (u / na));
        const cf* p = Xg + base + st0 * l15 + st1 * kq;
#pragma unroll
        for (int i = 0; i < 4; ++i) pre[i] = ldg4(p + 4 * st1 * i);
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = along_k ? l15 : kq + 4 * i, r = along_k ? kq + 4 * i : l15;
            const int o = (2 * k) * TRP + r;
            Xr[o] = pre[i][0]; Xi[o] = pre[i][1]; Xr[o + TRP] = pre[i][2]; Xi[o + TRP] = pre[i][3];
        }
    };
    v4d G00r = {0, 0, 0, 0}, G00i = G00r, G00c = G00r, G01r = G00r, G01i = G00r, G01c = G00r, G11r = G00r, G11i = G00r, G11c = G00r;
    int u = u_begin + w;
    if (u < u_end) issue(u);
    for (; u < u_end; u += 4) {
        commit();
        __builtin_amdgcn_wave_barrier();                          // LDS is in order per wave; only the compiler must not reorder
        if (u + 4 < u_end) issue(u + 4);
        // ---- X' = X A, in place: two blocks of 16 columns ----------------------------------------------------------------
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int ro = (16 * mb + l15) * TRP + 4 * kq;
            const v4f ar4 = *reinterpret_cast<const v4f*>(Xr + ro), ai4 = *reinterpret_cast<const v4f*>(Xi + ro);
            v4f k1 = {0.f, 0.f, 0.f, 0.f}, k2 = k1, k3 = k1;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                k1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ar4[t] + ai4[t], br[t], k1, 0, 0, 0);
                k2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ar4[t], bd[t], k2, 0, 0, 0);
                k3 = __builtin_amdgcn_mfma_f32_16x16x4f32(ai4[t], bs[t], k3, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {                         // C[row = column 4 kq + r of the block][col = r' = l15]
                const int o = (16 * mb + 4 * kq + r) * TRP + l15;
                Xr[o] = k1[r] - k3[r]; Xi[o] = k1[r] + k2[r];
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- G += X'^dagger X' over the 16 rows: blocks (0,0), (0,1), (1,1); lane (l15, kq) supplies rows 4 kq + c ------------------
        {
            const int r0 = l15 * TRP + 4 * kq, r1 = (16 + l15) * TRP + 4 * kq;
            const v4f p0 = *reinterpret_cast<const v4f*>(Xr + r0), p1 = *reinterpret_cast<const v4f*>(Xi + r0);
            const v4f q0 = *reinterpret_cast<const v4f*>(Xr + r1), q1 = *reinterpret_cast<const v4f*>(Xi + r1);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double pr = (double)p0[c], pi = (double)p1[c], qr = (double)q0[c], qi = (double)q1[c];
                const double ps = pr + pi, qs = qr + qi;
                G00r = __builtin_amdgcn_mfma_f64_16x16x4f64(ps, pr, G00r, 0, 0, 0);
                G00i = __builtin_amdgcn_mfma_f64_16x16x4f64(pi, pr - pi, G00i, 0, 0, 0);
                G00c = __builtin_amdgcn_mfma_f64_16x16x4f64(pr, ps, G00c, 0, 0, 0);
                G01r = __builtin_amdgcn_mfma_f64_16x16x4f64(ps, qr, G01r, 0, 0, 0);
                G01i = __builtin_amdgcn_mfma_f64_16x16x4f64(pi, qr - qi, G01i, 0, 0, 0);
                G01c = __builtin_amdgcn_mfma_f64_16x16x4f64(pr, qs, G01c, 0, 0, 0);
                G11r = __builtin_amdgcn_mfma_f64_16x16x4f64(qs, qr, G11r, 0, 0, 0);
                G11i = __builtin_amdgcn_mfma_f64_16x16x4f64(qi, qr - qi, G11i, 0, 0, 0);
                G11c = __builtin_amdgcn_mfma_f64_16x16x4f64(qr, qs, G11c, 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();                          // the slab has been consumed: the next commit may overwrite it
    }
    // ---- the four waves' accumulators -> one partial: (2, 3) -> LDS, (0, 1) add; 1 -> LDS, 0 adds and writes ---------------------
    struct alignas(16) cd { double re, im; };
    cd* const R = reinterpret_cast<cd*>(smem);                   // [2 waves][3 blocks][256]
    cd v[3][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        v[0][r].re = G00r[r] - G00i[r]; v[0][r].im = G00r[r] - G00c[r];
        v[1][r].re = G01r[r] - G01i[r]; v[1][r].im = G01r[r] - G01c[r];
        v[2][r].re = G11r[r] - G11i[r]; v[2][r].im = G11r[r] - G11c[r];
    }
    __syncthreads();                                              // every wave is done with its slab
    if (w >= 2) {
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) R[((w - 2) * 3 + b) * 256 + 64 * r + lane] = v[b][r];
    }
    __syncthreads();
    if (w < 2) {
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const cd o = R[(w * 3 + b) * 256 + 64 * r + lane]; v[b][r].re += o.re; v[b][r].im += o.im; }
    }
    __syncthreads();
    if (w == 1) {
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) R[b * 256 + 64 * r + lane] = v[b][r];
    }
    __syncthreads();
    if (w == 0) {
        cd* __restrict__ part = reinterpret_cast<cd*>(it.partial) + (size_t)lc * 1024;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int I = b == 2 ? 1 : 0, J = b == 0 ? 0 : 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const cd o = R[b * 256 + 64 * r + lane];
                cd g; g.re = v[b][r].re + o.re; g.im = v[b][r].im + o.im;
                const int i = 16 * I + kq + 4 * r, j = 16 * J + l15;
                part[i + 32 * j] = g;
                if (I != J) { cd c; c.re = g.re; c.im = -g.im; part[j + 32 * i] = c; }      // G[j][i] = conj(G[i][j])
            }
        }
    }
}

// the shape the kernel covers: d = 2, a 16-dimensional bond leg b, fastest outer leg r (the lowest leg that is not b) of dimension 16,
// whole fibers of r contiguous in 256-byte runs (everything below the bond is a multiple of 16 fibers, or the bond is leg 0)
bool gauge_gram32_covers(int d, int z, const int* chi, int bleg, int rleg) {
    if (d != 2 || z < 2 || bleg < 0 || bleg >= z || rleg < 0 || rleg >= z || rleg == bleg) return false;
    if (chi[bleg] != 16 || chi[rleg] != 16) return false;
    if (rleg != (bleg == 0 ? 1 : 0)) return false;
    return true;
}
int gauge_gram32_units(int z, const int* chi, int bleg) {
    long long n = 1; for (int i = 0; i < z; ++i) if (i != bleg) n *= chi[i];
    return (int)(n / 16);
}
void launch_mfma_gauge_gram32(hipStream_t s, const GramItem* d_items, int nitems, int total_chunks) {
    if (total_chunks <= 0) return;
    const size_t lds = 2 * 3 * 256 * 16;                          // the reduction of the accumulators (24 KiB) > the four slabs (20 KiB)
    set_max_dynamic_lds((const void*)mfma_gauge_gram32_kernel, lds);
    hipLaunchKernelGGL(mfma_gauge_gram32_kernel, dim3(total_chunks), dim3(256), lds, s, d_items, nitems); TNQS_CHECK_LAUNCH();
}

}  // namespace tnqs
