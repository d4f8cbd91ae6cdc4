// mfma_common.hpp -- device-side helpers shared by the MFMA translation units (kernels_mfma.hip, kernels_plane.hip, kernels_chi64.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace tnqs {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
struct alignas(8) cf { float re, im; };

// GLOBAL-memory loads / stores.  The tensor pointers reach the kernels through item structs in device memory, so the compiler knows
// nothing about their address space and emits FLAT instructions; a flat load counts on BOTH vmcnt and lgkmcnt and returns out of order
// with respect to LDS traffic, so the first `s_waitcnt lgkmcnt(0)` in front of an LDS-fed MFMA also waits for every prefetch load in
// flight -- the "prefetch" is synchronous and a full memory latency is exposed per tile (seen in the ISA of every kernel of round 1).
// Casting to address space 1 gives global_load / global_store, which count on vmcnt only.
#define TNQS_AS1 __attribute__((address_space(1)))
__device__ __forceinline__ v4f ldg4(const void* p) { return *(const v4f TNQS_AS1*)(p); }
__device__ __forceinline__ v2f ldg2(const void* p) { return *(const v2f TNQS_AS1*)(p); }
__device__ __forceinline__ void stg4(void* p, v4f v) { *(v4f TNQS_AS1*)(p) = v; }
__device__ __forceinline__ void stg2(void* p, v2f v) { *(v2f TNQS_AS1*)(p) = v; }
__device__ __forceinline__ cf ldgc(const cf* p) { const v2f t = ldg2(p); cf r; r.re = t[0]; r.im = t[1]; return r; }
__device__ __forceinline__ void stgc(cf* p, cf v) { v2f t = {v.re, v.im}; stg2(p, t); }

// LDS-only workgroup barrier: __syncthreads() also drains vmcnt (global loads AND stores in flight), which would
// serialise the prefetch / store streams against the LDS hand-offs (cdna_hip_programming.md, "Pipelining across barriers").
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ------------------------------------------------------------------------------------------------------------
// complex 32 x 32 accumulator tile for v_mfma_f32_32x32x2_f32 (k = 2 per instruction: lanes 0..31 hold k = 0, lanes 32..63 k = 1).
//   M3 = false: the textbook product, four real MFMAs per complex rank-2 update, two accumulators (re, im).
//   M3 = true : Gauss' three-multiplication product, three real MFMAs and three accumulators
//                   k1 = (ar + ai) br,   k2 = ar (bi - br),   k3 = ai (br + bi);      re = k1 - k3,  im = k1 + k2
//               The operand sums cost three VALU instructions per update (they overlap the 3 x 64 matrix-core cycles); the recombination
//               is two VALU instructions per accumulator register, once per tile.  Error: |delta| <= c u (|a_r| + |a_i|)(|b_r| + |b_i|) per
//               term, i.e. the same NORMWISE bound as the four-multiplication product; the componentwise bound of the imaginary part is
//               lost (a tiny imaginary part next to a large real one carries the large one's rounding), which the BP messages and site
//               tensors -- compared and used normwise everywhere -- do not rely on.  TNQS_NO_3M=1 selects M3 = false everywhere.
// ------------------------------------------------------------------------------------------------------------
template <bool M3> struct CAcc32 {
    v16f a, b, c;
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int r = 0; r < 16; ++r) { a[r] = 0.f; b[r] = 0.f; if (M3) c[r] = 0.f; }
    }
    // acc += (ar + i ai) (br + i bi)
    __device__ __forceinline__ void mac(float ar, float ai, float br, float bi) {
        if (M3) {
            a = __builtin_amdgcn_mfma_f32_32x32x2f32(ar + ai, br, a, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, bi - br, c, 0, 0, 0);
            b = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, br + bi, b, 0, 0, 0);
        } else {
            a = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, br, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x2f32(-ai, bi, a, 0, 0, 0);
            b = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, bi, b, 0, 0, 0);
            b = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, br, b, 0, 0, 0);
        }
    }
    // the same update with the A-side combination supplied by the caller (a0 = ar + ai when M3, -ai otherwise: a constant of the wave in
    // the kernels that keep their matrix in registers); FIRST: the accumulators start from zero (inline constant, no register clearing)
    template <bool FIRST> __device__ __forceinline__ void mac_pre(float a0, float ar, float ai, float br, float bi) {
        const v16f z = (v16f)(0.f);
        if (M3) {
            a = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, br, FIRST ? z : a, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, bi - br, FIRST ? z : c, 0, 0, 0);
            b = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, br + bi, FIRST ? z : b, 0, 0, 0);
        } else {
            a = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, br, FIRST ? z : a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bi, a, 0, 0, 0);
            b = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, bi, FIRST ? z : b, 0, 0, 0);
            b = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, br, b, 0, 0, 0);
        }
    }
    // the same update with the B-side combinations supplied by the caller: (bd, bs) = (bi - br, br + bi) when M3, (bi, -bi) otherwise
    template <bool FIRST> __device__ __forceinline__ void mac_bpre(float ar, float ai, float br, float bd, float bs) {
        const v16f z = (v16f)(0.f);
        if (M3) {
            a = __builtin_amdgcn_mfma_f32_32x32x2f32(ar + ai, br, FIRST ? z : a, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, bd, FIRST ? z : c, 0, 0, 0);
            b = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, bs, FIRST ? z : b, 0, 0, 0);
        } else {
            a = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, br, FIRST ? z : a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, bs, a, 0, 0, 0);
            b = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, bd, FIRST ? z : b, 0, 0, 0);
            b = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, br, b, 0, 0, 0);
        }
    }
    // (re, im) of every accumulator register, in place: a <- re, b <- im  (after mac / mac_pre / mac_bpre)
    __device__ __forceinline__ void finish() {
        if (M3) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float k1 = a[r]; a[r] = k1 - b[r]; b[r] = k1 + c[r]; }
        }
    }
    // the same, one register at a time (the accumulators stay as they are): for accumulator tiles that are the next product's A operand
    __device__ __forceinline__ float re(int r) const { return M3 ? a[r] - b[r] : a[r]; }
    __device__ __forceinline__ float im(int r) const { return M3 ? a[r] + c[r] : b[r]; }
    // acc += (ar + i ai) conj(br + i bi); M3 keeps  sum (ar + ai) br,  sum ar (bi + br),  sum ai (br - bi)  -- no negated operand -- and
    // finish_conj() recombines  re = k1 - k3,  im = k1 - sum ar (bi + br)
    __device__ __forceinline__ void mac_conj(float ar, float ai, float br, float bi) {
        if (M3) {
            a = __builtin_amdgcn_mfma_f32_32x32x2f32(ar + ai, br, a, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, bi + br, c, 0, 0, 0);
            b = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, br - bi, b, 0, 0, 0);
        } else {
            a = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, br, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, bi, a, 0, 0, 0);
            b = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, br, b, 0, 0, 0);
            b = __builtin_amdgcn_mfma_f32_32x32x2f32(-ar, bi, b, 0, 0, 0);
        }
    }
    __device__ __forceinline__ void finish_conj() {
        if (M3) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float k1 = a[r]; a[r] = k1 - b[r]; b[r] = k1 - c[r]; }
        }
    }
};

// ------------------------------------------------------------------------------------------------------------
// f64 Gram tiles (v_mfma_f64_16x16x4_f64; LDS tile: planes Xr / Xi, element (column c, tile row r) at c * TRP + r; lane = (l15, kq): column
// l15 of a 16-column panel, tile rows 16 kq .. 16 kq + 15).  A set of upper-triangle blocks that share one panel P: ROW = P supplies the rows
// (A operand) of every block and Q[j] the columns, !ROW = P supplies the columns (B operand) and Q[j] the rows; DIAGJ >= 0: block DIAGJ is (P, P).
// The shared panel's values are read from LDS, converted to f64 and combined ONCE per four k-steps for all N blocks (a block on its own
// converts four values and forms three sums per k-step).  out[i][j] += x[i] conj(x[j]); M3: (cr, ci, cc) accumulate
// sum (ar+ai) br, sum ai (br-bi), sum ar (bi+br), i.e. re = cr - ci, im = cr - cc.
// ------------------------------------------------------------------------------------------------------------
typedef double v4d __attribute__((ext_vector_type(4)));
template <int N, bool ROW, int DIAGJ, bool M3>
__device__ __forceinline__ void gram_f64_shared(const float* __restrict__ Xr, const float* __restrict__ Xi, int TRP, int l15, int kq, int P,
                                                const int (&Q)[N], v4d (&cr)[N], v4d (&ci)[N], v4d (&cc)[N]) {
#pragma unroll 1
    for (int hq = 0; hq < 4; ++hq) {                              // tile rows 16 kq + 4 hq .. + 3
        const int ro = (16 * P + l15) * TRP + 16 * kq + 4 * hq;
        const v4f p0 = *reinterpret_cast<const v4f*>(Xr + ro), p1 = *reinterpret_cast<const v4f*>(Xi + ro);
        double pr[4], pi[4], ps[4], pd[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { pr[c] = (double)p0[c]; pi[c] = (double)p1[c]; ps[c] = pr[c] + pi[c]; pd[c] = pr[c] - pi[c]; }
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const bool dg = (j == DIAGJ);
            v4f o0 = p0, o1 = p1;
            if (!dg) { const int rb = (16 * Q[j] + l15) * TRP + 16 * kq + 4 * hq; o0 = *reinterpret_cast<const v4f*>(Xr + rb); o1 = *reinterpret_cast<const v4f*>(Xi + rb); }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double qr = dg ? pr[c] : (double)o0[c], qi = dg ? pi[c] : (double)o1[c];
                if (M3) {
                    if (ROW) {                                       // a = p, b = q
                        cr[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ps[c], qr, cr[j], 0, 0, 0);
                        ci[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(pi[c], qr - qi, ci[j], 0, 0, 0);
                        cc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(pr[c], qi + qr, cc[j], 0, 0, 0);
                    } else {                                         // a = q, b = p
                        cr[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(qr + qi, pr[c], cr[j], 0, 0, 0);
                        ci[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(qi, pd[c], ci[j], 0, 0, 0);
                        cc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(qr, ps[c], cc[j], 0, 0, 0);
                    }
                } else {
                    const double ar = ROW ? pr[c] : qr, ai = ROW ? pi[c] : qi, br = ROW ? qr : pr[c], bi = ROW ? qi : pi[c];
                    cr[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, cr[j], 0, 0, 0);
                    ci[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, br, ci[j], 0, 0, 0);
                    cr[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, bi, cr[j], 0, 0, 0);
                    ci[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ar, bi, ci[j], 0, 0, 0);
                }
            }
        }
    }
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------------------------------------------------
// tile <-> thread map (division-free inner loops).  A tile holds TA x TB fibers; one "unit" is VEC memory-adjacent
// elements of one k-slice ((s=0,s=1) of a fiber when D == 2, fibers (a, a+1) when D == 1 and TA is even).  Thread t
// owns unit u = t % U of the k-slices kp, kp+KP, ... .
// ------------------------------------------------------------------------------------------------------------
struct TileMap {
    int U, KP, u, kp;          // units per k-slice, k phases, this thread's unit / first k
    int row0, row1, c0, c1;    // LDS row and (s) column offset of the unit's two elements (row1 < 0: single)
    int al, bl, al1;           // tile-local fiber coordinates (validity tests)
    int vec;                   // elements per unit (1 or 2)
    long long off;             // element offset of the unit inside the tile's origin (k = 0)
    bool active;
};
__device__ __forceinline__ TileMap make_map(int tid, int D, int TA, int TB, long long PA, int K) {
    TileMap m;
    const int rows = TA * TB;
    m.vec = (D == 2) ? 2 : ((D == 1 && (TA % 2 == 0) && (PA % 2 == 0)) ? 2 : 1);
    m.U = D * rows / m.vec;
    m.KP = m.U >= 256 ? 1 : 256 / m.U;
    m.u = tid % m.U; m.kp = tid / m.U;
    m.active = tid < m.U * m.KP;
    int e0 = m.u * m.vec;                 // first element index in (s, al, bl) order
    int s = e0 % D; int row = e0 / D;
    m.al = row % TA; m.bl = row / TA;
    m.row0 = row; m.c0 = s;
    if (m.vec == 2) { if (D == 2) { m.row1 = row; m.c1 = 1; m.al1 = m.al; } else { m.row1 = row + 1; m.c1 = 0; m.al1 = m.al + 1; } }
    else { m.row1 = -1; m.c1 = 0; m.al1 = m.al; }
    m.off = s + (long long)D * (m.al + PA * (long long)K * m.bl);
    return m;
}

__device__ __forceinline__ TileMap make_map_wave(int lane, int D, int TA, int TB, long long PA, int K) {
    TileMap m;
    const int rows = TA * TB;
    m.vec = (D == 2) ? 2 : ((D == 1 && (TA % 2 == 0) && (PA % 2 == 0)) ? 2 : 1);
    m.U = D * rows / m.vec;
    m.KP = m.U >= 64 ? 1 : 64 / m.U;
    m.u = lane % m.U; m.kp = lane / m.U;
    m.active = lane < m.U * m.KP;
    int e0 = m.u * m.vec;
    int s = e0 % D; int row = e0 / D;
    m.al = row % TA; m.bl = row / TA;
    m.row0 = row; m.c0 = s;
    if (m.vec == 2) { if (D == 2) { m.row1 = row; m.c1 = 1; m.al1 = m.al; } else { m.row1 = row + 1; m.c1 = 0; m.al1 = m.al + 1; } }
    else { m.row1 = -1; m.c1 = 0; m.al1 = m.al; }
    m.off = s + (long long)D * (m.al + PA * (long long)K * m.bl);
    return m;
}


}  // namespace tnqs
