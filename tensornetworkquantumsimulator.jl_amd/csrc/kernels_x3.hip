// kernels_x3.hip -- the chi = 32 plane kernels of the BP level with the f32 products carried by the bf16 matrix cores (round 5).
//
// v_mfma_f32_32x32x2_f32 runs at the f32 VECTOR rate (64 cycles per SIMD for 4096 multiply-adds); v_mfma_f32_32x32x16_bf16 multiplies
// 32768 bf16 pairs in 32 cycles -- sixteen times the rate, products exact (8 x 8 significand bits), accumulation in f32 exactly as the f32
// instruction accumulates.  An f32 number IS the sum of three bf16 numbers:
//      x = h + m + l,   h = x truncated to its leading 8 significand bits,  m = (x - h) truncated likewise,  l = x - h - m   (all exact: 8 + 8 + 8 = 24 bits)
// so an f32 product is the sum of nine exact bf16 products; the three smallest ones (m l, l m, l l: <= 2^-24 |x y|, the size of ONE f32 rounding
// of the product) are dropped, the other six are accumulated in f32:
//      x y  =  h h' + (h m' + m h') + (m m' + h l' + l h')  +  O(2^-24 |x y|)
// i.e. an f32 multiply-add chain with the product rounded to 24 bits before it is added -- the error model of an UNFUSED f32 multiply-add, normwise
// the same class as the three-multiplication f32 product this replaces (mfma_common.hpp).  Six bf16 instructions of 32 cycles replace sixteen f32
// instructions of 64 (k = 16 against k = 2): 5.3 x the matrix rate per real product; with the textbook four real products per complex one (no operand
// sums to split) 4 x 6 x 32 = 768 cycles per 16 complex k against 3 x 8 x 64 = 1536 of the 3M f32 route.  The splitting costs 5.5 vector
// instructions per f32 value (and, subtract, and, subtract, 1.5 byte-permutes to pack pairs), on the VALU next to the matrix pipe.
// The kernels keep the data movement of the f32 kernels they replace (kernels_mfma.hip); with the matrix time halved they are bound by HBM.
// TNQS_NO_BF16X3=1 selects the f32 kernels.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <type_traits>
#define TNQS_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) throw std::runtime_error(std::string("HIP kernel launch failed (") + __func__ + "): " + hipGetErrorString(e_)); } while (0)
#include "kernels.hpp"
#include "mfma_common.hpp"
#include "launch_util.hpp"
#include "x3_common.hpp"

namespace tnqs {

__device__ __forceinline__ long long x3_slice_base(const PairGeom& g, int sl) {
    int a0 = sl % g.n0; int r1 = sl / g.n0; int a1 = r1 % g.n1; int a2 = r1 / g.n1;
    return (long long)a0 * g.t0 + (long long)a1 * g.t1 + (long long)a2 * g.t2;
}

// ------------------------------------------------------------------------------------------------------------
// BOTH messages a degree-4 site sends into a linear forest (the pass of mfma_pair_gram2_kernel, kernels_mfma.hip: same items, same partials):
//      step 1  C[j'][kept] = sum_k M[k][j'] X[k][kept]      step 2  O[kept][kept'] += sum_j' C[j'][kept] conj Y[j'][kept']
// HALF slices: a workgroup takes 8 of a slice's 16 companions per phase (workgroups lw and lw + 8 of a group of 16 sit on the same XCD and take the two
// 64-byte halves of every line at the same time) -- the f32 kernel walks quarters (32-byte pieces), and with the matrix time halved it is the number of
// memory requests in flight, not the matrix pipe, that bounds the pass (measured: 3.1 TB/s with quarters and the matrix instructions removed).  The 16 planes
// of a phase (8 X + 8 Y, 137 KiB) fill the LDS ONCE; the next phase's planes wait in registers (16 x 16 bytes per thread, issued at the start of the phase:
// a whole phase to arrive) and are committed between two barriers.  Wave w: message m = w >> 2, companions (w & 3) and (w & 3) + 4, one after the other.
// One v_mfma_f32_32x32x16_bf16 covers sixteen k: lane (ln, h) supplies eight of them,
//      step 1:  k  = 16 n + 8 h + e                       (n = 0, 1: the instruction, e = 0..7 the slot): eight CONSECUTIVE plane elements
//      step 2:  j' = 4 h + 8 (2 n + (e >> 2)) + (e & 3)   = the rows of C this lane's accumulator registers 8 n + e hold -- C never leaves the registers
// M lives in registers as split A operands (48 registers); X, Y and C are split on the fly.  Planes in LDS with an EVEN pitch (34) so that a lane's
// eight consecutive elements are four aligned 16-byte reads (message 0; message 1 reads the same planes transposed, 8-byte reads along the lanes).
// ------------------------------------------------------------------------------------------------------------
// SINGLE: one message per item (partial_y only; My, partial_x unused) -- the eight waves take the eight companions of a phase, one each
template <int MODE, bool SINGLE = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void x3_pair_gram2_kernel(const PairGram2Item* __restrict__ items, int nitems) {
    constexpr int P = 34;                      // row pitch in complex elements
    constexpr int PS = 32 * P + 2;             // plane stride: X planes of companions 0..7, then their Y planes
    // scheduling barrier: VALU / SALU may cross; LDS, global-memory and matrix instructions keep their program order (left to itself the compiler sinks the
    // global loads to the end of the phase, directly in front of the commit that waits for them, and the LDS reads directly in front of their first use)
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
    const int lw = gw - it.wg_begin;                        // wg_begin is a multiple of 16
    const int half = (lw >> 3) & 1, pw = ((lw >> 4) << 3) | (lw & 7);
    const int s_begin = pw * it.spw, s_end = min(nslices, s_begin + it.spw);
    const cf* __restrict__ Xg = reinterpret_cast<const cf*>(it.X);
    const cf* __restrict__ Yg = reinterpret_cast<const cf*>(it.Y);
    const int comp = SINGLE ? w : (w & 3), msg = SINGLE ? 0 : (w >> 2);
    // A operands of step 1: A[i = j' = ln][k] = M[k + 32 ln], k = 16 n + 8 h + e
    P3 Mr[2], Mi[2];
    {
        const cf* __restrict__ M = reinterpret_cast<const cf*>(msg ? it.My : it.Mx) + 32 * ln + 8 * h;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            float xr[8], xi[8];
#pragma unroll
            for (int e = 0; e < 8; e += 2) { const v4f v = ldg4(M + 16 * n + e); xr[e] = v[0]; xi[e] = v[1]; xr[e + 1] = v[2]; xi[e + 1] = v[3]; }
            Mr[n] = split8<MODE>(xr); Mi[n] = split8<MODE>(xi);
        }
    }
    // mover: thread -> (companion pair f4 = 0..3 of the half, first segment sg0 = 0..127); segment j: ix = sg0 & 31, iy = (sg0 >> 5) + 4 j
    const int f4 = tid & 3, sg0 = tid >> 2;
    const int ix0 = sg0 & 31, iy0 = sg0 >> 5;
    const long long toff = (long long)(4 * half + f4) * g.cstr + g.sx * ix0 + g.sy * iy0, tstr = 4 * g.sy;
    v2f* const lbase = L + (2 * f4) * PS + iy0 * P + ix0;                // element (ix, iy) at [iy][ix]
    v4f px[8], py[8];
    auto commit = [&]() {
        if (MODE == 3 || MODE == 4) return;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v2f* p0 = lbase + 4 * P * j;                                 // iy advances by 4 per j
            v2f a; a[0] = px[j][0]; a[1] = px[j][1]; v2f b; b[0] = px[j][2]; b[1] = px[j][3];
            p0[0] = a; p0[PS] = b;
            v2f c; c[0] = py[j][0]; c[1] = py[j][1]; v2f d; d[0] = py[j][2]; d[1] = py[j][3];
            p0[8 * PS] = c; p0[9 * PS] = d;
        }
    };
    v16f Or, Oi;                               // O_re = sum Cr Yr + Ci Yi;  O_im = sum Ci Yr + (-Cr) Yi
#pragma unroll
    for (int r = 0; r < 16; ++r) { Or[r] = 0.f; Oi[r] = 0.f; }
    if (s_begin < s_end) {
        const long long b = x3_slice_base(g, s_begin) + toff;
#pragma unroll
        for (int j = 0; j < 8; ++j) { px[j] = ldg4(Xg + b + tstr * j); py[j] = ldg4(Yg + b + tstr * j); }
        commit();
    }
    lds_barrier();
    auto run = [&](auto msg_c) {
    constexpr int MSG = decltype(msg_c)::value;
    // eight plane elements (index i0 .. i0 + 3 and i1 .. i1 + 3 along the contracted direction) of the kept index ln: message 0 reads rows, message 1 columns
    auto read8 = [&](const v2f* PL, int i0, int i1, float (&re)[8], float (&im)[8]) {
        if (MODE == 4) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { re[e] = __int_as_float(i0 + e + ln); im[e] = __int_as_float(i1 + e); }
            return;
        }
        if (MSG == 0) {
            const v4f* a = reinterpret_cast<const v4f*>(PL + ln * P + i0); const v4f* b = reinterpret_cast<const v4f*>(PL + ln * P + i1);
            const v4f a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
            re[0] = a0[0]; im[0] = a0[1]; re[1] = a0[2]; im[1] = a0[3]; re[2] = a1[0]; im[2] = a1[1]; re[3] = a1[2]; im[3] = a1[3];
            re[4] = b0[0]; im[4] = b0[1]; re[5] = b0[2]; im[5] = b0[3]; re[6] = b1[0]; im[6] = b1[1]; re[7] = b1[2]; im[7] = b1[3];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const v2f u = PL[(i0 + e) * P + ln], v = PL[(i1 + e) * P + ln]; re[e] = u[0]; im[e] = u[1]; re[4 + e] = v[0]; im[4 + e] = v[1]; }
        }
    };
    for (int sl = s_begin; sl < s_end; ++sl) {
        const bool more = sl + 1 < s_end;
        const long long nb = x3_slice_base(g, more ? sl + 1 : sl) + toff;          // (the last phase re-reads its own slice: no branch in the stream)
        float ar[8], ai[8], br[8], bi[8];
        read8(L + comp * PS, 8 * h, 8 * h + 4, ar, ai);
        if (MODE == 5) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { px[j] = ldg4(Xg + nb + tstr * j); py[j] = ldg4(Yg + nb + tstr * j); }
        }
#pragma unroll
        for (int t = 0; t < (SINGLE ? 1 : 2); ++t) {
            const v2f* PX = L + (comp + 4 * t) * PS; const v2f* PY = PX + 8 * PS;
            // ---- step 1: C[j'][kept] = sum_k M[k][j'] X[k][kept] -------------------------------------------------------------------------
            v16f Cr, Ci;
            read8(PX, 16 + 8 * h, 16 + 8 * h + 4, br, bi);
            if (t == 0 && (MODE < 3)) {
#pragma unroll
                for (int j = 0; j < 2; ++j) { px[j] = ldg4(Xg + nb + tstr * j); py[j] = ldg4(Yg + nb + tstr * j); }
            }
            TNQS_PIN();
            {
                const P3 xr = split8<MODE>(ar), xi = split8<MODE>(ai);
                mac6x2<true, MODE>(Cr, Mr[0], xr, Ci, Mr[0], xi);
                mac6x2<false, MODE>(Cr, Mi[0], neg(xi), Ci, Mi[0], xr);
            }
            read8(PY, 4 * h, 4 * h + 8, ar, ai);                                         // first operands of step 2
            if (t == 0 && (MODE < 3)) {
#pragma unroll
                for (int j = 2; j < 4; ++j) { px[j] = ldg4(Xg + nb + tstr * j); py[j] = ldg4(Yg + nb + tstr * j); }
            }
            TNQS_PIN();
            {
                const P3 xr = split8<MODE>(br), xi = split8<MODE>(bi);
                mac6x2<false, MODE>(Cr, Mr[1], xr, Ci, Mr[1], xi);
                mac6x2<false, MODE>(Cr, Mi[1], neg(xi), Ci, Mi[1], xr);
            }
            // ---- step 2: O[kept][kept'] += sum_j' C[j'][kept] conj Y[j'][kept'] ----------------------------------------------------------
            read8(PY, 16 + 4 * h, 16 + 4 * h + 8, br, bi);
            if (t == 0 && (MODE < 3)) {
#pragma unroll
                for (int j = 4; j < 6; ++j) { px[j] = ldg4(Xg + nb + tstr * j); py[j] = ldg4(Yg + nb + tstr * j); }
            }
            TNQS_PIN();
            {
                float cr[8], ci[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { cr[e] = Cr[e]; ci[e] = Ci[e]; }
                const P3 pcr = split8<MODE>(cr), pci = split8<MODE>(ci), yr = split8<MODE>(ar), yi = split8<MODE>(ai);
                mac6x2<false, MODE>(Or, pcr, yr, Oi, pci, yr);
                mac6x2<false, MODE>(Or, pci, yi, Oi, neg(pcr), yi);
            }
            if (t == 0 && (MODE < 3)) {
#pragma unroll
                for (int j = 6; j < 8; ++j) { px[j] = ldg4(Xg + nb + tstr * j); py[j] = ldg4(Yg + nb + tstr * j); }
            }
            if (t == 0 && !SINGLE) read8(L + (comp + 4) * PS, 8 * h, 8 * h + 4, ar, ai);            // first operands of the second companion
            TNQS_PIN();
            {
                float cr[8], ci[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { cr[e] = Cr[8 + e]; ci[e] = Ci[8 + e]; }
                const P3 pcr = split8<MODE>(cr), pci = split8<MODE>(ci), yr = split8<MODE>(br), yi = split8<MODE>(bi);
                mac6x2<false, MODE>(Or, pcr, yr, Oi, pci, yr);
                mac6x2<false, MODE>(Or, pci, yi, Oi, neg(pcr), yi);
            }
        }
        TNQS_PIN();
        lds_barrier();                                                  // every wave has read its planes
        if (more) commit();
        lds_barrier();
    }
    };
    if (SINGLE || msg == 0) run(std::integral_constant<int, 0>{}); else run(std::integral_constant<int, 1>{});
    // one partial per workgroup and message: the four waves of a message are summed through the (now free) planes
    cf* __restrict__ p1 = reinterpret_cast<cf*>(it.partial_y) + (size_t)lw * 1024;
    cf* __restrict__ p2 = reinterpret_cast<cf*>(it.partial_x) + (size_t)lw * 1024;
    v2f* const R = L;                                           // 8 blocks of 32 x 33
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        v2f v; v[0] = Or[r]; v[1] = Oi[r];
        R[w * (32 * 33) + ln * 33 + i] = v;                     // element (i, j = ln)
    }
    lds_barrier();
    if (SINGLE) {
        for (int e = tid; e < 1024; e += 512) {
            const int i = e & 31, j = e >> 5;
            float sr = 0.f, si = 0.f;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) { const v2f v = R[ww * (32 * 33) + j * 33 + i]; sr += v[0]; si += v[1]; }
            cf o; o.re = sr; o.im = si; stgc(p1 + e, o);
        }
        return;
    }
    for (int e = tid; e < 2048; e += 512) {
        const int mm = e >> 10, ee = e & 1023, i = ee & 31, j = ee >> 5;
        float sr = 0.f, si = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) { const v2f v = R[(4 * mm + ww) * (32 * 33) + j * 33 + i]; sr += v[0]; si += v[1]; }
        cf o; o.re = sr; o.im = si; stgc((mm ? p2 : p1) + ee, o);
    }
}
#undef TNQS_PIN
int x3_pair_gram2_group() { return 16; }
// one message per item (the entries of a level that have no partner message: PairGram2Item with My = partial_x = null)
void launch_x3_pair_gram1(hipStream_t s, const PairGram2Item* d_items, int nitems, int total_wgs) {
    if (total_wgs <= 0) return;
    const size_t lds = (size_t)16 * (32 * 34 + 2) * 2 * sizeof(float);
    set_max_dynamic_lds((const void*)x3_pair_gram2_kernel<0, true>, lds);
    hipLaunchKernelGGL((x3_pair_gram2_kernel<0, true>), dim3(total_wgs), dim3(512), lds, s, d_items, nitems);
    TNQS_CHECK_LAUNCH();
}
void launch_x3_pair_gram2(hipStream_t s, const PairGram2Item* d_items, int nitems, int total_wgs) {
    if (total_wgs <= 0) return;
    const size_t lds = (size_t)16 * (32 * 34 + 2) * 2 * sizeof(float);
#ifdef TNQS_EXPERIMENTS
    // timing experiments (never in the shipped build): 1 no splitting, 2 no matrix instructions, 3 no global loads / commits, 4 no LDS reads either, 5 all loads at the phase start
    static const int mode = [] { const char* e = std::getenv("TNQS_X3_MODE"); return e ? std::atoi(e) : 0; }();
    #define TNQS_X3_LAUNCH(M) { set_max_dynamic_lds((const void*)x3_pair_gram2_kernel<M>, lds); hipLaunchKernelGGL(x3_pair_gram2_kernel<M>, dim3(total_wgs), dim3(512), lds, s, d_items, nitems); }
    if (mode == 1) TNQS_X3_LAUNCH(1) else if (mode == 2) TNQS_X3_LAUNCH(2) else if (mode == 3) TNQS_X3_LAUNCH(3) else if (mode == 4) TNQS_X3_LAUNCH(4) else if (mode == 5) TNQS_X3_LAUNCH(5) else TNQS_X3_LAUNCH(0)
    #undef TNQS_X3_LAUNCH
#else
    set_max_dynamic_lds((const void*)x3_pair_gram2_kernel<0>, lds);
    hipLaunchKernelGGL(x3_pair_gram2_kernel<0>, dim3(total_wgs), dim3(512), lds, s, d_items, nitems);
#endif
    TNQS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// pair of mode products on two 32-dimensional legs in one pass (mfma_pair_kernel, kernels_mfma.hip: same items, same workgroup -> half-slice map, same
// movers and double buffering, one barrier per phase):
//   step 1  Y[iy][jx] = sum_ix S[ix][iy] Mx[ix][jx]      A = S^T from LDS: lane (ln = iy, h) reads ix = 16 n + 8 h + e, eight consecutive elements of row iy
//   step 2  S'[jx][jy] = sum_iy Y[iy][jx] My[iy][jy]      A = Y's accumulator registers 8 n + e (iy = 4 h + 8 (2 n + (e >> 2)) + (e & 3)), split on the fly
// Both matrices live in registers as split B operands (96 registers).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void x3_pair_kernel(const PairItem* __restrict__ items, int nitems) {
    constexpr int P = 34;
    constexpr int PS = 32 * P + 2;             // plane stride in complex elements
    constexpr int BUF = 8 * PS;
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
    // B operands: Mx[k = ix][j = ln] at ix + 32 ln, ix = 16 n + 8 h + e;  My[k = iy][j = ln], iy = 4 h + 16 n + 8 (e >> 2) + (e & 3)
    // (My's pieces wait in LDS behind the planes, 12 KiB: 96 registers of matrices next to two staging sets do not fit)
    P3 Mxr[2], Mxi[2];
    u4* const MyL = reinterpret_cast<u4*>(L + 16 * PS) + lane;             // [(n, re / im, piece)][lane]
    {
        const cf* __restrict__ Mx = reinterpret_cast<const cf*>(it.Mx) + 32 * ln; const cf* __restrict__ My = reinterpret_cast<const cf*>(it.My) + 32 * ln;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            float xr[8], xi[8];
#pragma unroll
            for (int e = 0; e < 8; e += 2) { const v4f v = ldg4(Mx + 16 * n + 8 * h + e); xr[e] = v[0]; xi[e] = v[1]; xr[e + 1] = v[2]; xi[e + 1] = v[3]; }
            Mxr[n] = split8(xr); Mxi[n] = split8(xi);
#pragma unroll
            for (int e = 0; e < 8; e += 2) { const v4f v = ldg4(My + 4 * h + 16 * n + 8 * (e >> 2) + (e & 3)); xr[e] = v[0]; xi[e] = v[1]; xr[e + 1] = v[2]; xi[e + 1] = v[3]; }
            const P3 yr_ = split8(xr), yi_ = split8(xi);
            if (w == 0) { MyL[(6 * n + 0) * 64] = yr_.h; MyL[(6 * n + 1) * 64] = yr_.m; MyL[(6 * n + 2) * 64] = yr_.l; MyL[(6 * n + 3) * 64] = yi_.h; MyL[(6 * n + 4) * 64] = yi_.m; MyL[(6 * n + 5) * 64] = yi_.l; }
        }
    }
    // mover: thread -> (companion pair f4 = 0..3 of the half, first segment sg0 = 0..127); segment j: ix = sg0 & 31, iy = (sg0 >> 5) + 4 j
    const int f4 = tid & 3, sg0 = tid >> 2;
    const int ix0 = sg0 & 31, iy0 = sg0 >> 5;
    const long long toff = (long long)(4 * half + f4) * g.cstr + g.sx * ix0 + g.sy * iy0, tstr = 4 * g.sy;
    v2f* const lbase = L + (2 * f4) * PS + iy0 * P + ix0;            // element (ix, iy) of a plane at [iy][ix]; iy advances by 4 per j
    // two staging register sets: the loads of phase p + 2 are issued at the start of phase p and committed at the end of phase p + 1 (a phase is ~2 us of
    // matrix work, less than the memory latency under load: with one set the commit waited for loads issued half a phase earlier)
    v4f preA[8], preB[8];
    #define TNQS_PIN() __builtin_amdgcn_sched_barrier(0x2 | 0x4)      // VALU / SALU may cross; LDS, global and matrix instructions keep their order
    auto commit = [&](int buf, const v4f (&pre)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v2f* p0 = lbase + buf * BUF + 4 * P * j;
            v2f a; a[0] = pre[j][0]; a[1] = pre[j][1]; v2f b; b[0] = pre[j][2]; b[1] = pre[j][3];
            p0[0] = a; p0[PS] = b;
        }
    };
    auto store1 = [&](int buf, long long ob, int j) {            // one 16-byte piece of a finished phase: LDS -> global
        const v2f* p0 = lbase + buf * BUF + 4 * P * j;
        const v2f a = p0[0], c = p0[PS];
        v4f v; v[0] = a[0]; v[1] = a[1]; v[2] = c[0]; v[3] = c[1];
        stg4(out + ob + tstr * j, v);
    };
    if (s_begin < s_end) {
        const long long b = x3_slice_base(g, s_begin) + toff;
#pragma unroll
        for (int j = 0; j < 8; ++j) preB[j] = ldg4(in + b + tstr * j);
        const long long b1 = x3_slice_base(g, s_begin + 1 < s_end ? s_begin + 1 : s_begin) + toff;
#pragma unroll
        for (int j = 0; j < 8; ++j) preA[j] = ldg4(in + b1 + tstr * j);
        commit(0, preB);
    }
    lds_barrier();
    // one phase: slice sl from LDS buffer `buf`; cur = the staged slice sl + 1 (committed at the end), nxt = receives slice sl + 2
    auto phase = [&](int sl, int buf, const v4f (&cur)[8], v4f (&nxt)[8]) {
        const bool more = sl + 1 < s_end, prev = sl > s_begin;
        const long long nb = x3_slice_base(g, sl + 2 < s_end ? sl + 2 : sl) + toff;      // (past the end: a slice of its own is re-read, no branch in the stream)
        const long long ob = x3_slice_base(g, prev ? sl - 1 : sl) + toff;
        v2f* Pw = L + buf * BUF + w * PS;
        // ---- step 1 ---------------------------------------------------------------------------------------------------------------
        v16f Yr, Yi;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            float ar[8], ai[8];
            const v4f* a = reinterpret_cast<const v4f*>(Pw + ln * P + 16 * n + 8 * h);
#pragma unroll
            for (int q = 0; q < 4; ++q) { const v4f v = a[q]; ar[2 * q] = v[0]; ai[2 * q] = v[1]; ar[2 * q + 1] = v[2]; ai[2 * q + 1] = v[3]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) nxt[4 * n + j] = ldg4(in + nb + tstr * (4 * n + j));
            TNQS_PIN();
            const P3 sr = split8(ar), si = split8(ai), nsi = neg(si);
            if (n == 0) mac6x2<true>(Yr, sr, Mxr[n], Yi, sr, Mxi[n]); else mac6x2<false>(Yr, sr, Mxr[n], Yi, sr, Mxi[n]);
            mac6x2<false>(Yr, nsi, Mxi[n], Yi, si, Mxr[n]);
        }
        // ---- step 2 (and the previous phase's plane on its way out) -----------------------------------------------------------------
        v16f Sr, Si;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            float yr[8], yi[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { yr[e] = Yr[8 * n + e]; yi[e] = Yi[8 * n + e]; }
            const P3 pr = split8(yr), pi = split8(yi), npi = neg(pi);
            if (prev) {
#pragma unroll
                for (int j = 0; j < 4; ++j) store1(buf ^ 1, ob, 4 * n + j);
            }
            P3 myr, myi;
            myr.h = MyL[(6 * n + 0) * 64]; myr.m = MyL[(6 * n + 1) * 64]; myr.l = MyL[(6 * n + 2) * 64]; myi.h = MyL[(6 * n + 3) * 64]; myi.m = MyL[(6 * n + 4) * 64]; myi.l = MyL[(6 * n + 5) * 64];
            TNQS_PIN();
            if (n == 0) mac6x2<true>(Sr, pr, myr, Si, pr, myi); else mac6x2<false>(Sr, pr, myr, Si, pr, myi);
            mac6x2<false>(Sr, npi, myi, Si, pi, myr);
        }
        TNQS_PIN();
        // S'[jx = (r & 3) + 8 (r >> 2) + 4 h][jy = ln] -> LDS [jy][jx] (the plane is private to this wave; its reads are complete)
#pragma unroll
        for (int r = 0; r < 16; r += 2) { const int jx = (r & 3) + 8 * (r >> 2) + 4 * h; v4f v; v[0] = Sr[r]; v[1] = Si[r]; v[2] = Sr[r + 1]; v[3] = Si[r + 1]; *reinterpret_cast<v4f*>(Pw + ln * P + jx) = v; }
        if (more) commit(buf ^ 1, cur);           // over the locations this thread stored from above
        lds_barrier();
    };
    for (int sl = s_begin; sl < s_end; sl += 2) {
        phase(sl, 0, preA, preB);
        if (sl + 1 < s_end) phase(sl + 1, 1, preB, preA);
    }
    if (s_begin < s_end) {                         // the last phase's plane
        const int buf = (s_end - 1 - s_begin) & 1;
        const long long ob = x3_slice_base(g, s_end - 1) + toff;
#pragma unroll
        for (int j = 0; j < 8; ++j) store1(buf, ob, j);
    }
    #undef TNQS_PIN
}
void launch_x3_pair(hipStream_t s, const PairItem* d_items, int nitems, int total_wgs) {
    if (total_wgs <= 0) return;
    const size_t lds = (size_t)16 * (32 * 34 + 2) * 2 * sizeof(float) + 12 * 64 * 16;
    set_max_dynamic_lds((const void*)x3_pair_kernel, lds);
    hipLaunchKernelGGL(x3_pair_kernel, dim3(total_wgs), dim3(512), lds, s, d_items, nitems);
    TNQS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// register-direct fiber GEMM with KK = D K = 64 contracted and up to 64 output columns (mfma_rowgemm_kernel<2, 2, D>, kernels_chi64.hip: same items, same
// tiles, same stores): the chi = 64 mode product (D = 1) and the chi = 32 gate epilogue with the site index folded in (D = 2).
//   out[(s', n), row] = sum_kk in[kk, row] X[kk, (s', n)],   kk = (s, k);   computed transposed, C'[nn][row] = sum_kk X^T[nn][kk] in[row][kk]
// B' operand = the tensor, straight from global memory into the registers that are split for the matrix cores: lane (ln = row, h) supplies the eight
// k-slots kk = 16 n + 8 h + e of instruction n (D = 2: four 16-byte loads -- both site components of four k -- D = 1: eight 8-byte loads); A' operand =
// X^T, split ONCE per workgroup into LDS in operand order (48 KiB: [n][nb][re / im][piece][lane] x 16 bytes, one conflict-free 16-byte read per piece).
// Two half tiles of operand registers as in the f32 kernel: a half is refilled with the next tile's data as soon as its instructions are issued.
// ------------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256, 2) void x3_rowgemm64_kernel(const FiberItem* __restrict__ items, int nitems, double* __restrict__ norm_partials) {
    constexpr int KK = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u4* const Xl = reinterpret_cast<u4*>(smem);              // [(n, nb)][re h m l, im h m l][lane]
    __shared__ double sh_red[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, ln = lane & 31, h = lane >> 5;
    int lo = 0, hi = nitems - 1;
    const int gw = blockIdx.x;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (items[mid].tile_begin <= gw) lo = mid; else hi = mid - 1; }
    const FiberItem it = items[lo];
    const int K = it.K, No = it.No, NN = it.Do * No;
    const long long PA = it.PA;
    const int ntiles = it.nta * it.ntb;
    const int t_begin = (gw - it.tile_begin) * it.tpw, t_end = min(ntiles, t_begin + it.tpw);
    const cf* __restrict__ in = reinterpret_cast<const cf*>(it.in);
    const cf* __restrict__ X = reinterpret_cast<const cf*>(it.X);
    cf* __restrict__ out = reinterpret_cast<cf*>(it.out);
    for (int e = tid; e < 4 * 2 * 64; e += 256) {            // (n, nb, lane): eight consecutive kk of column nn
        const int l = e & 63, nb = (e >> 6) & 1, n = e >> 7;
        const int nn = 32 * nb + (l & 31), kk0 = 16 * n + 8 * (l >> 5);
        float xr[8], xi[8];
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
            v4f v = {0.f, 0.f, 0.f, 0.f};
            if (nn < NN) v = ldg4(X + kk0 + q + (size_t)KK * nn);
            xr[q] = v[0]; xi[q] = v[1]; xr[q + 1] = v[2]; xi[q + 1] = v[3];
        }
        const P3 pr = split8(xr), pi = split8(xi);
        u4* d = Xl + (size_t)((2 * n + nb) * 6) * 64 + l;
        d[0] = pr.h; d[64] = pr.m; d[128] = pr.l; d[192] = pi.h; d[256] = pi.m; d[320] = pi.l;
    }
    __syncthreads();                                           // the only workgroup barrier
    float hb[2][32];                                           // two half tiles: instructions n = 2 hf, 2 hf + 1 -> (re, im) of eight slots each
    const bool rows_b = PA < 32;
    const int RB = rows_b ? 32 / (int)PA : 1;
    const long long kst = (long long)D * PA;                                                   // stride of the contracted index (elements)
    const long long lane_in = rows_b ? (long long)D * (ln % (int)PA) + kst * K * (ln / (int)PA) : (long long)D * ln;
    const long long lane_out = rows_b ? (long long)D * (ln % (int)PA) + kst * No * (ln / (int)PA) : (long long)D * ln;
    auto base_in = [&](int t) { return rows_b ? kst * K * RB * t : (long long)D * 32 * (t % it.nta) + kst * K * (t / it.nta); };
    auto base_out = [&](int t) { return rows_b ? kst * No * RB * t : (long long)D * 32 * (t % it.nta) + kst * No * (t / it.nta); };
    // one 32-bit per-lane byte offset for every load of the kernel (row and lane half), everything else of an address is wave-uniform: the loads take the
    // scalar-base form instead of one 64-bit address register pair each (sixteen of them live per half tile spilled the D = 1 variant at two workgroups per CU)
    const unsigned voff = (unsigned)((lane_in + kst * (D == 1 ? 8 : 4) * h) * (long long)sizeof(cf));
    auto issue_half = [&](int t, int hf) {
        const cf* pu = in + base_in(t);
#pragma unroll
        for (int nn2 = 0; nn2 < 2; ++nn2) {
            const int n = 2 * hf + nn2;
            if (D == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const v2f v = ldg2(reinterpret_cast<const char*>(pu + kst * (16 * n + e)) + voff); hb[hf][16 * nn2 + e] = v[0]; hb[hf][16 * nn2 + 8 + e] = v[1]; }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const v4f v = ldg4(reinterpret_cast<const char*>(pu + kst * (8 * n + j)) + voff);
                                              hb[hf][16 * nn2 + 2 * j] = v[0]; hb[hf][16 * nn2 + 8 + 2 * j] = v[1]; hb[hf][16 * nn2 + 2 * j + 1] = v[2]; hb[hf][16 * nn2 + 8 + 2 * j + 1] = v[3]; }
            }
        }
    };
    double nrm = 0;
    int t = t_begin + w;
    if (t < t_end) { issue_half(t, 0); issue_half(t, 1); }
    for (; t < t_end; t += 4) {
        v16f Cr[2], Ci[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
            for (int nn2 = 0; nn2 < 2; ++nn2) {
                const int n = 2 * hf + nn2;
                float br[8], bi[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { br[e] = hb[hf][16 * nn2 + e]; bi[e] = hb[hf][16 * nn2 + 8 + e]; }
                const P3 pbr = split8(br), pbi = split8(bi), nbi = neg(pbi);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const u4* a = Xl + (size_t)((2 * n + nb) * 6) * 64 + lane;
                    P3 xr, xi; xr.h = a[0]; xr.m = a[64]; xr.l = a[128]; xi.h = a[192]; xi.m = a[256]; xi.l = a[320];
                    if (n == 0) mac6x2<true>(Cr[nb], xr, pbr, Ci[nb], xr, pbi); else mac6x2<false>(Cr[nb], xr, pbr, Ci[nb], xr, pbi);
                    mac6x2<false>(Cr[nb], xi, nbi, Ci[nb], xi, pbr);
                }
            }
            __builtin_amdgcn_sched_barrier(0);                                   // the refill must not be hoisted above the instructions that read the half
            if (t + 4 < t_end) issue_half(t + 4, hf);
        }
        float nf = 0.f;
        cf* p = out + base_out(t) + lane_out;
        if (D == 1) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = 32 * nb + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (n < No) { cf v; v.re = Cr[nb][r]; v.im = Ci[nb][r]; stgc(p + PA * n, v); nf += v.re * v.re + v.im * v.im; }
                }
        } else {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int nn = 32 * nb + (r & 3) + 8 * (r >> 2) + 4 * h;  // even: s' = 0 of n = nn / 2; register r + 1 is s' = 1
                    const int n = nn >> 1;
                    if (n < No) {
                        v4f v = {Cr[nb][r], Ci[nb][r], Cr[nb][r + 1], Ci[nb][r + 1]};
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
void launch_x3_rowgemm64(hipStream_t s, const FiberItem* d_items, int nitems, int total_wgs, int D, double* d_norm_partials) {
    if (total_wgs <= 0) return;
    const size_t lds = (size_t)4 * 2 * 6 * 64 * 16;
    if (D == 1) { set_max_dynamic_lds((const void*)x3_rowgemm64_kernel<1>, lds); hipLaunchKernelGGL(x3_rowgemm64_kernel<1>, dim3(total_wgs), dim3(256), lds, s, d_items, nitems, d_norm_partials); }
    else { set_max_dynamic_lds((const void*)x3_rowgemm64_kernel<2>, lds); hipLaunchKernelGGL(x3_rowgemm64_kernel<2>, dim3(total_wgs), dim3(256), lds, s, d_items, nitems, d_norm_partials); }
    TNQS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// Gram with f32 accumulation, KK = D K <= 64 (mfma_gram64_kernel, kernels_chi64.hip: same items, tiles, loads, LDS image and partials): the BP message Gram
// at chi = 64 and at chi = 32 with the site index kept.  The sixteen tile rows a wave takes per half are exactly one v_mfma_f32_32x32x16_bf16 (lane half h:
// eight CONSECUTIVE rows of its column, already contiguous in the [kk][row] image): 96 bf16 instructions per tile and wave where the f32 kernel issued 96 of
// twice the length, and 552 vector instructions of splitting.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void x3_gram64_kernel(const GramItem* __restrict__ items, int nitems) {
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
    v16f Cr[2], Ci[2];                                           // out = X conj(Y):  re = Xr Yr + Xi Yi,  im = Xi Yr + Xr (-Yi)
#pragma unroll
    for (int r = 0; r < 16; ++r) { Cr[0][r] = 0.f; Cr[1][r] = 0.f; Ci[0][r] = 0.f; Ci[1][r] = 0.f; }
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
            // one matrix instruction covers the sixteen rows of this half (lane half h: eight of them): operands split on the fly
            const P3 pxr = split8(xr), pxi = split8(xi);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const P3 pyr = split8(yr[b]), pyi = split8(yi[b]);
                mac6x2<false>(Cr[b], pxr, pyr, Ci[b], pxi, pyr);
                mac6x2<false>(Cr[b], pxi, pyi, Ci[b], pxr, neg(pyi));
            }
        }
    }
    // one partial per chunk: the two row halves (waves w, w + 2) are summed through the free tile buffers
    lds_barrier();
    v2f* const R = reinterpret_cast<v2f*>(smem);                // [rh][j][i], pitch 65: 2 * 64 * 65 * 8 B = 66.5 KB
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h, j = 32 * b + ln;
            v2f v = {Cr[b][r], Ci[b][r]};
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
bool launch_x3_gram64(hipStream_t s, const GramItem* d_items, int nitems, int total_chunks, int KKmax) {
    if (KKmax > 64) return false;
    if (total_chunks <= 0) return true;
    const size_t lds = (size_t)4 * 64 * 68 * sizeof(float);
    set_max_dynamic_lds((const void*)x3_gram64_kernel, lds);
    hipLaunchKernelGGL(x3_gram64_kernel, dim3(total_chunks), dim3(256), lds, s, d_items, nitems);
    TNQS_CHECK_LAUNCH();
    return true;
}

}  // namespace tnqs
