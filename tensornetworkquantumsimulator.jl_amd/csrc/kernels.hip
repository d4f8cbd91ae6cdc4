// kernels.hip -- generic (any dims, c64/c128) HIP kernels of the BP-gauged gate-application path, gfx950.
// These are the always-correct fiber-tile kernels; the MFMA fast paths for the hot shapes live in
// kernels_mfma.hip and are validated against these.
//
// Reference call sites replaced (paths relative to the reference repo):
//   fiber_gemm  : ITensors `contract`/`apply` in src/Apply/simple_update.jl:27,43-44,62-64 and the message
//                 absorptions of src/MessagePassing/abstractbeliefpropagationcache.jl:180
//   gram        : final contraction with dag(prime(psi)) (abstract...:180) and the R-factor Gram of the QR
//                 step (simple_update.jl:47-48, replaced by an f64 Gram + eigen factorisation)
//   msg_finalize: abstract...:182-187 (m / sum(m)) + message_diff beliefpropagationcache.jl:17-21
//   jacobi      : `eigen` (src/utils.jl:29-35,94-108) and `factorize_svd` (simple_update.jl:53-59)
//   gate_theta / gate_finish : simple_update.jl:51-59 + NDTensors truncate! rule, apply_gates.jl:126-135
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <stdexcept>
#include <string>
#define TNQS_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) throw std::runtime_error(std::string("HIP kernel launch failed (") + __func__ + "): " + hipGetErrorString(e_)); } while (0)
#include "kernels.hpp"
#include "launch_util.hpp"

namespace tnqs {

template <class T> struct alignas(2 * sizeof(T)) cx { T re, im; };

template <class T> __device__ __forceinline__ cx<T> cmake(T a, T b) { cx<T> r; r.re = a; r.im = b; return r; }
template <class T> __device__ __forceinline__ void cfma(cx<T>& acc, const cx<T>& a, const cx<T>& b) {
    acc.re = fma(a.re, b.re, acc.re); acc.re = fma(-a.im, b.im, acc.re);
    acc.im = fma(a.re, b.im, acc.im); acc.im = fma(a.im, b.re, acc.im);
}
// acc += a * conj(b)
template <class T> __device__ __forceinline__ void cfma_conj(cx<T>& acc, const cx<T>& a, const cx<T>& b) {
    acc.re = fma(a.re, b.re, acc.re); acc.re = fma(a.im, b.im, acc.re);
    acc.im = fma(a.im, b.re, acc.im); acc.im = fma(-a.re, b.im, acc.im);
}
template <class T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <class T> __device__ __forceinline__ T eps_of();
template <> __device__ __forceinline__ float eps_of<float>() { return FLT_EPSILON; }
template <> __device__ __forceinline__ double eps_of<double>() { return DBL_EPSILON; }

// block-wide sum of a double (blockDim.x <= 1024); result valid in every thread
__device__ __forceinline__ double block_sum(double v, double* sh /* >= 17 doubles */) {
    v = wave_sum(v);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0; for (int i = 0; i < nw; ++i) t += sh[i]; sh[16] = t; }
    __syncthreads();
    return sh[16];
}

// ------------------------------------------------------------------------------------------------------------
// fiber_gemm
// ------------------------------------------------------------------------------------------------------------
template <class T, int NB>
__global__ __launch_bounds__(256) void fiber_gemm_kernel(const FiberItem* __restrict__ items, int nitems, int TR,
                                                         double* __restrict__ norm_partials) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cx<T>* tile = reinterpret_cast<cx<T>*>(smem);
    __shared__ double sh_red[17];
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
    const size_t PA = it.PA;
    const cx<T>* __restrict__ in = reinterpret_cast<const cx<T>*>(it.in);
    const int ntile_el = D * TA * K * TB;
    for (int e = tid; e < ntile_el; e += 256) {
        int s = e % D; int r1 = e / D; int al = r1 % TA; int r2 = r1 / TA; int k = r2 % K; int bl = r2 / K;
        cx<T> v = cmake<T>(0, 0);
        if (al < na && bl < nb) v = in[s + D * ((size_t)(a0 + al) + PA * ((size_t)k + (size_t)K * (b0 + bl)))];
        tile[(s + D * k) * TR + (al + TA * bl)] = v;
    }
    __syncthreads();
    const int Do = it.Do, No = it.No, NN = Do * No;
    const int r = tid % TR, g = tid / TR, G = 256 / TR;
    const int al = r % TA, bl = r / TA;
    const bool valid = (r < TA * TB) && al < na && bl < nb;
    const cx<T>* __restrict__ X = reinterpret_cast<const cx<T>*>(it.X);
    cx<T>* __restrict__ out = reinterpret_cast<cx<T>*>(it.out);
    double nrm = 0;
    for (int nn0 = g * NB; nn0 < NN; nn0 += G * NB) {
        cx<T> acc[NB];
        int col[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) { acc[j] = cmake<T>(0, 0); col[j] = min(nn0 + j, NN - 1) * KK; }
        for (int kk = 0; kk < KK; ++kk) {
            const cx<T> a = tile[kk * TR + r];
#pragma unroll
            for (int j = 0; j < NB; ++j) cfma(acc[j], a, X[col[j] + kk]);
        }
        if (valid) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                int nn = nn0 + j;
                if (nn < NN) {
                    int sp = nn % Do, n = nn / Do;
                    out[sp + Do * ((size_t)(a0 + al) + PA * ((size_t)n + (size_t)No * (b0 + bl)))] = acc[j];
                    nrm += (double)acc[j].re * acc[j].re + (double)acc[j].im * acc[j].im;
                }
            }
        }
    }
    if (it.want_norm) {   // uniform per block
        double t = block_sum(nrm, sh_red);
        if (tid == 0) norm_partials[gt] = t;
    }
}

template <class T>
void launch_fiber_gemm(hipStream_t s, const FiberItem* d_items, int nitems, int total_tiles, int TR, int KKmax,
                       double* d_norm_partials) {
    if (total_tiles <= 0) return;
    size_t lds = (size_t)KKmax * TR * sizeof(cx<T>);
    hipLaunchKernelGGL((fiber_gemm_kernel<T, 8>), dim3(total_tiles), dim3(256), lds, s, d_items, nitems, TR, d_norm_partials); TNQS_CHECK_LAUNCH();
}
template void launch_fiber_gemm<float>(hipStream_t, const FiberItem*, int, int, int, int, double*);
template void launch_fiber_gemm<double>(hipStream_t, const FiberItem*, int, int, int, int, double*);

// ------------------------------------------------------------------------------------------------------------
// gram
// ------------------------------------------------------------------------------------------------------------
template <class T, class Acc, int MAXB>
__global__ __launch_bounds__(256) void gram_kernel(const GramItem* __restrict__ items, int nitems, int TR) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    int lo = 0, hi = nitems - 1;
    const int gc = blockIdx.x;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (items[mid].chunk_begin <= gc) lo = mid; else hi = mid - 1; }
    const GramItem it = items[lo];
    const int lc = gc - it.chunk_begin;
    const int D = it.D, K = it.K, TA = it.TA, TB = it.TB, KK = D * K;
    const int KKp = KK + 1;                      // row pitch of the LDS tiles [row][kk]
    const size_t PA = it.PA;
    const bool same = (it.X == it.Y);
    cx<T>* Xt = reinterpret_cast<cx<T>*>(smem);
    cx<T>* Yt = same ? Xt : Xt + (size_t)TR * KKp;
    const cx<T>* __restrict__ Xg = reinterpret_cast<const cx<T>*>(it.X);
    const cx<T>* __restrict__ Yg = reinterpret_cast<const cx<T>*>(it.Y);
    const int KB = (KK + 1) >> 1;                // 2x2 output blocks per dimension
    const int nblk = KB * KB;
    const int ntiles = it.nta * it.ntb;
    const int t_begin = lc * it.tiles_per_chunk;
    const int t_end = min(ntiles, t_begin + it.tiles_per_chunk);
    cx<Acc>* __restrict__ part = reinterpret_cast<cx<Acc>*>(it.partial) + (size_t)lc * KK * KK;
    const int ntile_el = D * TA * K * TB;
    for (int pass0 = 0; pass0 < nblk; pass0 += 256 * MAXB) {
        cx<Acc> acc[MAXB][4];
        int bi[MAXB], bj[MAXB];
#pragma unroll
        for (int j = 0; j < MAXB; ++j) {
            int q = pass0 + tid + 256 * j;
            int qq = min(q, nblk - 1);
            bi[j] = qq % KB; bj[j] = qq / KB;
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[j][c] = cmake<Acc>(0, 0);
        }
        for (int t = t_begin; t < t_end; ++t) {
            const int ta = t % it.nta, tb = t / it.nta;
            const int a0 = ta * TA, b0 = tb * TB;
            const int na = min(TA, it.PA - a0), nb = min(TB, it.PB - b0);
            __syncthreads();
            for (int e = tid; e < ntile_el; e += 256) {
                int s = e % D; int r1 = e / D; int al = r1 % TA; int r2 = r1 / TA; int k = r2 % K; int bl = r2 / K;
                cx<T> vx = cmake<T>(0, 0), vy = cmake<T>(0, 0);
                if (al < na && bl < nb) {
                    size_t off = s + D * ((size_t)(a0 + al) + PA * ((size_t)k + (size_t)K * (b0 + bl)));
                    vx = Xg[off];
                    if (!same) vy = Yg[off];
                }
                int li = (al + TA * bl) * KKp + (s + D * k);
                Xt[li] = vx;
                if (!same) Yt[li] = vy;
            }
            // zero the pad column so clamped reads of index KK (odd KK) are harmless
            for (int e = tid; e < TA * TB; e += 256) { Xt[e * KKp + KK] = cmake<T>(0, 0); if (!same) Yt[e * KKp + KK] = cmake<T>(0, 0); }
            __syncthreads();
            const int nrow = TA * TB;
            for (int rr = 0; rr < nrow; ++rr) {
                const cx<T>* xr = Xt + rr * KKp;
                const cx<T>* yr = Yt + rr * KKp;
#pragma unroll
                for (int j = 0; j < MAXB; ++j) {
                    cx<T> x0 = xr[2 * bi[j]], x1 = xr[2 * bi[j] + 1];
                    cx<T> y0 = yr[2 * bj[j]], y1 = yr[2 * bj[j] + 1];
                    cx<Acc> X0 = cmake<Acc>(x0.re, x0.im), X1 = cmake<Acc>(x1.re, x1.im);
                    cx<Acc> Y0 = cmake<Acc>(y0.re, y0.im), Y1 = cmake<Acc>(y1.re, y1.im);
                    cfma_conj(acc[j][0], X0, Y0); cfma_conj(acc[j][1], X1, Y0);
                    cfma_conj(acc[j][2], X0, Y1); cfma_conj(acc[j][3], X1, Y1);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < MAXB; ++j) {
            int q = pass0 + tid + 256 * j;
            if (q < nblk) {
                int i0 = 2 * bi[j], j0 = 2 * bj[j];
                part[i0 + (size_t)KK * j0] = acc[j][0];
                if (i0 + 1 < KK) part[i0 + 1 + (size_t)KK * j0] = acc[j][1];
                if (j0 + 1 < KK) {
                    part[i0 + (size_t)KK * (j0 + 1)] = acc[j][2];
                    if (i0 + 1 < KK) part[i0 + 1 + (size_t)KK * (j0 + 1)] = acc[j][3];
                }
            }
        }
    }
}

template <class T, class Acc>
void launch_gram(hipStream_t s, const GramItem* d_items, int nitems, int total_chunks, int TR, int KKmax) {
    if (total_chunks <= 0) return;
    size_t lds = 2 * (size_t)TR * (KKmax + 1) * sizeof(cx<T>);
    hipLaunchKernelGGL((gram_kernel<T, Acc, 4>), dim3(total_chunks), dim3(256), lds, s, d_items, nitems, TR); TNQS_CHECK_LAUNCH();
}
template void launch_gram<float, float>(hipStream_t, const GramItem*, int, int, int, int);
template void launch_gram<float, double>(hipStream_t, const GramItem*, int, int, int, int);
template void launch_gram<double, double>(hipStream_t, const GramItem*, int, int, int, int);

// ------------------------------------------------------------------------------------------------------------
// reduce partials
// ------------------------------------------------------------------------------------------------------------
template <class Acc, class Out>
__global__ __launch_bounds__(256) void reduce_kernel(const ReduceItem* __restrict__ items, int nitems, int total) {
    int ge = blockIdx.x * 256 + threadIdx.x;
    if (ge >= total) return;
    int lo = 0, hi = nitems - 1;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (items[mid].elem_begin <= ge) lo = mid; else hi = mid - 1; }
    const ReduceItem it = items[lo];
    int e = ge - it.elem_begin;
    const cx<Acc>* p = reinterpret_cast<const cx<Acc>*>(it.partial);
    Acc re = 0, im = 0;
    for (int c = 0; c < it.nchunks; ++c) { cx<Acc> v = p[(size_t)c * it.n2 + e]; re += v.re; im += v.im; }
    if (it.conj) im = -im;
    reinterpret_cast<cx<Out>*>(it.out)[e] = cmake<Out>((Out)re, (Out)im);
}
template <class Acc, class Out>
void launch_reduce(hipStream_t s, const ReduceItem* d_items, int nitems, int total_elems) {
    if (total_elems <= 0) return;
    hipLaunchKernelGGL((reduce_kernel<Acc, Out>), dim3((total_elems + 255) / 256), dim3(256), 0, s, d_items, nitems, total_elems); TNQS_CHECK_LAUNCH();
}
template void launch_reduce<double, double>(hipStream_t, const ReduceItem*, int, int);
template void launch_reduce<float, float>(hipStream_t, const ReduceItem*, int, int);
template void launch_reduce<double, float>(hipStream_t, const ReduceItem*, int, int);

// ------------------------------------------------------------------------------------------------------------
// BP message epilogue: reduce partials, normalise by the sum of all elements, message_diff
// ------------------------------------------------------------------------------------------------------------
// several block-wide sums with ONE pair of barriers (blockDim.x <= 1024); results valid in every thread
template <int N>
__device__ __forceinline__ void block_sum_n(double (&v)[N], double* sh /* >= 17 N doubles */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = wave_sum(v[k]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) sh[N * w + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < N) { double t = 0; for (int i = 0; i < nw; ++i) t += sh[N * i + threadIdx.x]; sh[16 * N + threadIdx.x] = t; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = sh[16 * N + k];
}
// 1024 threads per message (round 5): with 256 a chi = 32 message was four elements per thread x 16 partials in dependent groups of eight loads, then six
// block-wide sums of three barriers each -- 50 us per launch on the critical path of every BP level, whatever the lattice size; now one element per thread and
// two reductions (element sum; the four sums of message_diff together)
template <class T>
__global__ __launch_bounds__(1024) void msg_finalize_kernel(const MsgFinalItem* __restrict__ items) {
    __shared__ double sh[17 * 4];
    const MsgFinalItem it = items[blockIdx.x];
    const int n2 = it.chi * it.chi;
    const int NT = blockDim.x;
    const cx<T>* p = reinterpret_cast<const cx<T>*>(it.partial);
    cx<T>* out = reinterpret_cast<cx<T>*>(it.new_msg);
    const cx<T>* old = reinterpret_cast<const cx<T>*>(it.old_msg);
    // pass 1: reduce chunks (fixed order) into new_msg, accumulate the element sum
    double s2[2] = {0, 0};
    for (int e = threadIdx.x; e < n2; e += NT) {
        // eight independent partial sums (fixed order): the loads of a thread do not depend on each other, so eight are in flight at a
        // time instead of one -- a message with thousands of partials took a millisecond here
        T pr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pi[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int c = 0;
        // (32 loads in flight, added in the same order as the groups of eight below: a level of the forest-cover order has a few messages of a few hundred
        //  partials each -- many short workgroups per site -- and their 32 dependent rounds of eight loads were 25 us of a 60 us level)
        for (; c + 32 <= it.nchunks; c += 32) {
            cx<T> v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = p[(size_t)(c + u) * n2 + e];
#pragma unroll
            for (int u = 0; u < 32; ++u) { pr[u & 7] += v[u].re; pi[u & 7] += v[u].im; }
        }
        for (; c + 8 <= it.nchunks; c += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { cx<T> v = p[(size_t)(c + u) * n2 + e]; pr[u] += v.re; pi[u] += v.im; }
        }
        for (; c < it.nchunks; ++c) { cx<T> v = p[(size_t)c * n2 + e]; pr[c & 7] += v.re; pi[c & 7] += v.im; }
        const T re = ((pr[0] + pr[1]) + (pr[2] + pr[3])) + ((pr[4] + pr[5]) + (pr[6] + pr[7]));
        const T im = ((pi[0] + pi[1]) + (pi[2] + pi[3])) + ((pi[4] + pi[5]) + (pi[6] + pi[7]));
        out[e] = cmake<T>(re, im);
        s2[0] += re; s2[1] += im;
    }
    block_sum_n<2>(s2, sh);
    const double sre = s2[0], sim = s2[1];
    // m / sum(m)   (abstractbeliefpropagationcache.jl:182-187; skipped when the sum is exactly zero)
    double ire = 1, iim = 0;
    if (it.normalize && (sre != 0 || sim != 0)) { double d = sre * sre + sim * sim; ire = sre / d; iim = -sim / d; }
    double d4[4] = {0, 0, 0, 0};       // Re, Im of dot(new, old), |new|^2, |old|^2
    for (int e = threadIdx.x; e < n2; e += NT) {
        cx<T> v = out[e];              // (written by this thread above)
        double re = v.re * ire - v.im * iim, im = v.re * iim + v.im * ire;
        cx<T> w = cmake<T>((T)re, (T)im);
        out[e] = w;
        double ore, oim;
        if (old) { ore = old[e].re; oim = old[e].im; } else { ore = (e % it.chi == e / it.chi) ? 1.0 : 0.0; oim = 0; }
        // dot(a, b) = sum conj(a) b with a = new, b = old  (beliefpropagationcache.jl:17-21)
        d4[0] += (double)w.re * ore + (double)w.im * oim;
        d4[1] += (double)w.re * oim - (double)w.im * ore;
        d4[2] += (double)w.re * w.re + (double)w.im * w.im;
        d4[3] += ore * ore + oim * oim;
    }
    block_sum_n<4>(d4, sh);
    if (threadIdx.x == 0 && it.diff_out) {
        double f = (d4[0] * d4[0] + d4[1] * d4[1]) / (d4[2] * d4[3]);
        *it.diff_out = 1.0 - f;
    }
}
template <class T> void launch_msg_finalize(hipStream_t s, const MsgFinalItem* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL((msg_finalize_kernel<T>), dim3(nitems), dim3(1024), 0, s, d_items); TNQS_CHECK_LAUNCH();
}
template void launch_msg_finalize<float>(hipStream_t, const MsgFinalItem*, int);
template void launch_msg_finalize<double>(hipStream_t, const MsgFinalItem*, int);

// ------------------------------------------------------------------------------------------------------------
// BP message of a SMALL site in one kernel (round 5): a site tensor of at most 8192 elements (64 KiB: heavy-hex chi = 16, every boundary site of a chi <= 16
// lattice) lives in LDS for the whole message -- absorb the incoming messages leg by leg (ping-pong between two LDS copies), then contract with conj(psi) over
// everything but the outgoing leg.  One workgroup per (site, outgoing message); the result is the raw message (one partial for msg_finalize).  The generic route
// streamed such a tensor through one fiber-GEMM launch per leg plus a Gram launch, each with its descriptor copy: 16 launch groups of ~90 us per heavy-hex layer at
// 0.09 TB/s (updated_message, abstractbeliefpropagationcache.jl:162-190)
// ------------------------------------------------------------------------------------------------------------
// The same message with EVERY leg 16-dimensional (heavy-hex at chi = 16: the shape the kernel exists for) on v_mfma_f32_16x16x4_f32 -- the scalar form below reads
// two LDS operands per multiply-add and is bound by the LDS bandwidth of its CU (55 us per degree-3 message); here an operand is read once per 16 multiply-adds.
// Lane l = (c = l & 15, g = l >> 4) supplies A[i = c][k = g] and B[k = g][j = c] and receives C[row = 4 g + r][col = c]; instruction t of a product takes
// contraction index 4 g + t (kernels_plane.hip).  Four real products per complex one.
//   absorb leg k (stride P):  out[fiber, qo] = sum_q cur[fiber, q] M[q, qo], computed transposed: A = M^T from registers, B = 16 fibers x 16 q from LDS,
//                             C[qo][fiber] stored along the fibers (contiguous); a wave takes tiles of 16 fibers
//   Gram over all but leg jo: out[i, j] = sum_rest cur[rest, i] conj psi[rest, j]: A, B = 4 rest values x 16 from the two LDS copies per instruction; the waves split
//                             the rest index and their 16 x 16 partial sums meet in LDS
typedef float v4f_ss __attribute__((ext_vector_type(4)));
template <int NT>
__device__ __forceinline__ void bp_small_site_mfma16(const SmallMsgItem& it, int E, char* smem, size_t smem_bytes) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = NT >> 6, c = lane & 15, g = lane >> 4;
    cx<float>* cur = reinterpret_cast<cx<float>*>(smem);
    cx<float>* nxt = cur + E;
    // psi: 16-byte loads, all in flight at once, kept in registers for the second copy (E <= 8192: at most four per thread); the message matrices of all legs
    // are fetched behind them (A[i = qo = c][k = q = 4 g + t] = M[q, qo]: four consecutive numbers per lane) -- one exposed memory latency per message
    typedef float v4f_ld __attribute__((ext_vector_type(4)));
    const v4f_ld* p4 = reinterpret_cast<const v4f_ld*>(it.psi);
    const int n4 = E >> 1;
    v4f_ld keep[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (tid + NT * u < n4) keep[u] = p4[tid + NT * u];
    float mr[8][4], mi[8][4];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (k < it.z && k != it.jo && it.M[k]) {
            const cx<float>* Mg = reinterpret_cast<const cx<float>*>(it.M[k]);
#pragma unroll
            for (int t = 0; t < 4; ++t) { const cx<float> v = Mg[(4 * g + t) + 16 * c]; mr[k][t] = v.re; mi[k][t] = v.im; }
        }
#pragma unroll
    for (int u = 0; u < 4; ++u) if (tid + NT * u < n4) reinterpret_cast<v4f_ld*>(cur)[tid + NT * u] = keep[u];
    __syncthreads();
    const int ntile = E >> 8;                                         // tiles of 16 fibers of 16
    int P = it.d;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (k >= it.z) break;
        if (k != it.jo && it.M[k]) {
            for (int T = w; T < ntile; T += nw) {
                const int F = 16 * T + c, pre = F % P, post = F / P;
                const size_t base = pre + (size_t)P * 16 * post;
                v4f_ss Cr = {0.f, 0.f, 0.f, 0.f}, Ci = Cr;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const cx<float> x = cur[base + (size_t)P * (4 * g + t)];                                               // B[k = q][j = fiber c]
                    Cr = __builtin_amdgcn_mfma_f32_16x16x4f32(mr[k][t], x.re, Cr, 0, 0, 0);
                    Cr = __builtin_amdgcn_mfma_f32_16x16x4f32(-mi[k][t], x.im, Cr, 0, 0, 0);
                    Ci = __builtin_amdgcn_mfma_f32_16x16x4f32(mr[k][t], x.im, Ci, 0, 0, 0);
                    Ci = __builtin_amdgcn_mfma_f32_16x16x4f32(mi[k][t], x.re, Ci, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) nxt[base + (size_t)P * (4 * g + r)] = cmake<float>(Cr[r], Ci[r]);              // C[row = qo = 4 g + r][col = fiber c]
            }
            __syncthreads();
            cx<float>* t_ = cur; cur = nxt; nxt = t_;
        }
        P *= 16;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) if (tid + NT * u < n4) reinterpret_cast<v4f_ld*>(nxt)[tid + NT * u] = keep[u];
    __syncthreads();
    int Po = it.d; for (int k = 0; k < it.jo; ++k) Po *= 16;
    const int nstep = E >> 6;                                         // instructions' worth of the rest index: 4 rest values each
    int na = nw < nstep ? nw : nstep;                                 // waves that take part; their 2 KiB partials must fit the kernel's LDS
    if ((size_t)na * 2048 > smem_bytes) na = (int)(smem_bytes / 2048);
    v4f_ss Or = {0.f, 0.f, 0.f, 0.f}, Oi = Or;
    if (w < na)
        for (int st = w; st < nstep; st += na) {
            const int R = 4 * st + g, pre = R % Po, post = R / Po;
            const size_t base = pre + (size_t)Po * 16 * post + (size_t)Po * c;
            const cx<float> a = cur[base], b = nxt[base];             // A[i = c][k = rest], B[k = rest][j = c]
            Or = __builtin_amdgcn_mfma_f32_16x16x4f32(a.re, b.re, Or, 0, 0, 0);
            Or = __builtin_amdgcn_mfma_f32_16x16x4f32(a.im, b.im, Or, 0, 0, 0);
            Oi = __builtin_amdgcn_mfma_f32_16x16x4f32(a.im, b.re, Oi, 0, 0, 0);
            Oi = __builtin_amdgcn_mfma_f32_16x16x4f32(-a.re, b.im, Oi, 0, 0, 0);
        }
    __syncthreads();                                                  // both copies have been consumed: the partial sums go over them
    cx<float>* part = reinterpret_cast<cx<float>*>(smem);
    if (w < na) {
#pragma unroll
        for (int r = 0; r < 4; ++r) part[256 * w + (4 * g + r) + 16 * c] = cmake<float>(Or[r], Oi[r]);                     // out[i + 16 j], i = 4 g + r, j = c
    }
    __syncthreads();
    static_assert(NT >= 256, "one thread per element of the 16 x 16 message");
    cx<float> val = cmake<float>(0.f, 0.f);
    if (tid < 256) {
        float sr = 0.f, si = 0.f;
        for (int u = 0; u < na; ++u) { const cx<float> v = part[256 * u + tid]; sr += v.re; si += v.im; }
        val = cmake<float>(sr, si);
    }
    if (!it.new_msg) { if (tid < 256) reinterpret_cast<cx<float>*>(it.out)[tid] = val; return; }
    // ---- the epilogue of msg_finalize_kernel on the message this workgroup holds: m / sum(m) (abstractbeliefpropagationcache.jl:182-187; skipped when the sum is
    // exactly zero), message_diff against the previous message (beliefpropagationcache.jl:17-21) -------------------------------------------------------------
    __shared__ double sh[17 * 4];
    double s2[2] = {(double)val.re, (double)val.im};
    block_sum_n<2>(s2, sh);
    const double sre = s2[0], sim = s2[1];
    double ire = 1, iim = 0;
    if (it.normalize && (sre != 0 || sim != 0)) { const double d = sre * sre + sim * sim; ire = sre / d; iim = -sim / d; }
    double d4[4] = {0, 0, 0, 0};       // Re, Im of dot(new, old), |new|^2, |old|^2
    if (tid < 256) {
        const double re = val.re * ire - val.im * iim, im = val.re * iim + val.im * ire;
        const cx<float> wv = cmake<float>((float)re, (float)im);
        reinterpret_cast<cx<float>*>(it.new_msg)[tid] = wv;
        const cx<float>* old = reinterpret_cast<const cx<float>*>(it.old_msg);
        double ore, oim;
        if (old) { ore = old[tid].re; oim = old[tid].im; } else { ore = ((tid & 15) == (tid >> 4)) ? 1.0 : 0.0; oim = 0; }
        d4[0] = (double)wv.re * ore + (double)wv.im * oim;
        d4[1] = (double)wv.re * oim - (double)wv.im * ore;
        d4[2] = (double)wv.re * wv.re + (double)wv.im * wv.im;
        d4[3] = ore * ore + oim * oim;
    }
    block_sum_n<4>(d4, sh);
    if (tid == 0 && it.diff_out) *it.diff_out = 1.0 - (d4[0] * d4[0] + d4[1] * d4[1]) / (d4[2] * d4[3]);
}
template <int NT>
__global__ __launch_bounds__(NT) void bp_small_site_kernel(const SmallMsgItem* __restrict__ items) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const SmallMsgItem it = items[blockIdx.x];
    const int tid = threadIdx.x;
    int E = it.d; for (int k = 0; k < it.z; ++k) E *= it.chi[k];
    cx<float>* cur = reinterpret_cast<cx<float>*>(smem);
    cx<float>* nxt = cur + E;
    cx<float>* Ms = nxt + E;                                          // one message matrix, TRANSPOSED: Ms[qo + c q] = M[q + c qo] (<= 32 x 32)
    const cx<float>* psi = reinterpret_cast<const cx<float>*>(it.psi);
    if (it.mfma) { bp_small_site_mfma16<NT>(it, E, smem, (size_t)E * 16 + 32 * 32 * 8); return; }
    for (int e = tid; e < E; e += NT) cur[e] = psi[e];
    __syncthreads();
    int P = it.d;                                                     // stride of leg k
    for (int k = 0; k < it.z; ++k) {
        const int c = it.chi[k];
        if (k != it.jo && it.M[k]) {
            const cx<float>* Mg = reinterpret_cast<const cx<float>*>(it.M[k]);
            for (int e = tid; e < c * c; e += NT) Ms[(e / c) + c * (e % c)] = Mg[e];
            __syncthreads();
            // one OUTPUT element per thread and step: out[pre, qo, post] = sum_q cur[pre, q, post] M[q, qo].  Consecutive threads take consecutive output
            // elements: the stores are contiguous, the lanes of a wave read few distinct fibers (broadcasts) and consecutive entries of the transposed matrix
            // (pre, qo, post) of e = tid + NT t by carries instead of divisions: an integer division is ~40 instructions, more than the 16-term sum it would index
            int pre = tid % P, qo = (tid / P) % c, post = tid / (P * c);
            const int dpre = NT % P, dq = (NT / P) % c, dpost = NT / (P * c);
            for (int e = tid; e < E; e += NT, pre += dpre, qo += dq, post += dpost) {
                if (pre >= P) { pre -= P; ++qo; }
                if (qo >= c) { qo -= c; ++post; }
                const cx<float>* src = cur + pre + (size_t)P * c * post;
                const cx<float>* mrow = Ms + qo;
                float ar = 0.f, ai = 0.f, br = 0.f, bi = 0.f;
                int q = 0;
                for (; q + 1 < c; q += 2) {
                    const cx<float> v0 = src[(size_t)P * q], m0 = mrow[c * q], v1 = src[(size_t)P * (q + 1)], m1 = mrow[c * (q + 1)];
                    ar += v0.re * m0.re - v0.im * m0.im; ai += v0.re * m0.im + v0.im * m0.re;
                    br += v1.re * m1.re - v1.im * m1.im; bi += v1.re * m1.im + v1.im * m1.re;
                }
                if (q < c) { const cx<float> v0 = src[(size_t)P * q], m0 = mrow[c * q]; ar += v0.re * m0.re - v0.im * m0.im; ai += v0.re * m0.im + v0.im * m0.re; }
                nxt[e] = cmake<float>(ar + br, ai + bi);
            }
            __syncthreads();
            cx<float>* t = cur; cur = nxt; nxt = t;
        }
        P *= c;
    }
    // psi again, into the free copy; then out[i + co j] = sum_rest cur[rest, i] conj(psi[rest, j]) over everything but the outgoing leg
    for (int e = tid; e < E; e += NT) nxt[e] = psi[e];
    __syncthreads();
    int Po = it.d; for (int k = 0; k < it.jo; ++k) Po *= it.chi[k];
    const int co = it.chi[it.jo], nrest = E / co;
    cx<float>* out = reinterpret_cast<cx<float>*>(it.out);
    // thread = (output element o, slice of the rest index): co^2 outputs x nsl slices fill the workgroup; partial sums meet in LDS (behind the message matrix)
    const int no = co * co;
    int nsl = NT / no; if (nsl < 1) nsl = 1; if (nsl > 16) nsl = 16;
    float* red = reinterpret_cast<float*>(Ms);                        // 2 * NT floats <= 8 KiB (the matrix slot holds 8 KiB)
    for (int o0 = 0; o0 < no; o0 += NT / nsl) {
        const int o = o0 + tid / nsl, sl = tid % nsl;
        float ar = 0.f, ai = 0.f;
        if (o < no && tid < (NT / nsl) * nsl) {
            const int i = o % co, j = o / co;
            int pre = sl % Po, post = sl / Po; const int dpre = nsl % Po, dpost = nsl / Po;
            for (int r = sl; r < nrest; r += nsl, pre += dpre, post += dpost) {
                if (pre >= Po) { pre -= Po; ++post; }
                const size_t base = pre + (size_t)Po * co * post;
                const cx<float> a = cur[base + (size_t)Po * i], b = nxt[base + (size_t)Po * j];
                ar += a.re * b.re + a.im * b.im; ai += a.im * b.re - a.re * b.im;
            }
        }
        __syncthreads();
        red[2 * tid] = ar; red[2 * tid + 1] = ai;
        __syncthreads();
        if (sl == 0 && o < no && tid < (NT / nsl) * nsl) {
            float sr = 0.f, si = 0.f;
            for (int u = 0; u < nsl; ++u) { sr += red[2 * (tid + u)]; si += red[2 * (tid + u) + 1]; }
            out[o] = cmake<float>(sr, si);
        }
    }
}
void launch_bp_small_site(hipStream_t s, const SmallMsgItem* d_items, int nitems, int max_elems) {
    if (nitems <= 0) return;
    const size_t lds = (size_t)max_elems * 16 + 32 * 32 * 8;
    // 1024 threads: the LDS footprint allows one workgroup per CU, and with 256 threads (one wave per SIMD) every LDS read of the dependent sums was exposed --
    // 107 us per message of a heavy-hex degree-3 site whatever the number of messages in the launch
    set_max_dynamic_lds((const void*)bp_small_site_kernel<1024>, (size_t)(160 * 1024 - 1024));
    hipLaunchKernelGGL(bp_small_site_kernel<1024>, dim3(nitems), dim3(1024), lds, s, d_items); TNQS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// one-sided (Hestenes) Jacobi: A <- A J_1 J_2 ..., V <- V J_1 J_2 ...  until the columns of A are orthogonal.
// One workgroup per matrix, one wave per column pair, round-robin pair ordering.  m <= 64 R (R = 4 or 8: up to 512 rows), any n.
// ------------------------------------------------------------------------------------------------------------
// Rotation parameters.  f32: the hardware reciprocal (square root).  f64: the compiler's IEEE sqrt and division are ~55 dependent
// instructions each, and a Jacobi round has three of each on its critical path (most of the 1.2 us per round of the f64 kernels); the
// hardware seeds (v_rsq_f64 / v_rcp_f64, ~2^-23 relative) with two Newton steps are ~8 instructions and good to a few ulp, which is all
// a plane rotation needs (c^2 + s^2 = 1 to 1e-15).  Arguments are normal, positive numbers here (see the guards on g2).
template <class T> __device__ __forceinline__ T fast_rsqrt(T x);
template <> __device__ __forceinline__ float fast_rsqrt<float>(float x) { return __frsqrt_rn(x); }
template <> __device__ __forceinline__ double fast_rsqrt<double>(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double h = x * y; y = y * (1.5 - 0.5 * h * y);
    h = x * y; y = y * (1.5 - 0.5 * h * y);
    return y;
}
template <class T> __device__ __forceinline__ T fast_rcp(T x);
template <> __device__ __forceinline__ float fast_rcp<float>(float x) { return __frcp_rn(x); }
template <> __device__ __forceinline__ double fast_rcp<double>(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = y * (2.0 - x * y); y = y * (2.0 - x * y);
    return y;
}
template <class T> __device__ __forceinline__ T fast_sqrt(T x);          // x >= 1 at the call sites
template <> __device__ __forceinline__ float fast_sqrt<float>(float x) { return sqrtf(x); }
template <> __device__ __forceinline__ double fast_sqrt<double>(double x) { return x * fast_rsqrt<double>(x); }
template <class T, int R>          // R rows per lane: m <= 64 R
__global__ __launch_bounds__(1024) void jacobi_kernel(const JacobiItem* __restrict__ items, int max_sweeps) {
    __shared__ int s_rot;
    const JacobiItem it = items[blockIdx.x];
    if (it.only_if && *it.only_if == 0) return;          // conditional item (svd_batch: polishing sweeps only where the preprocessing failed)
    if (it.pre && theta_pre_takes(it.dyn, it.dm, it.dn, it.QB != nullptr)) return;      // taken by theta_svd_pre_kernel
    cx<T>* A = reinterpret_cast<cx<T>*>(it.A);
    cx<T>* V = reinterpret_cast<cx<T>*>(it.V);
    int m_ = it.m, n_ = it.n;
    if (it.dyn) { int nf; theta_dims(it.dyn, it.dm, it.dn, m_, nf, n_); }
    const int m = m_, n = n_;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int ne = n + (n & 1);
    // (f32: 2 eps, see jacobi_lds_sweeps)
    const T tol = eps_of<T>() * (sizeof(T) == 4 ? (T)2 : sqrt((T)(m > 4 ? m : 4)));
    // scale to ||A||_F = O(1) by an exact power of two for the sweeps (see jacobi_lds_kernel: squared inner products underflow in f32)
    __shared__ double s_redg[17];
    double frog = 0;
    for (int e = threadIdx.x; e < m * n; e += blockDim.x) { cx<T> v = A[e]; frog += (double)v.re * v.re + (double)v.im * v.im; }
    frog = block_sum(frog, s_redg);
    int kexp = 0;
    if (frog > 0 && frog < 1e300) { kexp = -(ilogb(frog) / 2); kexp = kexp > 120 ? 120 : (kexp < -120 ? -120 : kexp); }
    const T sc_in = (T)ldexp(1.0, kexp), sc_out = (T)ldexp(1.0, -kexp);
    if (kexp != 0) { for (int e = threadIdx.x; e < m * n; e += blockDim.x) { cx<T> v = A[e]; A[e] = cmake<T>(v.re * sc_in, v.im * sc_in); } }
    __syncthreads();
    int sweep = 0;
    for (; sweep < max_sweeps && n > 1; ++sweep) {
        if (threadIdx.x == 0) s_rot = 0;
        __syncthreads();
        for (int round = 0; round < ne - 1; ++round) {
            for (int pi = w; pi < ne / 2; pi += nw) {
                int p, q;
                if (pi == 0) { p = ne - 1; q = round; }
                else { p = (round + pi) % (ne - 1); q = (round - pi + (ne - 1)) % (ne - 1); }
                if (p > q) { int t = p; p = q; q = t; }
                if (q >= n) continue;
                cx<T> ap[R], aq[R];
                T alpha = 0, beta = 0, gre = 0, gim = 0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    int i = lane + 64 * r;
                    if (i < m) {
                        ap[r] = A[i + (size_t)m * p]; aq[r] = A[i + (size_t)m * q];
                        alpha += ap[r].re * ap[r].re + ap[r].im * ap[r].im;
                        beta += aq[r].re * aq[r].re + aq[r].im * aq[r].im;
                        gre += ap[r].re * aq[r].re + ap[r].im * aq[r].im;     // conj(ap) * aq
                        gim += ap[r].re * aq[r].im - ap[r].im * aq[r].re;
                    }
                }
                alpha = wave_sum(alpha); beta = wave_sum(beta); gre = wave_sum(gre); gim = wave_sum(gim);
                const T g2 = gre * gre + gim * gim;
                if (g2 > (sizeof(T) == 4 ? (T)1e-36 : (T)1e-290) && g2 > tol * tol * alpha * beta) {      // (g2 normal: see fast_rsqrt)
                    const T iga = fast_rsqrt<T>(g2);
                    const T pre = gre * iga, pim = -gim * iga;         // e^{-i phi}
                    const T zeta = (beta - alpha) * (T)0.5 * iga;
                    const T az = fabs(zeta);
                    const T t = (zeta >= 0 ? (T)1 : (T)-1) * fast_rcp<T>(az + fast_sqrt<T>(1 + az * az));
                    const T c = fast_rsqrt<T>(1 + t * t), sn = c * t;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        int i = lane + 64 * r;
                        if (i < m) {
                            T qre = aq[r].re * pre - aq[r].im * pim, qim = aq[r].re * pim + aq[r].im * pre;
                            A[i + (size_t)m * p] = cmake<T>(c * ap[r].re - sn * qre, c * ap[r].im - sn * qim);
                            A[i + (size_t)m * q] = cmake<T>(sn * ap[r].re + c * qre, sn * ap[r].im + c * qim);
                        }
                    }
                    if (V)
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        int i = lane + 64 * r;
                        if (i < n) {
                            cx<T> vp = V[i + (size_t)n * p], vq = V[i + (size_t)n * q];
                            T qre = vq.re * pre - vq.im * pim, qim = vq.re * pim + vq.im * pre;
                            V[i + (size_t)n * p] = cmake<T>(c * vp.re - sn * qre, c * vp.im - sn * qim);
                            V[i + (size_t)n * q] = cmake<T>(sn * vp.re + c * qre, sn * vp.im + c * qim);
                        }
                    }
                    if (lane == 0) s_rot = 1;
                }
            }
            __syncthreads();
        }
        const int rot = s_rot;
        __syncthreads();
        if (!rot) { ++sweep; break; }
    }
    __syncthreads();
    if (kexp != 0) { for (int e = threadIdx.x; e < m * n; e += blockDim.x) { cx<T> v = A[e]; A[e] = cmake<T>(v.re * sc_out, v.im * sc_out); } }
    if (threadIdx.x == 0 && it.sweeps_out) *it.sweeps_out = sweep;
}
// LDS-resident variant: A (and V when it fits / is wanted) live in LDS for the whole factorisation; global memory is
// touched twice.  it.V == nullptr: rotations are not accumulated (the caller recovers V = A0^dagger (U Sigma) Sigma^-2).
// A QUARTER wave (16 lanes) owns one column pair, so a 16-wave workgroup rotates 64 pairs at once (one full round of a
// 128-column matrix); the dot products reduce inside 16-lane rows.  Columns are padded by 2 elements so the four
// quarter-waves of a wave hit different LDS banks.
// all-reduce over a 16-lane row with DPP row rotations (VALU, no LDS crossbar): after adding the rotations by 8, 4, 2, 1 every lane
// holds the row sum (the same summation tree in every lane of the row, so the four quarter-waves' decisions stay uniform per row)
template <int ROR> __device__ __forceinline__ float dpp_ror_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + ROR, 0xf, 0xf, false));
}
template <int ROR> __device__ __forceinline__ double dpp_ror_d(double v) {
    long long b = __builtin_bit_cast(long long, v);
    int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x120 + ROR, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x120 + ROR, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_ror_f<8>(v); v += dpp_ror_f<4>(v); v += dpp_ror_f<2>(v); v += dpp_ror_f<1>(v);
    return v;
}
__device__ __forceinline__ double row16_sum(double v) {
    v += dpp_ror_d<8>(v); v += dpp_ror_d<4>(v); v += dpp_ror_d<2>(v); v += dpp_ror_d<1>(v);
    return v;
}
// the sweeps of the LDS-resident factorisation.  FULL: m == 16*RQ, n even and n/2 a multiple of 4 -- every quarter-wave of every
// participating wave owns a real pair and all RQ row slots are real rows, so the per-row / per-pair guards (exec-mask juggling in the
// hottest loop) disappear.
template <class T, int RQ, bool FULL>
__device__ __forceinline__ int jacobi_lds_sweeps(cx<T>* A, cx<T>* V, bool hasV, int m, int n, int mp, int np_, int max_sweeps, T tiny, int* s_rot) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int grp = lane >> 4, l16 = lane & 15;
    const int ne = n + (n & 1);
    const int nslots = 4 * nw;
    // Convergence threshold on |<a_p, a_q>| / (|a_p| |a_q|).  f64: eps sqrt(m), the worst-case rounding bound of the inner product.  f32: 2 eps -- the
    // rounding noise of an m-term inner product of nearly orthogonal columns is ~eps / sqrt(m) of |a_p| |a_q| (random signs), so 2 eps is still 20 x
    // above it, and the looser eps sqrt(m) = 1.3e-6 (rounds 1-3) left the singular vectors an order of magnitude less orthogonal than LAPACK's:
    // measured on a ten-layer chi = 32 evolution, <Z> drifted 1-3e-5 from the ComplexF64 run with the old threshold and 1-3e-6 with this one (the
    // oracle's own f32 run: 1-4e-6), for 8.1 instead of 7.1 sweeps per gate.
    const T tol = eps_of<T>() * (sizeof(T) == 4 ? (T)2 : sqrt((T)(m > 4 ? m : 4)));
    const int rq = (m + 15) >> 4, rqv = (n + 15) >> 4;
    int sweep = 0;
    for (; sweep < max_sweeps && n > 1; ++sweep) {
        if (threadIdx.x == 0) *s_rot = 0;
        __syncthreads();
        for (int round = 0; round < ne - 1; ++round) {
            for (int base = 4 * w; base < ne / 2; base += nslots) {
                // wave-uniform trip count (the row reductions need all four quarter-waves); idle quarters are predicated off
                const int pi = base + grp;
                int p = 0, q = 0; bool act = FULL || pi < ne / 2;
                if (act) {
                    if (pi == 0) { p = ne - 1; q = round; }
                    else { p = round + pi; if (p >= ne - 1) p -= ne - 1; q = round - pi; if (q < 0) q += ne - 1; }
                    if (p > q) { int t = p; p = q; q = t; }
                    if (!FULL) act = q < n;
                }
                cx<T> ap[RQ], aq[RQ];
                T alpha = 0, beta = 0, gre = 0, gim = 0;
#pragma unroll
                for (int r = 0; r < RQ; ++r) {
                    int i = l16 + 16 * r;
                    if (FULL || (r < rq && act && i < m)) {
                        ap[r] = A[i + mp * p]; aq[r] = A[i + mp * q];
                        alpha += ap[r].re * ap[r].re + ap[r].im * ap[r].im;
                        beta += aq[r].re * aq[r].re + aq[r].im * aq[r].im;
                        gre += ap[r].re * aq[r].re + ap[r].im * aq[r].im;
                        gim += ap[r].re * aq[r].im - ap[r].im * aq[r].re;
                    }
                }
                alpha = row16_sum(alpha); beta = row16_sum(beta); gre = row16_sum(gre); gim = row16_sum(gim);
                const T g2 = gre * gre + gim * gim;
                // f32: g2 must be a NORMAL number -- the fast reciprocal square root returns inf for (flushed) denormals
                const bool rot = act && g2 > (sizeof(T) == 4 ? (T)1e-36 : (T)1e-290) && g2 > tol * tol * alpha * beta && !(alpha < tiny && beta < tiny);
                if (rot) {
                    const T iga = fast_rsqrt<T>(g2);
                    const T pre = gre * iga, pim = -gim * iga;
                    const T zeta = (beta - alpha) * (T)0.5 * iga;
                    const T az = fabs(zeta);
                    const T t = (zeta >= 0 ? (T)1 : (T)-1) * fast_rcp<T>(az + fast_sqrt<T>(1 + az * az));
                    const T c = fast_rsqrt<T>(1 + t * t), sn = c * t;
#pragma unroll
                    for (int r = 0; r < RQ; ++r) {
                        int i = l16 + 16 * r;
                        if (FULL || (r < rq && i < m)) {
                            T qre = aq[r].re * pre - aq[r].im * pim, qim = aq[r].re * pim + aq[r].im * pre;
                            A[i + mp * p] = cmake<T>(c * ap[r].re - sn * qre, c * ap[r].im - sn * qim);
                            A[i + mp * q] = cmake<T>(sn * ap[r].re + c * qre, sn * ap[r].im + c * qim);
                        }
                    }
                    if (hasV) {
#pragma unroll
                        for (int r = 0; r < RQ; ++r) {
                            int i = l16 + 16 * r;
                            if (r < rqv && i < n) {
                                cx<T> vp = V[i + np_ * p], vq = V[i + np_ * q];
                                T qre = vq.re * pre - vq.im * pim, qim = vq.re * pim + vq.im * pre;
                                V[i + np_ * p] = cmake<T>(c * vp.re - sn * qre, c * vp.im - sn * qim);
                                V[i + np_ * q] = cmake<T>(sn * vp.re + c * qre, sn * vp.im + c * qim);
                            }
                        }
                    }
                    if (l16 == 0) *s_rot = 1;
                }
            }
            __syncthreads();
        }
        const int rotd = *s_rot;
        __syncthreads();
        if (!rotd) { ++sweep; break; }
    }
    return sweep;
}
// ComplexF32, full tiles: the same sweeps written on (re, im) pairs so that the compiler emits packed f32 operations
// (v_pk_fma_f32 with operand swizzles): 10 packed operations per row instead of ~22
typedef float v2f_t __attribute__((ext_vector_type(2)));
template <int RQ>
__device__ __forceinline__ int jacobi_lds_sweeps_f32_full(cx<float>* A, int m, int n, int mp, int max_sweeps, float tiny, int* s_rot) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int grp = lane >> 4, l16 = lane & 15;
    const int ne = n;                                   // n is even here
    const int nslots = 4 * nw;
    const float tol = eps_of<float>() * 2.0f;                       // (see jacobi_lds_sweeps)
    v2f_t* Av = reinterpret_cast<v2f_t*>(A);
    int sweep = 0;
    for (; sweep < max_sweeps && n > 1; ++sweep) {
        if (threadIdx.x == 0) *s_rot = 0;
        __syncthreads();
        for (int round = 0; round < ne - 1; ++round) {
            for (int base = 4 * w; base < ne / 2; base += nslots) {
                const int pi = base + grp;
                int p, q;
                if (pi == 0) { p = ne - 1; q = round; }
                else { p = round + pi; if (p >= ne - 1) p -= ne - 1; q = round - pi; if (q < 0) q += ne - 1; }
                if (p > q) { int t = p; p = q; q = t; }
                v2f_t* cp = Av + l16 + mp * p; v2f_t* cq = Av + l16 + mp * q;
                v2f_t ap[RQ], aq[RQ];
                v2f_t sa = {0.f, 0.f}, sb = {0.f, 0.f}, g1 = {0.f, 0.f}, g2v = {0.f, 0.f};
#pragma unroll
                for (int r = 0; r < RQ; ++r) {
                    ap[r] = cp[16 * r]; aq[r] = cq[16 * r];
                    sa += ap[r] * ap[r]; sb += aq[r] * aq[r];
                    g1 += ap[r] * aq[r];
                    g2v += ap[r] * __builtin_shufflevector(aq[r], aq[r], 1, 0);
                }
                float alpha = row16_sum(sa.x + sa.y), beta = row16_sum(sb.x + sb.y), gre = row16_sum(g1.x + g1.y), gim = row16_sum(g2v.x - g2v.y);
                const float g2 = gre * gre + gim * gim;
                const bool rot = g2 > 1e-36f && g2 > tol * tol * alpha * beta && !(alpha < tiny && beta < tiny);
                if (rot) {
                    const float iga = fast_rsqrt<float>(g2);
                    const float pre = gre * iga, pim = -gim * iga;
                    const float zeta = (beta - alpha) * 0.5f * iga;
                    const float az = fabsf(zeta);
                    const float t = (zeta >= 0 ? 1.f : -1.f) * fast_rcp<float>(az + sqrtf(1 + az * az));
                    const float c = fast_rsqrt<float>(1 + t * t), sn = c * t;
                    const v2f_t e1 = {pre, pim}, e2 = {-pim, pre}, cc = {c, c}, ss = {sn, sn};
#pragma unroll
                    for (int r = 0; r < RQ; ++r) {
                        const v2f_t qv = __builtin_shufflevector(aq[r], aq[r], 0, 0) * e1 + __builtin_shufflevector(aq[r], aq[r], 1, 1) * e2;
                        cp[16 * r] = cc * ap[r] - ss * qv;
                        cq[16 * r] = ss * ap[r] + cc * qv;
                    }
                    if (l16 == 0) *s_rot = 1;
                }
            }
            __syncthreads();
        }
        const int rotd = *s_rot;
        __syncthreads();
        if (!rotd) { ++sweep; break; }
    }
    return sweep;
}
template <class T, int RQ>              // RQ = rows per lane: m <= 16*RQ
__global__ __launch_bounds__(1024) void jacobi_lds_kernel(const JacobiItem* __restrict__ items, int max_sweeps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_rot;
    const JacobiItem it = items[blockIdx.x];
    if (it.pre && theta_pre_takes(it.dyn, it.dm, it.dn, it.QB != nullptr)) return;      // taken by theta_svd_pre_kernel
    cx<T>* Ag = reinterpret_cast<cx<T>*>(it.A);
    cx<T>* Vg = reinterpret_cast<cx<T>*>(it.V);
    int m_ = it.m, n_ = it.n;
    if (it.dyn) { int nf; theta_dims(it.dyn, it.dm, it.dn, m_, nf, n_); }      // dimensions found on the device (JacobiItem::dyn)
    const int m = m_, n = n_;
    const int mp = m + 2, np_ = n + 2;     // padded column pitches
    cx<T>* A = reinterpret_cast<cx<T>*>(smem);
    cx<T>* V = A + (size_t)mp * n;
    const bool hasV = Vg != nullptr;
    // Without V the order of the columns is free (the caller ranks the singular values itself): they enter the sweeps sorted by decreasing
    // norm (de Rijk), which the cyclic sweeps converge from in fewer passes than from an arbitrary order
    __shared__ float s_cn[256]; __shared__ unsigned char s_pos[256];
    const bool sorted = !hasV && n <= 256 && n > 2;
    if (sorted) {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
        for (int j = w; j < n; j += nw) {
            float s2 = 0.f;
            for (int i = lane; i < m; i += 64) { const cx<T> v = Ag[i + (size_t)m * j]; s2 += (float)v.re * (float)v.re + (float)v.im * (float)v.im; }
            s2 = wave_sum(s2);
            if (lane == 0) s_cn[j] = s2 == s2 ? s2 : 0.f;
        }
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += blockDim.x) {
            int rk = 0; const float cj = s_cn[j];
            for (int v = 0; v < n; ++v) rk += (s_cn[v] > cj) || (s_cn[v] == cj && v < j);
            s_pos[j] = (unsigned char)rk;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < m * n; e += blockDim.x) A[(e % m) + mp * (int)s_pos[e / m]] = Ag[e];
    } else
    for (int e = threadIdx.x; e < m * n; e += blockDim.x) A[(e % m) + mp * (e / m)] = Ag[e];
    if (hasV) for (int e = threadIdx.x; e < n * n; e += blockDim.x) V[(e % n) + np_ * (e / n)] = cmake<T>((e % n) == (e / n) ? (T)1 : (T)0, (T)0);
    __syncthreads();
    // ||A||_F^2 is invariant under the rotations.  A pair of columns that are BOTH below n eps^2 ||A||_F^2 (singular values under
    // ~sqrt(n) eps ||A||_F: rounding noise of a rank-deficient matrix) is left alone -- otherwise noise columns keep rotating
    // against each other for many sweeps without changing any singular value that matters.
    __shared__ double s_red[17];
    double fro = 0;
    for (int e = threadIdx.x; e < m * n; e += blockDim.x) { cx<T> v = A[(e % m) + mp * (e / m)]; fro += (double)v.re * v.re + (double)v.im * v.im; }
    fro = block_sum(fro, s_red);
    // The sweeps square inner products (g^2, alpha*beta): in f32 that underflows for a matrix of small magnitude (theta of a state whose
    // tensors carry a small norm: singular values 1e-5 already put the products of the smaller columns into the denormal range, the
    // rotation phases lose their unit modulus and the "rotations" stop being unitary).  The matrix is therefore scaled by an exact power
    // of two to ||A||_F = O(1) for the sweeps and scaled back when it is written out; V does not change.
    int kexp = 0;
    if (fro > 0 && fro < 1e300) { kexp = -(ilogb(fro) / 2); kexp = kexp > 120 ? 120 : (kexp < -120 ? -120 : kexp); }
    const T sc_in = (T)ldexp(1.0, kexp), sc_out = (T)ldexp(1.0, -kexp);
    if (kexp != 0) {
        for (int e = threadIdx.x; e < m * n; e += blockDim.x) { cx<T>& v = A[(e % m) + mp * (e / m)]; v.re *= sc_in; v.im *= sc_in; }
        fro = ldexp(fro, 2 * kexp);
        __syncthreads();
    }
    const T tiny = (T)((double)n * (double)eps_of<T>() * (double)eps_of<T>() * fro);
    const bool full = (m == 16 * RQ) && !(n & 1) && !((n >> 1) & 3);
    int sweep;
    if (full && sizeof(T) == 4 && !hasV) sweep = jacobi_lds_sweeps_f32_full<RQ>(reinterpret_cast<cx<float>*>(A), m, n, mp, max_sweeps, (float)tiny, &s_rot);
    else if (full) sweep = jacobi_lds_sweeps<T, RQ, true>(A, V, hasV, m, n, mp, np_, max_sweeps, tiny, &s_rot);
    else sweep = jacobi_lds_sweeps<T, RQ, false>(A, V, hasV, m, n, mp, np_, max_sweeps, tiny, &s_rot);
    __syncthreads();
    for (int e = threadIdx.x; e < m * n; e += blockDim.x) { cx<T> v = A[(e % m) + mp * (e / m)]; Ag[e] = cmake<T>(v.re * sc_out, v.im * sc_out); }
    if (hasV) for (int e = threadIdx.x; e < n * n; e += blockDim.x) Vg[e] = V[(e % n) + np_ * (e / n)];
    if (threadIdx.x == 0 && it.sweeps_out) *it.sweeps_out = sweep;
}

typedef double v4d_t __attribute__((ext_vector_type(4)));
template <class FA, class FB> __device__ __forceinline__ void ztile_mm(int K, int i, int j, FA fa, FB fb, v4d_t& cr, v4d_t& ci) {
    const int kq = (threadIdx.x & 63) >> 4;
    for (int k0 = 0; k0 < K; k0 += 4) {
        const cx<double> a = fa(i, k0 + kq), b = fb(k0 + kq, j);
        cr = __builtin_amdgcn_mfma_f64_16x16x4f64(a.re, b.re, cr, 0, 0, 0);
        cr = __builtin_amdgcn_mfma_f64_16x16x4f64(-a.im, b.im, cr, 0, 0, 0);
        ci = __builtin_amdgcn_mfma_f64_16x16x4f64(a.re, b.im, ci, 0, 0, 0);
        ci = __builtin_amdgcn_mfma_f64_16x16x4f64(a.im, b.re, ci, 0, 0, 0);
    }
}

// the same tile product for FULL tiles with complex f32 operands in LDS (no guards, loads hoisted by unrolling): this lane supplies A[row l15][k] = ap[k * as]
// (conjugated when CA) and B[k][column l15] = bp[k * bs]; K a multiple of 4
template <bool CA> __device__ __forceinline__ void tile_mm_f32(const cx<float>* ap, int as, const cx<float>* bp, int bs, int K, v4d_t& cr, v4d_t& ci) {
    const int kq = (threadIdx.x & 63) >> 4;
    ap += kq * as; bp += kq * bs;
#pragma unroll 4
    for (int k0 = 0; k0 < K; k0 += 4) {
        const cx<float> a = ap[k0 * as], b = bp[k0 * bs];
        const double ar = a.re, ai = CA ? -(double)a.im : (double)a.im, br = b.re, bi = b.im;
        cr = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, cr, 0, 0, 0);
        cr = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai, bi, cr, 0, 0, 0);
        ci = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, bi, ci, 0, 0, 0);
        ci = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, br, ci, 0, 0, 0);
    }
}
// ------------------------------------------------------------------------------------------------------------
// Preconditioned theta SVD (round 5): ComplexF32, V not wanted, tall or square A (m >= n), n <= 64, m <= 128 -- the 128 x 64 low-rank factor
// of a chi = 32 gate and every smaller theta.  ONE workgroup per gate, everything in LDS:
//   1. A -> LDS, columns sorted by decreasing norm (de Rijk), scaled to ||A||_F = O(1) by a power of two;
//   2. G = A^dagger A in f64 (f32 products are exact in f64: G is the exact Gram matrix of the rounded A);
//   3. G = L L^dagger, right-looking Cholesky with one barrier per column (a collapsed pivot -- A rank deficient, the normal case early in
//      an evolution -- is replaced by 1e-13 of the largest one: directions below 3e-7 sigma_max are f32 noise of A anyway);
//   4. one-sided Jacobi on the COLUMNS OF L (n x n, f32): L J = U_L Sigma.  L = R^dagger of the QR factorisation of A: orthogonalising the rows
//      of R instead of the columns of A is the Drmac-Veselic preconditioning -- the sorted triangular factor is graded, L^dagger L is much
//      closer to diagonal than A^dagger A: 4-5 sweeps instead of 8 on the thetas of the benchmark (scratch numpy model: oracle thetas of a 4 x 4
//      chi = 32 lattice, 7-9 -> 3-5; evolved chi = 16 states, 7-8 -> 5-7), and each sweep rotates n rows instead of m;
//   5. A^dagger A = L L^dagger = U_L Sigma^2 U_L^dagger: the normalised columns of L J ARE the right singular vectors of A, so
//      U Sigma = A U_L -- an (m x n)(n x n) product accumulated in f64 -- goes back to global memory where the rotated A used to go.
// No inverse of R, no accumulated rotations.  What the kernel replaces took 0.46-0.81 ms per colour batch (8.1 sweeps x 63 rounds on 128 rows).
// ------------------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(NT) void theta_svd_pre_kernel(const JacobiItem* __restrict__ items, int max_sweeps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_rot;
    __shared__ double s_red[17];
    __shared__ double s_cn[64]; __shared__ double s_piv[64]; __shared__ unsigned char s_pos[64]; __shared__ unsigned char s_perm[64];
    __shared__ double s_dmax; __shared__ double s_sig[64]; __shared__ int s_bad;
    // it.V is never an output here (V is not accumulated); the kernel tests pass a buffer of 8 x 64-bit slots that receives the constant-rate clock at the
    // phase boundaries (engine: null)
#define PRE_STAMP(k) do { if (tstamp && threadIdx.x == 0) tstamp[k] = wall_clock64(); } while (0)
    const JacobiItem it = items[blockIdx.x];
    unsigned long long* tstamp = reinterpret_cast<unsigned long long*>(it.V);
    cx<float>* Ag = reinterpret_cast<cx<float>*>(it.A);
    int m_ = it.m, n_ = it.n;
    if (it.dyn) { int nf; theta_dims(it.dyn, it.dm, it.dn, m_, nf, n_); }
    const int m = m_, n = n_, tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6, nw = NT >> 6;
    PRE_STAMP(0);
    // pre != 0 (engine): the item is taken only when the low-rank route survived on the device and its factor fits (theta_pre_takes); the plain Jacobi
    // kernel launched next to this one makes the complementary decision.  pre == 0 (kernel tests): the dimensions given decide
    if (it.pre ? !theta_pre_takes(it.dyn, it.dm, it.dn, it.QB != nullptr) : (n < 2 || m < n || n > 64 || m > 128)) return;
    const int mp = m + 2, gp = n + 1, xp = n + 2;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    cx<float>* Mf = reinterpret_cast<cx<float>*>(smem);                                   // sorted A, column a at mp * a
    const size_t m_bytes = (((size_t)mp * n * sizeof(cx<float>)) + 15) & ~(size_t)15;
    cx<double>* Gd = reinterpret_cast<cx<double>*>(smem + m_bytes);                      // G / L (lower triangle), element (i, j) at i + gp * j
    cx<float>* X = reinterpret_cast<cx<float>*>(smem + m_bytes);                         // later: L in f32, column k at xp * k (over the start of Gd)
    // ---- 1. column norms (first pass over A: 64 KiB, L2 resident afterwards), de Rijk order, SORTED load (column a of the LDS copy = column s_perm[a] of A);
    // the power-of-two scaling is applied to G (exactly) instead of to the entries ------------------------------------------------------------------
    for (int j = w; j < n; j += nw) {
        double s2 = 0;
        for (int i = lane; i < m; i += 64) { const cx<float> v = Ag[i + (size_t)m * j]; s2 += (double)v.re * v.re + (double)v.im * v.im; }
        s2 = wave_sum(s2);
        if (lane == 0) { s_cn[j] = s2 == s2 ? s2 : 0.0; if (!(s2 == s2) || s2 > 1e300) s_bad = 1; }
    }
    __syncthreads();
    for (int j = tid; j < n; j += NT) {
        int rk = 0; const double cj = s_cn[j];
        for (int v = 0; v < n; ++v) rk += (s_cn[v] > cj) || (s_cn[v] == cj && v < j);
        s_pos[j] = (unsigned char)rk; s_perm[rk] = (unsigned char)j;
    }
    double fro = 0; for (int j = tid; j < n; j += NT) fro += s_cn[j];
    fro = block_sum(fro, s_red);                                                         // (also orders s_pos before the load below)
    // Degenerate input (round-5 advisor finding).  theta identically ZERO: its SVD is U Sigma = 0 -- A stays as it is, V = 0; before this guard the largest
    // Cholesky pivot was 0, every pivot was replaced by 1, L became the identity and the unformed columns left as sigma_j e_0 with sigma = 1: weight in S and in
    // the truncation error that the matrix does not have.  A NaN / infinite entry: the column norm was mapped to 0 and the column treated as rank deficient
    // instead of flagged -- now A stays as it is (the NaNs reach gate_finish, which reports TNQS_ERR_NUMERIC like the plain Jacobi route) and V is NaN too
    if (s_bad || !(fro > 0)) {
        if (it.Vout) {
            cx<float>* Vg = reinterpret_cast<cx<float>*>(it.Vout);
            int rows = n;
            if (it.QB && it.dyn && it.dyn[7] > 0) { int mq, nq, kq_; theta_dims(it.dyn, it.dm, it.dn, mq, nq, kq_); rows = nq; (void)mq; (void)kq_; }
            const float fill = s_bad ? __builtin_nanf("") : 0.f;
            for (int e = tid; e < rows * n; e += NT) Vg[e] = cmake<float>(fill, fill);
        }
        if (tid == 0 && it.sweeps_out) *it.sweeps_out = 0;
        return;
    }
    int kexp = 0;
    if (fro > 0 && fro < 1e300) { kexp = -(ilogb(fro) / 2); kexp = kexp > 120 ? 120 : (kexp < -120 ? -120 : kexp); }
    const double sc2 = ldexp(1.0, 2 * kexp), sc_out = ldexp(1.0, -kexp);                 // G is formed at ||A||_F = O(1): L, the sweeps and s_sig live at that scale
    fro = ldexp(fro, 2 * kexp);
    for (int e = tid; e < m * n; e += NT) Mf[(e % m) + mp * (int)s_pos[e / m]] = Ag[e];
    __syncthreads();
    PRE_STAMP(1);
    const int l15 = lane & 15, kq = lane >> 4;
    const bool full16 = !(m & 15) && !(n & 15);          // every 16 x 16 tile is full: the unguarded tile products
    // ---- 2. G = A^dagger A (sorted order), lower triangle, f64 matrix cores: one wave per 16 x 16 tile, operands converted from the f32 columns in LDS ------
    {
        const int nt = (n + 15) >> 4, ntile = nt * (nt + 1) / 2;
        for (int t = w; t < ntile; t += nw) {
            int ti = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
            while (ti * (ti + 1) / 2 > t) --ti;
            while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
            const int tj = t - ti * (ti + 1) / 2;                                       // ti >= tj
            const int ia = 16 * ti + l15, ja = 16 * tj + l15;
            const cx<float>* ci_ = Mf + mp * (ia < n ? ia : 0); const cx<float>* cj_ = Mf + mp * (ja < n ? ja : 0);
            v4d_t cr = {0, 0, 0, 0}, ci = {0, 0, 0, 0};
            if (full16) tile_mm_f32<true>(ci_, 1, cj_, 1, m, cr, ci);                    // G[i][j] = sum_r conj(A[r][i]) A[r][j]
            else ztile_mm(m, ia, ja,
                     [&](int i, int r) { cx<double> v = cmake<double>(0, 0); if (i < n && r < m) { const cx<float> a = ci_[r]; v = cmake<double>(a.re, -a.im); } return v; },
                     [&](int r, int j) { cx<double> v = cmake<double>(0, 0); if (j < n && r < m) { const cx<float> a = cj_[r]; v = cmake<double>(a.re, a.im); } return v; }, cr, ci);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ti + kq + 4 * r, j = 16 * tj + l15;
                if (i < n && j < n && i >= j) Gd[i + gp * j] = cmake<double>(cr[r] * sc2, i == j ? 0.0 : ci[r] * sc2);
            }
        }
    }
    __syncthreads();
    PRE_STAMP(2);
    // ---- 3. Cholesky, right-looking from the unscaled column, one barrier per column (see chol_kernel) -------------------------------------
    if (tid < 64) {
        double mx = 0; for (int i = tid; i < n; i += 64) mx = fmax(mx, Gd[i + gp * i].re);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
        if (tid == 0) s_dmax = mx;
    }
    __syncthreads();
    const double ptiny = 1e-13 * s_dmax;
    auto pivot_of = [&](int k) { const double d = Gd[k + gp * k].re; return (d > ptiny) ? d : (ptiny > 0 ? ptiny : 1.0); };
    {
        constexpr int UT = (64 * 63 / 2 + NT - 1) / NT;           // trailing-triangle elements a thread owns at most (nested triangular numbering)
        unsigned char tr[UT], tc[UT];
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            const int e = tid + NT * u;
            int r = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
            while (r * (r + 1) / 2 > e) --r;
            while ((r + 1) * (r + 2) / 2 <= e) ++r;
            tr[u] = (unsigned char)r; tc[u] = (unsigned char)(e - r * (r + 1) / 2);
        }
        for (int k = 0; k < n - 1; ++k) {
            // a collapsed pivot (A rank deficient: the Schur complement left is rounding noise) ends the column: no trailing update from it, its entries below
            // the diagonal are dropped in step 4.  (Continuing with the clamped pivot divides noise by 1e-13: measured on a rank-20 factor, the entries of the
            // 44 noise columns grew to 1e22.)  Every thread reads the same diagonal entry, so the decision is uniform and the barrier count stays the same.
            if (!(Gd[k + gp * k].re > ptiny)) { __syncthreads(); continue; }
            const double dinv = 1.0 / pivot_of(k);
            const int mm = n - k - 1, k1 = k + 1, nt = mm * (mm + 1) / 2;
            const cx<double>* colk = Gd + gp * k;
            cx<double> li[UT], lj[UT], v[UT];
#pragma unroll
            for (int u = 0; u < UT; ++u) if (tid + NT * u < nt) { const int i = k1 + tr[u], j = k1 + tc[u]; li[u] = colk[i]; lj[u] = colk[j]; v[u] = Gd[i + gp * j]; }
#pragma unroll
            for (int u = 0; u < UT; ++u) if (tid + NT * u < nt) {
                const double sr = li[u].re * dinv, si = li[u].im * dinv;
                v[u].re -= sr * lj[u].re + si * lj[u].im; v[u].im -= si * lj[u].re - sr * lj[u].im;
                Gd[(k1 + tr[u]) + gp * (k1 + tc[u])] = v[u];
            }
            __syncthreads();
        }
    }
    for (int k = tid; k < n; k += NT) s_piv[k] = 1.0 / sqrt(pivot_of(k));
    __syncthreads();
    // ---- 4. X = L in f32 (over the start of the f64 array: everything is read before anything is written) --------------------------------
    {
        constexpr int UX = (64 * 64 + NT - 1) / NT;
        cx<float> xv[UX];
#pragma unroll
        for (int u = 0; u < UX; ++u) {
            const int e = tid + NT * u; const int i = e % n, k = e / n;
            xv[u] = cmake<float>(0.f, 0.f);
            if (e < n * n && i >= k) {
                const cx<double> a = Gd[i + gp * k]; const double r = s_piv[k];
                const bool dead = !(Gd[k + gp * k].re > ptiny);                      // collapsed pivot: the column is (0, ..., sqrt(ptiny), 0, ..., 0)
                xv[u] = (i == k) ? cmake<float>((float)(pivot_of(k) * r), 0.f) : (dead ? cmake<float>(0.f, 0.f) : cmake<float>((float)(a.re * r), (float)(a.im * r)));
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < UX; ++u) { const int e = tid + NT * u; if (e < n * n) X[(e % n) + xp * (e / n)] = xv[u]; }
    }
    __syncthreads();
    PRE_STAMP(3);
    // ---- 5. sweeps on the columns of L (n rows): the quarter-wave sweeps of jacobi_lds_kernel (an eighth-wave layout with four waves measured the same 45 us
    // per 64 x 64 sweep and, with its two-level rotation formulas, singular values 6-9 x less accurate) -------------------------------------------------------
    const float tiny = (float)((double)n * (double)eps_of<float>() * (double)eps_of<float>() * fro);
    int sweep;
    if (n == 64) sweep = jacobi_lds_sweeps_f32_full<4>(X, n, n, xp, max_sweeps, tiny, &s_rot);
    else if (n == 32) sweep = jacobi_lds_sweeps_f32_full<2>(X, n, n, xp, max_sweeps, tiny, &s_rot);      // chi = 16 gates: 16 full pairs on four waves, no guards
    else sweep = jacobi_lds_sweeps<float, 4, false>(X, (cx<float>*)nullptr, false, n, n, xp, 0, max_sweeps, tiny, &s_rot);
    __syncthreads();
    PRE_STAMP(4);
    // ---- 6. sigma_j = |x_j| (relative accuracy: what the truncation is decided on), U_L = normalised columns.  Only the `cap` largest singular values can survive
    // the truncation (JacobiItem::cap = the bond dimension cap of the gate): U Sigma and V are formed for those columns only; the others leave as sigma_j e_0,
    // which carries their weight into the truncation error and nothing else ------------------------------------------------------------------------------
    __shared__ unsigned char s_keep[64]; __shared__ int s_nk;
    for (int j = w; j < n; j += nw) {
        double s2 = 0;
        for (int i = lane; i < n; i += 64) { const cx<float> v = X[i + xp * j]; s2 += (double)v.re * v.re + (double)v.im * v.im; }
        s2 = wave_sum(s2);
        if (lane == 0) { s_cn[j] = s2 > 0 ? 1.0 / sqrt(s2) : 0.0; s_sig[j] = sqrt(s2); }
    }
    __syncthreads();
    const int cap = (it.cap > 0 && it.cap < n) ? it.cap : n;
    __shared__ unsigned char s_rank[64];
    for (int j = tid; j < n; j += NT) {
        int rk = 0; const double sj_ = s_sig[j];
        for (int v = 0; v < n; ++v) rk += (s_sig[v] > sj_) || (s_sig[v] == sj_ && v < j);
        s_keep[rk] = (unsigned char)j; s_rank[j] = (unsigned char)rk;                // columns by decreasing singular value
    }
    __syncthreads();
    // the consumer (gate_finish) ranks the columns again, by their f32 norms: everything within 1e-4 of the cap-th singular value is formed as well, so that a tie
    // at the cap -- the equal pseudo-values of collapsed pivots, or a degenerate pair -- can never make it pick a column that was not formed
    if (tid == 0) { int k = cap; const double thr = s_sig[s_keep[cap - 1]] * (1.0 - 1e-4); while (k < n && s_sig[s_keep[k]] >= thr) ++k; s_nk = k; }
    __syncthreads();
    const int nk = s_nk;
    for (int j = tid; j < n; j += NT)
        if ((int)s_rank[j] >= nk) for (int i = 0; i < m; ++i) Ag[i + (size_t)m * j] = cmake<float>(i == 0 ? (float)(s_sig[j] * sc_out) : 0.f, 0.f);
    const bool fullk = full16 && !(nk & 15);
    {
        // U Sigma = A U_L on the f64 matrix cores, computed TRANSPOSED (tile rows = kept column c of the result, lanes = row i: stores run along i).  A wave keeps
        // its (at most four) tiles in registers until the column norms are complete: the columns leave with the norm the sweeps found for them (s_sig) -- A u_j
        // carries an error of eps sigma_max in norm and direction like any product in working precision, the singular VALUE does not have to
        double* s_on = s_piv;                                        // column norms^2 of A X_final
        for (int j = tid; j < n; j += NT) s_on[j] = 0.0;
        __syncthreads();
        const int tr = (nk + 15) >> 4, tc = (m + 15) >> 4;           // tr * tc <= 4 * 8 = 32 tiles, at most four per wave
        v4d_t acr[4], aci[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = w + nw * q;
            acr[q] = v4d_t{0, 0, 0, 0}; aci[q] = v4d_t{0, 0, 0, 0};
            if (t < tr * tc) {
                const int c0 = 16 * (t % tr), i0 = 16 * (t / tr);
                const int jl = (int)s_keep[c0 + l15 < nk ? c0 + l15 : 0];              // this lane's column of X (A operand)
                if (fullk) tile_mm_f32<false>(X + xp * jl, 1, Mf + (i0 + l15), mp, n, acr[q], aci[q]);      // out[i][j] = sum_k A[i][k] X[k][j] (sorted columns of A)
                else ztile_mm(n, c0 + l15, i0 + l15,
                         [&](int c, int k) { cx<double> v = cmake<double>(0, 0); if (c < nk && k < n) { const cx<float> a = X[k + xp * jl]; v = cmake<double>(a.re, a.im); } return v; },
                         [&](int k, int i) { cx<double> v = cmake<double>(0, 0); if (i < m && k < n) { const cx<float> a = Mf[i + mp * k]; v = cmake<double>(a.re, a.im); } return v; },
                         acr[q], aci[q]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double p2 = acr[q][r] * acr[q][r] + aci[q][r] * aci[q][r];            // row c = c0 + kq + 4 r of the tile, column i = i0 + l15
                    if (i0 + l15 >= m) p2 = 0;
                    p2 = row16_sum(p2);
                    if (l15 == 0 && c0 + kq + 4 * r < nk) atomicAdd(&s_on[s_keep[c0 + kq + 4 * r]], p2);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = w + nw * q;
            if (t < tr * tc) {
                const int c0 = 16 * (t % tr), i0 = 16 * (t / tr);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = c0 + kq + 4 * r, i = i0 + l15;
                    if (c < nk && i < m) {
                        const int j = s_keep[c];
                        const double f = s_on[j] > 0 ? s_sig[j] / sqrt(s_on[j]) * sc_out : 0.0;
                        Ag[i + (size_t)m * j] = cmake<float>((float)(acr[q][r] * f), (float)(aci[q][r] * f));
                    }
                }
            }
        }
    }
    PRE_STAMP(5);
    // ---- 7. right singular vectors, kept columns only.  Low-rank theta = A Q^T (Q = B L^-dagger, (r2 d2) x n, f64; written by lowrank_m_kernel):
    // V = conj(Q) U_L on the f64 matrix cores.  A theta factorised as it stands (QB null): V = U_L itself, rows back in the original column order.  U_L is
    // orthonormal to f32 rounding whatever the spectrum, so V needs no division by Sigma^2 -- the recovery from the unrotated theta this replaces amplified the
    // error of a column of U Sigma by (sigma_max / sigma_j)^2 and therefore needed U Sigma orthogonal relative to each column's own norm ---------------------
    const bool lowrank = it.QB && (!it.dyn || it.dyn[7] > 0);          // (an item offered with its Q whose low-rank route was withdrawn on the device is theta itself)
    if (it.Vout && !lowrank) {
        cx<float>* Vg = reinterpret_cast<cx<float>*>(it.Vout);
        for (int e = tid; e < n * (n - nk); e += NT) Vg[(e % n) + (size_t)n * (int)s_keep[nk + e / n]] = cmake<float>(0.f, 0.f);      // columns that were not formed: zero, never garbage
        for (int e = tid; e < n * nk; e += NT) {
            const int k = e % n, j = s_keep[e / n];
            const cx<float> v = X[k + xp * j]; const float f = (float)s_cn[j];
            Vg[(int)s_perm[k] + (size_t)n * j] = cmake<float>(v.re * f, v.im * f);
        }
    } else if (it.Vout && it.dyn) {
        int mq, nq, kq_; theta_dims(it.dyn, it.dm, it.dn, mq, nq, kq_);      // nq = r2 d2: rows of Q and of V
        (void)mq; (void)kq_;
        const cx<double>* Q = reinterpret_cast<const cx<double>*>(it.QB);
        cx<float>* Vg = reinterpret_cast<cx<float>*>(it.Vout);
        const int tr = (nk + 15) >> 4, tc = (nq + 15) >> 4;
        for (int e = tid; e < nq * (n - nk); e += NT) Vg[(e % nq) + (size_t)nq * (int)s_keep[nk + e / nq]] = cmake<float>(0.f, 0.f);      // columns that were not formed: zero, never garbage
        for (int t = w; t < tr * tc; t += nw) {
            const int c0 = 16 * (t % tr), i0 = 16 * (t / tr);
            const int jl = (int)s_keep[c0 + l15 < nk ? c0 + l15 : 0];
            v4d_t cr = {0, 0, 0, 0}, ci = {0, 0, 0, 0};
            ztile_mm(n, c0 + l15, i0 + l15,                                              // V[i][j] = sum_k conj(Q[i][perm k]) X[k][j] / |x_j|
                     [&](int c, int k) { cx<double> v = cmake<double>(0, 0); if (c < nk && k < n) { const cx<float> a = X[k + xp * jl]; v = cmake<double>(a.re, a.im); } return v; },
                     [&](int k, int i) { cx<double> v = cmake<double>(0, 0); if (i < nq && k < n) { v = Q[i + (size_t)nq * (int)s_perm[k]]; v.im = -v.im; } return v; }, cr, ci);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = c0 + kq + 4 * r, i = i0 + l15;
                if (c < nk && i < nq) { const int j = s_keep[c]; const double f = s_cn[j]; Vg[i + (size_t)nq * j] = cmake<float>((float)(cr[r] * f), (float)(ci[r] * f)); }
            }
        }
    }
    PRE_STAMP(6);
#undef PRE_STAMP
    if (tid == 0 && it.sweeps_out) *it.sweeps_out = sweep;
}
void launch_theta_svd_pre(hipStream_t s, const JacobiItem* d_items, int nitems, int max_sweeps, int mmax, int nmax) {
    if (nitems <= 0) return;
    const size_t lds = theta_svd_pre_lds_bytes(mmax, nmax);
    set_max_dynamic_lds((const void*)theta_svd_pre_kernel<512>, (size_t)(160 * 1024 - 4096));
    hipLaunchKernelGGL((theta_svd_pre_kernel<512>), dim3(nitems), dim3(512), lds, s, d_items, max_sweeps); TNQS_CHECK_LAUNCH();
}

// V[:,u] = A0^dagger a_u / |a_u|^2   (a_u = column u of U Sigma), for the factorisations run without accumulating V.
// grid (item, column block of 8): a wave owns one output column u and keeps a_u in registers (lanes = rows, coalesced);
// every V[col, u] is one coalesced column read of A0 and a wave reduction.
template <class T, int R>                  // m <= 64 R rows
__global__ __launch_bounds__(512) void recover_v_kernel(const RecoverItem* __restrict__ items) {
    const RecoverItem it = items[blockIdx.x];
    if (it.pre && theta_pre_takes(it.dyn, it.dm, it.dn, it.pre == 2)) return;      // V already written by theta_svd_pre_kernel
    const cx<T>* A0 = reinterpret_cast<const cx<T>*>(it.A0);
    const cx<T>* A = reinterpret_cast<const cx<T>*>(it.A);
    cx<T>* V = reinterpret_cast<cx<T>*>(it.V);
    int m_ = it.m, n_ = it.n, nu_ = it.nu;
    if (it.dyn) theta_dims(it.dyn, it.dm, it.dn, m_, n_, nu_);
    const int m = m_, n = n_, nu = nu_;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int u = blockIdx.y * 8 + w;
    if (u >= nu) return;
    double are[R], aim[R], s2 = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        int i = lane + 64 * r;
        cx<T> a = (i < m) ? A[i + (size_t)m * u] : cmake<T>((T)0, (T)0);
        are[r] = a.re; aim[r] = a.im; s2 += are[r] * are[r] + aim[r] * aim[r];
    }
    s2 = wave_sum(s2);
    const double inv = s2 > 0 ? 1.0 / s2 : 0.0;
    constexpr int CU = 8;                    // columns in flight per iteration (loads of 8 columns overlap the reductions)
    for (int col0 = 0; col0 < n; col0 += CU) {
        double re[CU], im[CU];
#pragma unroll
        for (int c = 0; c < CU; ++c) {
            re[c] = 0; im[c] = 0;
            const int col = col0 + c;
            if (col < n) {
                const cx<T>* b0 = A0 + (size_t)m * col;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    int i = lane + 64 * r;
                    if (i < m) { cx<T> b = b0[i]; re[c] += (double)b.re * are[r] + (double)b.im * aim[r]; im[c] += (double)b.re * aim[r] - (double)b.im * are[r]; }   // conj(b) * a
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CU; ++c) { re[c] = wave_sum(re[c]); im[c] = wave_sum(im[c]); }
        if (lane < CU && col0 + lane < n) {
            double rr = 0, ii = 0;
#pragma unroll
            for (int c = 0; c < CU; ++c) if (lane == c) { rr = re[c]; ii = im[c]; }
            V[col0 + lane + (size_t)n * u] = cmake<T>((T)(rr * inv), (T)(ii * inv));
        }
    }
}
template <class T> void launch_recover_v(hipStream_t s, const RecoverItem* d_items, int nitems, int nmax) {
    if (nitems <= 0) return;
    // (rows: at most 512 = theta of d^2 chi <= 512; R = 8 costs registers only when it is needed, and the caller cannot know m per item here)
    hipLaunchKernelGGL((recover_v_kernel<T, 8>), dim3(nitems, (nmax + 7) / 8), dim3(512), 0, s, d_items); TNQS_CHECK_LAUNCH();
}
template void launch_recover_v<float>(hipStream_t, const RecoverItem*, int, int);
template void launch_recover_v<double>(hipStream_t, const RecoverItem*, int, int);

// lds_bytes: max over the items of (m*n + (V ? n*n : 0)) * sizeof(complex<T>); 0 selects the global-memory kernel
// A quarter wave rotates one column pair, so a round of an n-column matrix keeps n / 8 waves busy; waves beyond that only add to every
// barrier of the sweep (and a 1024-thread workgroup per 32 x 32 message matrix left three quarters of each CU's wave slots idling at
// barriers: 1140 such matrices per colour batch).  The workgroup is sized for the columns the matrices are expected to have (`ncols`;
// more columns than that still work: the slots loop).
template <class T, int RQ> static void launch_jacobi_lds(hipStream_t s, const JacobiItem* d_items, int nitems, int max_sweeps, size_t lds_bytes, int ncols) {
    set_max_dynamic_lds((const void*)jacobi_lds_kernel<T, RQ>, (size_t)(160 * 1024 - 2048));
    int waves = (ncols + 7) / 8; waves = waves < 4 ? 4 : (waves > 16 ? 16 : waves);
    hipLaunchKernelGGL((jacobi_lds_kernel<T, RQ>), dim3(nitems), dim3(64 * waves), lds_bytes, s, d_items, max_sweeps); TNQS_CHECK_LAUNCH();
}
// mmax: largest row count among the items (selects the rows-per-lane instantiation); ncols: expected column count (0: mmax)
template <class T> void launch_jacobi(hipStream_t s, const JacobiItem* d_items, int nitems, int max_sweeps, size_t lds_bytes, int mmax, int ncols) {
    if (nitems <= 0) return;
    if (ncols <= 0) ncols = mmax;
    if (lds_bytes > 0 && lds_bytes <= 160 * 1024 - 2048 && mmax <= 256) {
        if (mmax <= 32) launch_jacobi_lds<T, 2>(s, d_items, nitems, max_sweeps, lds_bytes, ncols);
        else if (mmax <= 64) launch_jacobi_lds<T, 4>(s, d_items, nitems, max_sweeps, lds_bytes, ncols);
        else if (mmax <= 96) launch_jacobi_lds<T, 6>(s, d_items, nitems, max_sweeps, lds_bytes, ncols);
        else if (mmax <= 128) launch_jacobi_lds<T, 8>(s, d_items, nitems, max_sweeps, lds_bytes, ncols);
        else launch_jacobi_lds<T, 16>(s, d_items, nitems, max_sweeps, lds_bytes, ncols);
    } else {
        if (mmax > 512) throw std::runtime_error("launch_jacobi: more than 512 rows");
        if (mmax <= 256) { hipLaunchKernelGGL((jacobi_kernel<T, 4>), dim3(nitems), dim3(1024), 0, s, d_items, max_sweeps); }
        else { hipLaunchKernelGGL((jacobi_kernel<T, 8>), dim3(nitems), dim3(1024), 0, s, d_items, max_sweeps); }
        TNQS_CHECK_LAUNCH();
    }
}
template void launch_jacobi<float>(hipStream_t, const JacobiItem*, int, int, size_t, int, int);
template void launch_jacobi<double>(hipStream_t, const JacobiItem*, int, int, size_t, int, int);

// ------------------------------------------------------------------------------------------------------------
// small sites (N < n): matricise psi~ to f64, and turn the rotated columns (U Sigma) into the (A, V) pair gate_eigs reads
// ------------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void small_svd_prepare_kernel(const SmallSvdItem* __restrict__ items) {
    const SmallSvdItem it = items[blockIdx.x];
    const cx<T>* src = reinterpret_cast<const cx<T>*>(it.src);
    cx<double>* M = reinterpret_cast<cx<double>*>(it.M);
    const int n = it.d * it.chi_b; const size_t tot = (size_t)n * it.low * it.hi;
    for (size_t e = threadIdx.x; e < tot; e += 256) {
        int s = (int)(e % it.d); size_t r = e / it.d; int lo = (int)(r % it.low); size_t r2 = r / it.low; int ib = (int)(r2 % it.chi_b); int hi = (int)(r2 / it.chi_b);
        cx<T> v = src[e];
        M[(s + it.d * ib) + (size_t)n * (lo + (size_t)it.low * hi)] = cmake<double>((double)v.re, -(double)v.im);     // M = Psi^dagger: G = M M^dagger, eigenvectors = left singular vectors
    }
}
template <class T> void launch_small_svd_prepare(hipStream_t s, const SmallSvdItem* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL((small_svd_prepare_kernel<T>), dim3(nitems), dim3(256), 0, s, d_items); TNQS_CHECK_LAUNCH();
}
template void launch_small_svd_prepare<float>(hipStream_t, const SmallSvdItem*, int);
template void launch_small_svd_prepare<double>(hipStream_t, const SmallSvdItem*, int);
// column j of M J = sigma_j u_j:  V[:,j] = u_j, A[:,j] = sigma_j^2 u_j (so that Re(v_j^dagger a_j) = sigma_j^2 = the eigenvalue of G)
__global__ __launch_bounds__(256) void small_svd_finish_kernel(const SmallSvdItem* __restrict__ items) {
    const SmallSvdItem it = items[blockIdx.x];
    const cx<double>* M = reinterpret_cast<const cx<double>*>(it.M);
    cx<double>* A = reinterpret_cast<cx<double>*>(it.GA);
    cx<double>* V = reinterpret_cast<cx<double>*>(it.GV);
    const int n = it.d * it.chi_b, N = it.low * it.hi;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int j = w; j < n; j += 4) {
        double s2 = 0;
        if (j < N) for (int i = lane; i < n; i += 64) { cx<double> v = M[i + (size_t)n * j]; s2 += v.re * v.re + v.im * v.im; }
        s2 = wave_sum(s2);
        const double sg = sqrt(s2), inv = sg > 0 ? 1.0 / sg : 0.0;
        for (int i = lane; i < n; i += 64) {
            cx<double> v = (j < N) ? M[i + (size_t)n * j] : cmake<double>(0, 0);
            V[i + (size_t)n * j] = cmake<double>(v.re * inv, v.im * inv);
            A[i + (size_t)n * j] = cmake<double>(v.re * sg, v.im * sg);
        }
    }
}
void launch_small_svd_finish(hipStream_t s, const SmallSvdItem* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL(small_svd_finish_kernel, dim3(nitems), dim3(256), 0, s, d_items); TNQS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// Cholesky factor of the Gram matrix (the R factor of the thin QR, simple_update.jl:45-48, when G has full rank)
// ------------------------------------------------------------------------------------------------------------
// Right-looking, ONE workgroup barrier per column: the trailing update of step k works from the UNSCALED column k,
//   A[i][j] -= A[i][k] conj(A[j][k]) / A[k][k]     (columns j > k; column k itself is never written again),
// every thread derives the pivot from A[k][k] by the same rule, and the scaling L[i][k] = A[i][k] / sqrt(A[k][k]) happens for all columns
// at once at the end.  With Lt = unit lower triangular, Lt[i][k] = A[i][k] / A[k][k], this is G = Lt D Lt^dagger, L = Lt D^1/2.
// The INVERSE rides along in the same steps (round 3): M = Lt^-1 is what the same row operations make of the identity,
//   M[i][c] -= Lt[i][k] M[k][c]     (rows i > k, columns c <= k, M[k][k] = 1),
// kept in the free strict upper triangle (M[i][c] at A[c + np i]); L^-1 = D^-1/2 M.  No separate substitution phase, no extra barrier.
// Within a step every element update is independent: a thread takes elements e = tid + 256 u of the trailing triangle (row-major
// triangular numbering, which is NESTED: the first m (m + 1) / 2 numbers are the triangle of size m, so a thread's (row, column) pairs
// are decoded once for the whole factorisation) and of the (rows > k) x (columns <= k) rectangle, eight at a time with all their LDS
// loads issued before the first store -- the column step is then one LDS round trip deep instead of one per element.
// (History: a first version scaled the column between two extra barriers per step and inverted L with one serial thread per column:
// 214 us per launch on a 64 x 64 matrix; the one-barrier version with a per-thread loop over columns and a 4-lane substitution for the
// inverse: 122 us = 10 load + 64 column loop (1 us per column, a chain of LDS latencies) + 45 inverse.)
template <int NT, int UT>      // threads, and the trailing-triangle elements a thread owns at most (UT a multiple of 3; NT * UT >= 96 * 97 / 2)
__global__ __launch_bounds__(NT) void chol_kernel(const CholItem* __restrict__ items) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double s_dmax;
    const CholItem it = items[blockIdx.x];
    // (NT = 1024, sixteen waves: the other waves of a SIMD cover a wave's LDS round trips)
    const int n = it.n, np = n + 1, tid = threadIdx.x;
    cx<double>* A = reinterpret_cast<cx<double>*>(smem);          // [col j][row i] at i + np*j, lower triangle becomes L (unscaled), strict upper M
    const cx<double>* G = reinterpret_cast<const cx<double>*>(it.G);
    for (int e = tid; e < n * n; e += NT) {                       // Hermitian part, as the eigen path sees it; zeros above the diagonal
        int i = e % n, j = e / n;
        cx<double> v = cmake<double>(0, 0);
        if (i >= j) { cx<double> a = G[i + (size_t)n * j], b = G[j + (size_t)n * i]; v = cmake<double>(0.5 * (a.re + b.re), 0.5 * (a.im - b.im)); }
        A[i + np * j] = v;
    }
    __syncthreads();
    if (tid < 64) {                                                // largest diagonal entry (one wave)
        double m = 0; for (int i = tid; i < n; i += 64) m = fmax(m, A[i + np * i].re);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
        if (tid == 0) s_dmax = m;
    }
    __syncthreads();
    const double tiny = it.tau * s_dmax;
    // the pivot rule: a pivot at or below tiny (or not a number) flags the item and is replaced, so that the factorisation completes
    auto pivot_of = [&](int k, bool& bad) { double d = A[k + np * k].re; bad = !(d > tiny); return bad ? (tiny > 0 ? tiny : 1.0) : d; };
    // this thread's elements of the trailing triangle: number e = r (r + 1) / 2 + c, 0 <= c <= r  (n <= 96: at most 4560 / 1024 -> 5)
    unsigned char tr[UT], tc[UT];
#pragma unroll
    for (int u = 0; u < UT; ++u) {
        const int e = tid + NT * u;
        int r = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
        while (r * (r + 1) / 2 > e) --r;
        while ((r + 1) * (r + 2) / 2 <= e) ++r;
        tr[u] = (unsigned char)r; tc[u] = (unsigned char)(e - r * (r + 1) / 2);
    }
    const bool wantW = it.Winv != nullptr;
    for (int k = 0; k < n; ++k) {
        bool bad; const double d = pivot_of(k, bad);
        if (bad && tid == 0) *it.fail = 1;
        const double dinv = 1.0 / d;
        const int m = n - k - 1, k1 = k + 1;
        const cx<double>* colk = A + np * k;                       // colk[i] = A[i][k]
        // ---- trailing triangle: A[i][j] -= (A[i][k] / d) conj(A[j][k]),  i = k1 + r,  j = k1 + c ----------------------------------------
        const int nt = m * (m + 1) / 2;
#pragma unroll
        for (int u0 = 0; u0 < UT; u0 += 3) {
            if (NT * u0 >= nt) break;                               // (workgroup-uniform)
            cx<double> li[3], lj[3], v[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (u0 + u < UT && tid + NT * (u0 + u) < nt) { const int i = k1 + tr[u0 + u], j = k1 + tc[u0 + u]; li[u] = colk[i]; lj[u] = colk[j]; v[u] = A[i + np * j]; }
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (u0 + u < UT && tid + NT * (u0 + u) < nt) {
                    const double sr = li[u].re * dinv, si = li[u].im * dinv;
                    v[u].re -= sr * lj[u].re + si * lj[u].im; v[u].im -= si * lj[u].re - sr * lj[u].im;
                    A[(k1 + tr[u0 + u]) + np * (k1 + tc[u0 + u])] = v[u];
                }
            }
        }
        // ---- inverse: M[i][c] -= (A[i][k] / d) M[k][c],  i = k1 + ri,  c <= k;  M[i][c] at A[c + np i], M[k][k] = 1 ------------------------
        if (wantW) {
            const int nr = m * k1;
            const float rk1 = 1.0f / (float)k1;
            for (int q0 = 0; q0 < nr; q0 += NT * 2) {
                cx<double> li[2], mk[2], v[2]; int ii[2], cc[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int q = q0 + tid + NT * u;
                    int ri = (int)((float)q * rk1); if (ri * k1 > q) --ri; if ((ri + 1) * k1 <= q) ++ri;
                    ii[u] = k1 + ri; cc[u] = q - ri * k1;
                    if (q < nr) {
                        li[u] = colk[ii[u]];
                        mk[u] = cc[u] == k ? cmake<double>(1.0, 0.0) : A[cc[u] + np * k];
                        v[u] = A[cc[u] + np * ii[u]];
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int q = q0 + tid + NT * u;
                    if (q < nr) {
                        const double sr = li[u].re * dinv, si = li[u].im * dinv;
                        v[u].re -= sr * mk[u].re - si * mk[u].im; v[u].im -= sr * mk[u].im + si * mk[u].re;
                        A[cc[u] + np * ii[u]] = v[u];
                    }
                }
            }
        }
        __syncthreads();
    }
    // L[i][k] = A[i][k] / sqrt(pivot_k), L[k][k] = sqrt(pivot_k);  W = (L^-1)^dagger: W[i + n a] = conj(M[a][i]) / sqrt(pivot_a) above the diagonal
    __shared__ double s_piv[96];
    for (int k = tid; k < n; k += NT) { bool bad; s_piv[k] = sqrt(pivot_of(k, bad)); }
    __syncthreads();
    cx<double>* L = reinterpret_cast<cx<double>*>(it.L);
    cx<double>* W = reinterpret_cast<cx<double>*>(it.Winv);
    for (int e = tid; e < n * n; e += NT) {
        const int i = e % n, k = e / n;
        const cx<double> a = A[i + np * k];
        const double r = 1.0 / s_piv[k];
        cx<double> l = cmake<double>(0, 0), w = cmake<double>(0, 0);
        if (i == k) { l = cmake<double>(s_piv[k], 0.0); w = cmake<double>(r, 0.0); }
        else if (i > k) l = cmake<double>(a.re * r, a.im * r);
        else w = cmake<double>(a.re * r, -a.im * r);
        L[e] = l;
        if (W) W[e] = w;
    }
}
// Same factorisation with the lower triangle PACKED in LDS (column j holds rows j..n-1): n up to 128 fits (132 KB), which the low-rank theta
// route needs at chi = 64 (K = kappa chi = 128).  Only L is produced (CholItem::Winv is not written: the packed layout has no spare triangle
// for the inverse: when CholItem::Winv is given, (L^-1)^dagger is built in place in global memory by a second phase).
__global__ __launch_bounds__(1024) void chol_packed_kernel(const CholItem* __restrict__ items) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double s_dmax;
    const CholItem it = items[blockIdx.x];
    constexpr int NT = 1024;                                         // sixteen waves: a thread's column loop is at most 8 long, and the other waves of its SIMD cover its LDS round trips
    const int n = it.n, tid = threadIdx.x;
    cx<double>* A = reinterpret_cast<cx<double>*>(smem);
    auto at = [n](int i, int j) { return (size_t)j * n - (size_t)j * (j - 1) / 2 + (i - j); };      // i >= j
    const cx<double>* G = reinterpret_cast<const cx<double>*>(it.G);
    for (int e = tid; e < n * n; e += NT) {
        int i = e % n, j = e / n; if (i < j) continue;
        cx<double> a = G[i + (size_t)n * j], b = G[j + (size_t)n * i];
        A[at(i, j)] = cmake<double>(0.5 * (a.re + b.re), 0.5 * (a.im - b.im));
    }
    __syncthreads();
    if (tid < 64) {
        double m = 0; for (int i = tid; i < n; i += 64) m = fmax(m, A[at(i, i)].re);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
        if (tid == 0) s_dmax = m;
    }
    __syncthreads();
    if (it.shift > 0) { for (int i = tid; i < n; i += NT) A[at(i, i)].re += it.shift * s_dmax; __syncthreads(); }
    const double tiny = it.tau * s_dmax;
    // right-looking with ONE barrier per column, as chol_kernel: trailing updates from the unscaled column, pivots by the same rule in every
    // thread, all columns scaled at the end
    auto pivot_of = [&](int k, bool& bad) { double d = A[at(k, k)].re; bad = !(d > tiny); return bad ? (tiny > 0 ? tiny : 1.0) : d; };
    const int ti = tid & 63, tj = tid >> 6;
    for (int k = 0; k < n; ++k) {
        bool bad; const double d = pivot_of(k, bad);
        if (bad && tid == 0) *it.fail = 1;
        const double dinv = 1.0 / d;
        for (int i = k + 1 + ti; i < n; i += 64) {
            const cx<double> li = A[at(i, k)];
            const cx<double> ls = cmake<double>(li.re * dinv, li.im * dinv);
            int j = k + 1 + tj;
            for (; j <= i; j += NT / 64) {
                const cx<double> lj = A[at(j, k)];
                cx<double> v = A[at(i, j)];
                v.re -= ls.re * lj.re + ls.im * lj.im; v.im -= ls.im * lj.re - ls.re * lj.im;
                A[at(i, j)] = v;
            }
        }
        __syncthreads();
    }
    __shared__ double s_pivs[128];
    for (int k = tid; k < n; k += NT) { bool bad; s_pivs[k] = sqrt(pivot_of(k, bad)); }
    __syncthreads();
    for (int e = tid; e < n * n; e += NT) {
        const int i = e % n, k = e / n;
        if (i < k) continue;
        if (i == k) A[at(k, k)] = cmake<double>(s_pivs[k], 0.0);
        else { const cx<double> v = A[at(i, k)]; const double r = 1.0 / s_pivs[k]; A[at(i, k)] = cmake<double>(v.re * r, v.im * r); }
    }
    __syncthreads();
    cx<double>* L = reinterpret_cast<cx<double>*>(it.L);
    for (int e = tid; e < n * n; e += NT) { int i = e % n, j = e / n; L[e] = (i >= j) ? A[at(i, j)] : cmake<double>(0, 0); }
    if (!it.Winv) return;
    // W = (L^-1)^dagger (upper triangular), W[c + n*i] = conj(Linv[i, c]).  L^-1 is built IN PLACE in the packed triangle, from the last column
    // to the first: column j of the inverse is  -Linv[j+1:, j+1:] L[j+1:, j] / L[j, j]  -- the trailing block is already inverted, column j still
    // holds L.  Eight threads per row i split the sum over k (one LDS read of Linv[i, k], consecutive in i, and one broadcast read of L[k, j] per
    // term); two barriers per column.  (Round 2 ran one thread per column of the inverse against global memory: 0.6 of the kernel's 0.78 ms
    // at n = 128.)  L itself has been written out above.
    __syncthreads();
    cx<double>* W = reinterpret_cast<cx<double>*>(it.Winv);
    const int row = tid >> 3, half = tid & 7;                     // eight threads per row split the sum over k
    for (int j = n - 1; j >= 0; --j) {
        const double dj = 1.0 / A[at(j, j)].re;
        const int i = j + 1 + row;
        double ar = 0, ai = 0;
        if (i < n) {
            int k = j + 1 + half;
            for (; k + 24 <= i; k += 32) {                       // four independent terms in flight
                cx<double> x[4], l[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { x[u] = A[at(i, k + 8 * u)]; l[u] = A[at(k + 8 * u, j)]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) { ar -= x[u].re * l[u].re - x[u].im * l[u].im; ai -= x[u].re * l[u].im + x[u].im * l[u].re; }
            }
            for (; k <= i; k += 8) {
                const cx<double> x = A[at(i, k)], l = A[at(k, j)];
                ar -= x.re * l.re - x.im * l.im; ai -= x.re * l.im + x.im * l.re;
            }
        }
        ar += __shfl_xor(ar, 1, 64); ai += __shfl_xor(ai, 1, 64);
        ar += __shfl_xor(ar, 2, 64); ai += __shfl_xor(ai, 2, 64);
        ar += __shfl_xor(ar, 4, 64); ai += __shfl_xor(ai, 4, 64);
        __syncthreads();                                   // column j has been read by everybody
        if (i < n && half == 0) A[at(i, j)] = cmake<double>(ar * dj, ai * dj);
        if (tid == 0) A[at(j, j)] = cmake<double>(dj, 0.0);
        __syncthreads();
    }
    for (int e = tid; e < n * n; e += NT) {
        const int c = e % n, i = e / n;
        cx<double> v = cmake<double>(0, 0);
        if (i >= c) { v = A[at(i, c)]; v.im = -v.im; }
        W[e] = v;
    }
}
void launch_chol_packed(hipStream_t s, const CholItem* d_items, int nitems, int nmax) {
    if (nitems <= 0) return;
    const size_t lds = (size_t)nmax * (nmax + 1) / 2 * 16;
    set_max_dynamic_lds((const void*)chol_packed_kernel, (size_t)(160 * 1024 - 2048));      // (+ ~1 KB of static LDS: pivots)
    hipLaunchKernelGGL(chol_packed_kernel, dim3(nitems), dim3(1024), lds, s, d_items); TNQS_CHECK_LAUNCH();
}
void launch_chol(hipStream_t s, const CholItem* d_items, int nitems, int nmax) {
    if (nitems <= 0) return;
    const size_t lds = (size_t)nmax * (nmax + 1) * 16;
    set_max_dynamic_lds((const void*)chol_kernel<1024, 6>, (size_t)(160 * 1024 - 1024));       // (the kernel also has ~0.8 KB of static LDS: pivots)
    hipLaunchKernelGGL((chol_kernel<1024, 6>), dim3(nitems), dim3(1024), lds, s, d_items); TNQS_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------------------
// environment square roots  (src/utils.jl:18-27 with safe_eigen :94-108: always f64)
// ------------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void env_prepare_kernel(const EnvItem* __restrict__ items) {
    const EnvItem it = items[blockIdx.x];
    const int n = it.n;
    const cx<T>* M = reinterpret_cast<const cx<T>*>(it.msg);
    cx<double>* H = reinterpret_cast<cx<double>*>(it.H);
    cx<double>* V = reinterpret_cast<cx<double>*>(it.V);
    for (int e = threadIdx.x; e < n * n; e += 256) {
        int i = e % n, j = e / n;
        double re, im;
        if (M) {
            cx<T> a = M[i + n * j], b = M[j + n * i];
            re = 0.5 * ((double)a.re + (double)b.re); im = 0.5 * ((double)a.im - (double)b.im);
        } else { re = (i == j) ? 1.0 : 0.0; im = 0; }
        H[e] = cmake<double>(re, im);
        V[e] = cmake<double>(i == j ? 1.0 : 0.0, 0.0);
    }
}
template <class T> void launch_env_prepare(hipStream_t s, const EnvItem* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL((env_prepare_kernel<T>), dim3(nitems), dim3(256), 0, s, d_items); TNQS_CHECK_LAUNCH();
}
template void launch_env_prepare<float>(hipStream_t, const EnvItem*, int);
template void launch_env_prepare<double>(hipStream_t, const EnvItem*, int);

template <class T>
__global__ __launch_bounds__(256) void env_finish_kernel(const EnvFinishItem* __restrict__ items) {
    __shared__ double lam[256], sq[256];
    __shared__ int s_full, s_err;
    const EnvFinishItem it = items[blockIdx.x];
    const int n = it.n;
    const cx<double>* A = reinterpret_cast<const cx<double>*>(it.A);
    const cx<double>* V = reinterpret_cast<const cx<double>*>(it.V);
    if (threadIdx.x == 0) { s_full = 1; s_err = 0; }
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += 256) {
        double l = 0;       // Rayleigh quotient v_j^dagger H v_j = Re(v_j^dagger a_j)
        for (int i = 0; i < n; ++i) { cx<double> v = V[i + n * j], a = A[i + n * j]; l += v.re * a.re + v.im * a.im; }
        lam[j] = l;
        // (sq[j]: sqrt(lambda_j) of the eigenvalues that are kept, -1 for the dropped ones -- once per eigenvalue instead of once per term below)
        // the reference casts the eigenvalues back to the message precision BEFORE the cutoff test (safe_eigen, src/utils.jl:100-107, then
        // `abs(x) < cutoff` on the Float32 value, :21-22): an eigenvalue within an f32 ulp of the cutoff must land on the same side here
        const double lt = (double)(T)l;
        const bool zero = (lt == 0) || (fabs(lt) < it.cutoff);
        if (zero) s_full = 0;
        else if (lt < 0) s_err = 1;       // Julia: sqrt(negative) -> DomainError (src/utils.jl:21)
        sq[j] = (zero || lt < 0) ? -1.0 : sqrt(l);
    }
    __syncthreads();
    cx<T>* ms = reinterpret_cast<cx<T>*>(it.msqrt);
    cx<T>* pr = reinterpret_cast<cx<T>*>(it.proj);
    for (int e = threadIdx.x; e < n * n; e += 256) {
        int i = e % n, l = e / n;
        cx<double> s1 = cmake<double>(0, 0), s2 = cmake<double>(0, 0);
        for (int j = 0; j < n; ++j) {
            const double sj = sq[j];
            if (sj < 0) continue;
            cx<double> vi = V[i + n * j], vl = V[l + n * j];
            cx<double> o = cmake<double>(vi.re * vl.re + vi.im * vl.im, vi.im * vl.re - vi.re * vl.im);  // vi conj(vl)
            s1.re += sj * o.re; s1.im += sj * o.im;
            s2.re += o.re; s2.im += o.im;
        }
        ms[e] = cmake<T>((T)s1.re, (T)s1.im);
        pr[e] = cmake<T>((T)s2.re, (T)s2.im);
    }
    if (threadIdx.x == 0) { it.flags[0] = s_full; it.flags[1] = s_err; }
}
template <class T> void launch_env_finish(hipStream_t s, const EnvFinishItem* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL((env_finish_kernel<T>), dim3(nitems), dim3(256), 0, s, d_items); TNQS_CHECK_LAUNCH();
}
template void launch_env_finish<float>(hipStream_t, const EnvFinishItem*, int);
template void launch_env_finish<double>(hipStream_t, const EnvFinishItem*, int);

// ------------------------------------------------------------------------------------------------------------
// per-gate small algebra.  With G_i = psi~_i^dagger psi~_i = W L W^dagger:  R_i = L^{1/2} W^dagger (any
// orthogonal factorisation psi~ = Q R gives the same gauge-invariant result as the reference's QR).
// ------------------------------------------------------------------------------------------------------------

__device__ void gate_eigs(const cx<double>* A, const cx<double>* V, int n, double* lam_tmp /*LDS n*/, double* lam_out,
                          int* idx_out, int* r_out, int* s_r /*LDS*/, double tau) {
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        double l = 0;
        for (int i = 0; i < n; ++i) { cx<double> v = V[i + n * j], a = A[i + n * j]; l += v.re * a.re + v.im * a.im; }
        lam_tmp[j] = l;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double lmax = 0;
        for (int j = 0; j < n; ++j) lmax = fmax(lmax, lam_tmp[j]);
        int r = 0;
        if (tau < 0) {       // shifted first pass (a second factorisation pass follows): nothing is dropped, l := max(l, 0) + |tau| l_max
            for (int j = 0; j < n; ++j) { lam_out[j] = fmax(lam_tmp[j], 0.0) - tau * lmax; idx_out[j] = j; }
            r = n;
        } else
        for (int j = 0; j < n; ++j)
            if (lam_tmp[j] > tau * lmax && lam_tmp[j] > 0) { lam_out[r] = lam_tmp[j]; idx_out[r] = j; ++r; }
        *r_out = r; *s_r = r;
    }
    __syncthreads();
}

// Cholesky site: R = L^dagger is read through the same (eigenvector, eigenvalue) interface with lambda = 1, all columns kept
__device__ void gate_full_rank(int n, double* lam_out, int* idx_out, int* r_out, int* s_r, const int* rk = nullptr) {
    for (int j = threadIdx.x; j < n; j += blockDim.x) { lam_out[j] = 1.0; idx_out[j] = j; }
    if (threadIdx.x == 0) { const int r = rk ? *rk : n; *r_out = r; *s_r = r; }
    __syncthreads();
}
// is the kept part of the factor ill-conditioned (smallest / largest squared singular value of psi~ below 1e-4: the f64 Gram route alone leaves a relative error eps / that ratio)?  Such ComplexF64
// sites get a second factorisation pass (engine.cpp).  Cholesky: from the pivots diag(L)^2; eigen: from the kept eigenvalues.
__device__ int gate_ill_conditioned(int chol, const cx<double>* L, int n, const double* lam, int r) {
    double lo = 1e300, hi = 0;
    if (chol) for (int j = 0; j < n; ++j) { double d = L[j + (size_t)n * j].re; d *= d; lo = fmin(lo, d); hi = fmax(hi, d); }
    else for (int j = 0; j < r; ++j) { lo = fmin(lo, lam[j]); hi = fmax(hi, lam[j]); }
    return (hi > 0 && lo < 1e-4 * hi) ? 1 : 0;
}

template <class T>
__global__ __launch_bounds__(1024) void gate_theta_kernel(const GateItem* __restrict__ items) {
    __shared__ double lam_tmp[512];
    __shared__ int s_r1, s_r2;
    // grid (gate, part): every part repeats the small serial prologue (identical values) and takes a strided share of the element loops --
    // with one workgroup per gate the kernel was pure latency (0.56 ms per colour batch whatever the batch size)
    const GateItem it = items[blockIdx.x];
    const int part = blockIdx.y, tid0 = part * blockDim.x + threadIdx.x, tstride = gridDim.y * blockDim.x;
    const cx<double>* A1 = reinterpret_cast<const cx<double>*>(it.GA1);
    const cx<double>* V1 = reinterpret_cast<const cx<double>*>(it.GV1);
    const cx<double>* A2 = reinterpret_cast<const cx<double>*>(it.GA2);
    const cx<double>* V2 = reinterpret_cast<const cx<double>*>(it.GV2);
    if (it.chol1) gate_full_rank(it.n1, it.lam1, it.idx1, &it.info[0], &s_r1, it.chol1 == 2 ? it.rk1 : nullptr); else gate_eigs(A1, V1, it.n1, lam_tmp, it.lam1, it.idx1, &it.info[0], &s_r1, it.tau1);
    if (it.chol2) gate_full_rank(it.n2, it.lam2, it.idx2, &it.info[1], &s_r2, it.chol2 == 2 ? it.rk2 : nullptr); else gate_eigs(A2, V2, it.n2, lam_tmp, it.lam2, it.idx2, &it.info[1], &s_r2, it.tau2);
    if (threadIdx.x == 0 && part == 0) {
        it.info[6] = (it.chol1 == 2 ? 0 : gate_ill_conditioned(it.chol1, V1, it.n1, it.lam1, s_r1))
                   | ((it.chol2 == 2 ? 0 : gate_ill_conditioned(it.chol2, V2, it.n2, it.lam2, s_r2)) << 1);
    }
    const int r1 = s_r1, r2 = s_r2, d1 = it.d1, d2 = it.d2, chi = it.chi;
    const int Mr = r1 * d1, Nc = r2 * d2;
    const bool wide = Mr < Nc;
    cx<T>* th = reinterpret_cast<cx<T>*>(it.theta);
    cx<T>* tv = reinterpret_cast<cx<T>*>(it.thetaV);
    const cx<double>* g = reinterpret_cast<const cx<double>*>(it.gate);
    const int dd = d1 * d2;
    // theta[(a,s1'),(c,s2')] = sum_{s1,s2} g[(s1' s2'),(s1 s2)] sum_b R1[a,(s1,b)] R2[c,(s2,b)],  R_i[a,(s,b)] = sqrt(l_a) conj(W_i[(s,b),a])
    // With the gate as an operator sum (opA / opB, lowA / lowB given: every ComplexF32 batch) theta = A B^T is formed from the factors by
    // gate_theta_mm_kernel on the f64 matrix cores; the element-wise loop below (128 dependent, uncoalesced loads per entry: 285 us per
    // 190-gate batch) only serves states without the factorisation (ComplexF64)
    const bool via_factors = it.kappa > 0 && it.lowA && it.lowB;
    if (!via_factors)
    for (int e = tid0; e < Mr * Nc; e += tstride) {
        int row = e % Mr, col = e / Mr;
        int a = row % r1, s1p = row / r1, c = col % r2, s2p = col / r2;
        const cx<double>* w1 = V1 + (size_t)it.n1 * it.idx1[a];
        const cx<double>* w2 = V2 + (size_t)it.n2 * it.idx2[c];
        cx<double> acc = cmake<double>(0, 0);
        for (int s1 = 0; s1 < d1; ++s1)
            for (int s2 = 0; s2 < d2; ++s2) {
                cx<double> gg = g[(s1p * d2 + s2p) + dd * (s1 * d2 + s2)];
                if (gg.re == 0 && gg.im == 0) continue;
                cx<double> cc = cmake<double>(0, 0);
                for (int b = 0; b < chi; ++b) {
                    cx<double> x = w1[s1 + d1 * b], y = w2[s2 + d2 * b];
                    // conj(x) * conj(y)
                    cc.re += x.re * y.re - x.im * y.im;
                    cc.im -= x.re * y.im + x.im * y.re;
                }
                cfma(acc, gg, cc);
            }
        double sc = sqrt(it.lam1[a] * it.lam2[c]);
        // one-sided Jacobi needs rows >= columns: a wide theta is stored as theta^dagger (Nc x Mr)
        cx<T>* th0 = reinterpret_cast<cx<T>*>(it.theta0);
        if (!wide) { cx<T> v = cmake<T>((T)(acc.re * sc), (T)(acc.im * sc)); th[e] = v; if (th0) th0[e] = v; }
        else { cx<T> v = cmake<T>((T)(acc.re * sc), (T)(-acc.im * sc)); th[col + (size_t)Nc * row] = v; if (th0) th0[col + (size_t)Nc * row] = v; }
    }
    const int nI = wide ? Mr : Nc;
    for (int e = tid0; e < nI * nI; e += tstride) tv[e] = cmake<T>((e % nI) == (e / nI) ? (T)1 : (T)0, (T)0);
    // low-rank route (GateItem): A[(a,s1'),(k,b)] = sum_s1 a_k[s1',s1] R1[a,(s1,b)],  B[(c,s2'),(k,b)] = sum_s2 b_k[s2',s2] R2[c,(s2,b)],  G = B^dagger B
    const int K = it.kappa * chi;
    const bool low = it.kappa > 0 && it.lowG && !wide && K < Nc && it.chi_cap <= K;      // (the host only hands out lowG where the route may be taken)
    if (threadIdx.x == 0 && part == 0) { it.info[5] = wide ? 1 : 0; it.info[7] = low ? K : 0; }       // info[7]: lowrank_g / chol / lowrank_m follow
    if (via_factors) {
        cx<double>* LA = reinterpret_cast<cx<double>*>(it.lowA);
        cx<double>* LB = reinterpret_cast<cx<double>*>(it.lowB);
        const cx<double>* oa = reinterpret_cast<const cx<double>*>(it.opA);
        const cx<double>* ob = reinterpret_cast<const cx<double>*>(it.opB);
        for (int e = tid0; e < Mr * K; e += tstride) {
            const int row = e % Mr, l = e / Mr, a = row % r1, s1p = row / r1, b = l % chi, k = l / chi;
            const cx<double>* w1 = V1 + (size_t)it.n1 * it.idx1[a];
            cx<double> acc = cmake<double>(0, 0);
            for (int s1 = 0; s1 < d1; ++s1) { cx<double> x = w1[s1 + d1 * b]; cfma(acc, oa[k * d1 * d1 + s1p + d1 * s1], cmake<double>(x.re, -x.im)); }
            const double sc = sqrt(it.lam1[a]);
            LA[e] = cmake<double>(acc.re * sc, acc.im * sc);
        }
        for (int e = tid0; e < Nc * K; e += tstride) {
            const int row = e % Nc, l = e / Nc, c = row % r2, s2p = row / r2, b = l % chi, k = l / chi;
            const cx<double>* w2 = V2 + (size_t)it.n2 * it.idx2[c];
            cx<double> acc = cmake<double>(0, 0);
            for (int s2 = 0; s2 < d2; ++s2) { cx<double> y = w2[s2 + d2 * b]; cfma(acc, ob[k * d2 * d2 + s2p + d2 * s2], cmake<double>(y.re, -y.im)); }
            const double sc = sqrt(it.lam2[c]);
            LB[e] = cmake<double>(acc.re * sc, acc.im * sc);
        }
    }
    if (!low && it.lowG && it.kappa > 0) {      // low-rank SVD route not taken: give chol_kernel a harmless identity
        cx<double>* LG = reinterpret_cast<cx<double>*>(it.lowG);
        for (int e = tid0; e < K * K; e += tstride) LG[e] = cmake<double>((e % K) == (e / K) ? 1.0 : 0.0, 0.0);
    }
}
// ---- small complex f64 products on v_mfma_f64_16x16x4_f64: one wave per 16 x 16 tile  C[i][j] (+)= sum_k a(i, k) b(k, j) -------------------
// Operand layout of the instruction: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15], C[row = (lane >> 4) + 4 r][col = lane & 15].
// fa(i, k) / fb(k, j) return the operand (zero outside the matrix); four real products per complex step (these kernels are latency, not
// throughput: the gain over the scalar loops is that a tile takes 2 loads per 4 x 256 multiply-adds instead of 2 per multiply-add)
// theta = A B^T from the operator-sum factors gate_theta_kernel wrote (lowA: Mr x K, lowB: Nc x K, complex128), to theta and theta0 in
// the state's precision; a wide theta is stored as its adjoint.  The tile orientation is chosen so that the lanes run along the
// contiguous index of the destination.
template <class T>
__global__ __launch_bounds__(1024) void gate_theta_mm_kernel(const GateItem* __restrict__ items) {
    const GateItem it = items[blockIdx.x];
    if (!(it.kappa > 0 && it.lowA && it.lowB)) return;
    const int Mr = it.info[0] * it.d1, Nc = it.info[1] * it.d2, K = it.kappa * it.chi;
    const bool wide = it.info[5] != 0;
    const cx<double>* LA = reinterpret_cast<const cx<double>*>(it.lowA);
    const cx<double>* LB = reinterpret_cast<const cx<double>*>(it.lowB);
    cx<T>* th = reinterpret_cast<cx<T>*>(it.theta);
    cx<T>* th0 = reinterpret_cast<cx<T>*>(it.theta0);
    const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
    const int w = blockIdx.y * (blockDim.x >> 6) + (threadIdx.x >> 6), nw = gridDim.y * (blockDim.x >> 6);
    // rows of the tile product = the index that is NOT contiguous in the destination: (c, s2') for theta[i + Mr j], (a, s1') for the adjoint
    const int R = wide ? Mr : Nc, Cn = wide ? Nc : Mr;              // tile rows run over R, tile columns (lanes) over Cn
    const cx<double>* PR = wide ? LA : LB; const cx<double>* PC = wide ? LB : LA;
    const int tr = (R + 15) >> 4, tc = (Cn + 15) >> 4;
    for (int t = w; t < tr * tc; t += nw) {
        const int r0 = 16 * (t % tr), c0 = 16 * (t / tr);
        v4d_t cr = {0, 0, 0, 0}, ci = {0, 0, 0, 0};
        ztile_mm(K, r0 + l15, c0 + l15,
                 [&](int i, int k) { return (i < R && k < K) ? PR[i + (size_t)R * k] : cmake<double>(0, 0); },
                 [&](int k, int j) { return (j < Cn && k < K) ? PC[j + (size_t)Cn * k] : cmake<double>(0, 0); }, cr, ci);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = r0 + kq + 4 * r, col = c0 + l15;
            if (row < R && col < Cn) {
                // not wide: theta[i = col][j = row] at col + Mr * row;  wide: stored adjoint theta^dagger[j = col][i = row] at col + Nc * row, conjugated
                const cx<T> v = cmake<T>((T)cr[r], (T)(wide ? -ci[r] : ci[r]));
                th[col + (size_t)Cn * row] = v; if (th0) th0[col + (size_t)Cn * row] = v;
            }
        }
    }
}
template <class T> void launch_gate_theta_mm(hipStream_t s, const GateItem* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL((gate_theta_mm_kernel<T>), dim3(nitems, 2), dim3(1024), 0, s, d_items);
    TNQS_CHECK_LAUNCH();
}
template void launch_gate_theta_mm<float>(hipStream_t, const GateItem*, int);
template void launch_gate_theta_mm<double>(hipStream_t, const GateItem*, int);
// G = B^dagger B of the low-rank route (GateItem): upper 16 x 16 tiles on the f64 matrix cores, mirrored
__global__ __launch_bounds__(1024) void lowrank_g_kernel(const GateItem* __restrict__ items) {
    const GateItem it = items[blockIdx.x];
    const int K = it.info[7];
    if (K <= 0) return;
    const int Nc = it.info[1] * it.d2;
    const cx<double>* LB = reinterpret_cast<const cx<double>*>(it.lowB);
    cx<double>* LG = reinterpret_cast<cx<double>*>(it.lowG);
    const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
    const int w = blockIdx.y * (blockDim.x >> 6) + (threadIdx.x >> 6), nw = gridDim.y * (blockDim.x >> 6);
    const int nt = (K + 15) >> 4;
    for (int t = w; t < nt * nt; t += nw) {
        const int ti = t % nt, tj = t / nt;
        if (ti > tj) continue;
        v4d_t cr = {0, 0, 0, 0}, ci = {0, 0, 0, 0};
        ztile_mm(Nc, 16 * ti + l15, 16 * tj + l15,                                   // G[i][j] = sum_row conj(B[row, i]) B[row, j]
                 [&](int i, int k) { cx<double> v = (i < K && k < Nc) ? LB[k + (size_t)Nc * i] : cmake<double>(0, 0); v.im = -v.im; return v; },
                 [&](int k, int j) { return (j < K && k < Nc) ? LB[k + (size_t)Nc * j] : cmake<double>(0, 0); }, cr, ci);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * ti + kq + 4 * r, j = 16 * tj + l15;
            if (i < K && j < K && i <= j) { LG[i + (size_t)K * j] = cmake<double>(cr[r], ci[r]); if (i != j) LG[j + (size_t)K * i] = cmake<double>(cr[r], -ci[r]); }
        }
    }
}
void launch_lowrank_g(hipStream_t s, const GateItem* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL(lowrank_g_kernel, dim3(nitems, 4), dim3(1024), 0, s, d_items);
    TNQS_CHECK_LAUNCH();
}
// theta[:, 0..K) := M = A conj(L) where G = L L^dagger (chol_kernel); on a collapsed pivot the full theta (already in place) stays.
// Tiles with rows = column index j of M, lanes = row index i (contiguous in theta).
template <class T>
__global__ __launch_bounds__(1024) void lowrank_m_kernel(const GateItem* __restrict__ items) {
    const GateItem it = items[blockIdx.x];
    const int K = it.info[7];
    if (K <= 0) return;
    if (*it.lowfail) { if (threadIdx.x == 0 && blockIdx.y == 0) it.info[7] = 0; return; }
    const int Mr = it.info[0] * it.d1;
    const cx<double>* LA = reinterpret_cast<const cx<double>*>(it.lowA);
    const cx<double>* L = reinterpret_cast<const cx<double>*>(it.lowL);
    cx<T>* th = reinterpret_cast<cx<T>*>(it.theta);
    const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
    const int w = blockIdx.y * (blockDim.x >> 6) + (threadIdx.x >> 6), nw = gridDim.y * (blockDim.x >> 6);
    const int tr = (K + 15) >> 4, tc = (Mr + 15) >> 4;
    for (int t = w; t < tr * tc; t += nw) {
        const int j0 = 16 * (t % tr), i0 = 16 * (t / tr);
        v4d_t cr = {0, 0, 0, 0}, ci = {0, 0, 0, 0};
        ztile_mm(K, j0 + l15, i0 + l15,                                              // M[i][j] = sum_{l >= j} A[i, l] conj(L[l, j]), L lower triangular
                 [&](int j, int l) { cx<double> v = (j < K && l < K && l >= j) ? L[l + (size_t)K * j] : cmake<double>(0, 0); v.im = -v.im; return v; },
                 [&](int l, int i) { return (i < Mr && l < K) ? LA[i + (size_t)Mr * l] : cmake<double>(0, 0); }, cr, ci);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + kq + 4 * r, i = i0 + l15;
            if (i < Mr && j < K) th[i + (size_t)Mr * j] = cmake<T>((T)cr[r], (T)ci[r]);
        }
    }
    // Q = B L^-dagger = B W (Nc x K, orthonormal columns; W upper triangular): theta = M Q^T, so the right singular vectors of theta are
    // conj(Q) times those of M -- what theta_svd_pre_kernel builds V from (no recovery from theta0)
    if (!it.lowQ || !it.lowW) return;
    const int Nc = it.info[1] * it.d2;
    const cx<double>* LB = reinterpret_cast<const cx<double>*>(it.lowB);
    const cx<double>* W = reinterpret_cast<const cx<double>*>(it.lowW);
    cx<double>* Q = reinterpret_cast<cx<double>*>(it.lowQ);
    const int qc = (Nc + 15) >> 4;
    for (int t = w; t < tr * qc; t += nw) {
        const int j0 = 16 * (t % tr), i0 = 16 * (t / tr);
        v4d_t cr = {0, 0, 0, 0}, ci = {0, 0, 0, 0};
        ztile_mm(K, j0 + l15, i0 + l15,                                              // Q[i][j] = sum_{l <= j} B[i, l] W[l, j]   (tile rows = j, lanes = i)
                 [&](int j, int l) { return (j < K && l < K && l <= j) ? W[l + (size_t)K * j] : cmake<double>(0, 0); },
                 [&](int l, int i) { return (i < Nc && l < K) ? LB[i + (size_t)Nc * l] : cmake<double>(0, 0); }, cr, ci);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + kq + 4 * r, i = i0 + l15;
            if (i < Nc && j < K) Q[i + (size_t)Nc * j] = cmake<double>(cr[r], ci[r]);
        }
    }
}
// theta, theta0 *= 2^k with k = -exponent of the largest |entry| of theta0 (exact); *texp = k.  Runs after gate_theta / lowrank_m.
template <class T>
__global__ __launch_bounds__(1024) void theta_scale_kernel(const GateItem* __restrict__ items) {
    __shared__ float s_max[16];
    __shared__ int s_k;
    const GateItem it = items[blockIdx.x];
    const int ne = it.info[0] * it.d1 * it.info[1] * it.d2;
    cx<T>* th = reinterpret_cast<cx<T>*>(it.theta);
    cx<T>* th0 = reinterpret_cast<cx<T>*>(it.theta0);
    float mx = 0.f;
    for (int e = threadIdx.x; e < ne; e += blockDim.x) { cx<T> v = th0[e]; mx = fmaxf(mx, fmaxf(fabsf((float)v.re), fabsf((float)v.im))); }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m2 = 0.f; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) m2 = fmaxf(m2, s_max[w]);
        int k = 0;
        if (m2 > 0.f && m2 < 3e38f) { k = -ilogbf(m2); k = k > 120 ? 120 : (k < -120 ? -120 : k); }
        s_k = k; *it.texp = k;
    }
    __syncthreads();
    const int k = s_k;
    if (k == 0) return;
    const T sc = (T)ldexp(1.0, k);
    for (int e = threadIdx.x; e < ne; e += blockDim.x) { cx<T> a = th[e], b = th0[e]; th[e] = cmake<T>(a.re * sc, a.im * sc); th0[e] = cmake<T>(b.re * sc, b.im * sc); }
}
template <class T> void launch_theta_scale(hipStream_t s, const GateItem* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL((theta_scale_kernel<T>), dim3(nitems), dim3(1024), 0, s, d_items); TNQS_CHECK_LAUNCH();
}
template void launch_theta_scale<float>(hipStream_t, const GateItem*, int);
template void launch_theta_scale<double>(hipStream_t, const GateItem*, int);
template <class T> void launch_lowrank_m(hipStream_t s, const GateItem* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL((lowrank_m_kernel<T>), dim3(nitems, 4), dim3(1024), 0, s, d_items);
    TNQS_CHECK_LAUNCH();
}
template void launch_lowrank_m<float>(hipStream_t, const GateItem*, int);
template void launch_lowrank_m<double>(hipStream_t, const GateItem*, int);
// CholeskyQR2 of B (LowQr2Item, kernels.hpp): B1 = B W1, W1 = L1^-dagger upper triangular
__global__ __launch_bounds__(1024) void lowrank_bw_kernel(const LowQr2Item* __restrict__ items) {
    const LowQr2Item it = items[blockIdx.x];
    const int K = it.info[7];
    if (K <= 0 || *it.fail1) return;
    const int Nc = it.info[1] * it.d2;
    const cx<double>* B = reinterpret_cast<const cx<double>*>(it.B);
    const cx<double>* W = reinterpret_cast<const cx<double>*>(it.W1);
    cx<double>* B1 = reinterpret_cast<cx<double>*>(it.B1);
    const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
    const int w = blockIdx.y * (blockDim.x >> 6) + (threadIdx.x >> 6), nw = gridDim.y * (blockDim.x >> 6);
    const int tr = (K + 15) >> 4, tc = (Nc + 15) >> 4;
    for (int t = w; t < tr * tc; t += nw) {                                             // tile rows = column j of B1, lanes = row i (contiguous)
        const int j0 = 16 * (t % tr), i0 = 16 * (t / tr);
        v4d_t cr = {0, 0, 0, 0}, ci = {0, 0, 0, 0};
        ztile_mm(K, j0 + l15, i0 + l15,                                                 // B1[i][j] = sum_{l <= j} B[i, l] W[l, j]
                 [&](int j, int l) { return (j < K && l < K && l <= j) ? W[l + (size_t)K * j] : cmake<double>(0, 0); },
                 [&](int l, int i) { return (i < Nc && l < K) ? B[i + (size_t)Nc * l] : cmake<double>(0, 0); }, cr, ci);
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int j = j0 + kq + 4 * r, i = i0 + l15; if (i < Nc && j < K) B1[i + (size_t)Nc * j] = cmake<double>(cr[r], ci[r]); }
    }
}
void launch_lowrank_bw(hipStream_t s, const LowQr2Item* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL(lowrank_bw_kernel, dim3(nitems, 4), dim3(1024), 0, s, d_items); TNQS_CHECK_LAUNCH();
}
// Lc = L1 L2 (both lower triangular), and the second pass's failure folded into the gate's flag
__global__ __launch_bounds__(1024) void lowrank_ll_kernel(const LowQr2Item* __restrict__ items) {
    const LowQr2Item it = items[blockIdx.x];
    const int K = it.info[7];
    if (K <= 0 || *it.fail1) return;
    if (*it.fail2) { if (threadIdx.x == 0) *it.fail1 = 1; return; }
    const cx<double>* L1 = reinterpret_cast<const cx<double>*>(it.L1);
    const cx<double>* L2 = reinterpret_cast<const cx<double>*>(it.L2);
    cx<double>* Lc = reinterpret_cast<cx<double>*>(it.Lc);
    const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int nt = (K + 15) >> 4;
    for (int t = w; t < nt * nt; t += nw) {                                             // tile rows = column j, lanes = row i
        const int j0 = 16 * (t % nt), i0 = 16 * (t / nt);
        v4d_t cr = {0, 0, 0, 0}, ci = {0, 0, 0, 0};
        if (i0 + 15 >= j0)
        ztile_mm(K, j0 + l15, i0 + l15,                                                 // Lc[i][j] = sum_{j <= l <= i} L1[i, l] L2[l, j]
                 [&](int j, int l) { return (j < K && l < K && l >= j) ? L2[l + (size_t)K * j] : cmake<double>(0, 0); },
                 [&](int l, int i) { return (i < K && l < K && l <= i) ? L1[i + (size_t)K * l] : cmake<double>(0, 0); }, cr, ci);
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int j = j0 + kq + 4 * r, i = i0 + l15; if (i < K && j < K) Lc[i + (size_t)K * j] = (i >= j) ? cmake<double>(cr[r], ci[r]) : cmake<double>(0, 0); }
    }
}
void launch_lowrank_ll(hipStream_t s, const LowQr2Item* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL(lowrank_ll_kernel, dim3(nitems), dim3(1024), 0, s, d_items); TNQS_CHECK_LAUNCH();
}
template <class T> void launch_gate_theta(hipStream_t s, const GateItem* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL((gate_theta_kernel<T>), dim3(nitems, 8), dim3(1024), 0, s, d_items);
    TNQS_CHECK_LAUNCH();
}
template void launch_gate_theta<float>(hipStream_t, const GateItem*, int);
template void launch_gate_theta<double>(hipStream_t, const GateItem*, int);

// ---- second factorisation pass (CholeskyQR2) of ill-conditioned ComplexF64 sites: kernels.hpp, Qr2RinvItem / Qr2ComposeItem ----------
__global__ __launch_bounds__(256) void qr2_rinv_kernel(const Qr2RinvItem* __restrict__ items) {
    const Qr2RinvItem it = items[blockIdx.x];
    const int n = it.n, r = *it.r;
    const cx<double>* W = reinterpret_cast<const cx<double>*>(it.GW);
    cx<double>* X = reinterpret_cast<cx<double>*>(it.X1);
    for (int e = threadIdx.x; e < n * n; e += 256) {
        const int i = e % n, a = e / n;
        if (a < r) { const double sc = 1.0 / sqrt(it.lam[a]); cx<double> w = W[i + (size_t)n * it.idx[a]]; X[e] = cmake<double>(w.re * sc, w.im * sc); }
        else X[e] = cmake<double>(0, 0);
    }
}
void launch_qr2_rinv(hipStream_t s, const Qr2RinvItem* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL(qr2_rinv_kernel, dim3(nitems), dim3(256), 0, s, d_items); TNQS_CHECK_LAUNCH();
}
__global__ __launch_bounds__(256) void qr2_compose_kernel(const Qr2ComposeItem* __restrict__ items) {
    __shared__ double lam2[256]; __shared__ int sel[256]; __shared__ int s_r2;
    const Qr2ComposeItem it = items[blockIdx.x];
    const int n = it.n, r1 = *it.r1;
    const cx<double>* A2 = reinterpret_cast<const cx<double>*>(it.A2);
    const cx<double>* V2 = reinterpret_cast<const cx<double>*>(it.V2);
    const cx<double>* X1 = reinterpret_cast<const cx<double>*>(it.X1);
    const cx<double>* W1 = reinterpret_cast<const cx<double>*>(it.GV1);
    for (int j = threadIdx.x; j < n; j += 256) {          // Rayleigh quotients, as gate_eigs
        double l = 0;
        for (int i = 0; i < n; ++i) { cx<double> v = V2[i + (size_t)n * j], a = A2[i + (size_t)n * j]; l += v.re * a.re + v.im * a.im; }
        lam2[j] = l;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double lmax = 0; for (int j = 0; j < n; ++j) lmax = fmax(lmax, lam2[j]);
        int r = 0; for (int j = 0; j < n; ++j) if (lam2[j] > it.tau * lmax && lam2[j] > 0) sel[r++] = j;
        s_r2 = r; *it.rk = r;
    }
    __syncthreads();
    const int r2 = s_r2;
    cx<double>* GV = reinterpret_cast<cx<double>*>(it.GVout);
    cx<double>* GW = reinterpret_cast<cx<double>*>(it.GWout);
    // GW[:,c] = X1 V2[:,j_c] / sqrt(l2_c);   GV[:,c] = sqrt(l2_c) sum_a sqrt(l1_a) W1[:, idx1_a] V2[a, j_c]   (= conj of row c of R2 R1)
    for (int e = threadIdx.x; e < n * n; e += 256) {
        const int i = e % n, c = e / n;
        if (c >= r2) { GV[e] = cmake<double>(0, 0); GW[e] = cmake<double>(0, 0); continue; }
        const int j = sel[c]; const double sq = sqrt(lam2[j]);
        cx<double> gw = cmake<double>(0, 0), gv = cmake<double>(0, 0);
        for (int a = 0; a < r1; ++a) {
            const cx<double> v = V2[a + (size_t)n * j];
            cfma(gw, X1[i + (size_t)n * a], v);
            const double s1 = sqrt(it.lam1[a]); const cx<double> w = W1[i + (size_t)n * it.idx1[a]];
            cfma(gv, cmake<double>(w.re * s1, w.im * s1), v);
        }
        GW[e] = cmake<double>(gw.re / sq, gw.im / sq); GV[e] = cmake<double>(gv.re * sq, gv.im * sq);
    }
}
void launch_qr2_compose(hipStream_t s, const Qr2ComposeItem* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL(qr2_compose_kernel, dim3(nitems), dim3(256), 0, s, d_items); TNQS_CHECK_LAUNCH();
}

template <class T>
__global__ __launch_bounds__(1024) void gate_finish_kernel(const GateItem* __restrict__ items) {
    __shared__ double sig[512];
    __shared__ int perm[512];
    __shared__ int s_keep;
    // grid (gate, part) as gate_theta_kernel: the ranking / truncation prologue is repeated per part, only part 0 writes its results
    const GateItem it = items[blockIdx.x];
    const int part = blockIdx.y;
    const int r1 = it.info[0], r2 = it.info[1], d1 = it.d1, d2 = it.d2;
    const int Mr = r1 * d1, Nc = r2 * d2;
    const bool wide = it.info[5] != 0;                                // theta stored as theta^dagger (Nc x Mr)
    const int ncol = wide ? Mr : Nc, ld = wide ? Nc : Mr;
    const cx<T>* th = reinterpret_cast<const cx<T>*>(it.theta);      // rotated columns: U Sigma (or V Sigma when wide)
    const cx<T>* tv = reinterpret_cast<const cx<T>*>(it.thetaV);     // accumulated rotations: V (or U when wide)
    const int ncolK = (!wide && it.info[7] > 0) ? it.info[7] : ncol;        // low-rank route: the remaining singular values are zero
    const double tsc = ldexp(1.0, -(*it.texp));                               // theta was scaled by 2^texp
    for (int u = threadIdx.x; u < ncol; u += blockDim.x) {
        double s2 = 0;
        if (u < ncolK) for (int i = 0; i < ld; ++i) { cx<T> v = th[i + (size_t)ld * u]; s2 += (double)v.re * v.re + (double)v.im * v.im; }
        sig[u] = (s2 == s2 && s2 < 1e300) ? sqrt(s2) * tsc : 0.0;     // a NaN / inf column must not poison the ranking below; tsc undoes theta_scale_kernel
    }
    __syncthreads();
    for (int u = threadIdx.x; u < ncol; u += blockDim.x) {    // rank by counting (descending, stable)
        int rk = 0; double su = sig[u];
        for (int v = 0; v < ncol; ++v) rk += (sig[v] > su) || (sig[v] == su && v < u);
        perm[rk] = u;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // NDTensors truncate! on P = S^2 (computed in the data's real precision), relative cutoff, mindim = 1
        const int nsv = min(Mr, Nc);     // number of singular values of theta
        int n = nsv;
        int status = 0;
        T truncerr = 0;
        T p0 = (T)sig[perm[0]]; p0 = p0 * p0;
        if (p0 <= 0) { n = 1; }
        else if (nsv > 1) {
            const int md = it.maxdim > 0 ? it.maxdim : nsv;
            while (n > md) { T s = (T)sig[perm[n - 1]]; truncerr += s * s; --n; }
            T scale = 0;
            for (int i = 0; i < nsv; ++i) { T s = (T)sig[perm[i]]; scale += s * s; }
            if (scale == 0) scale = 1;
            const T cut = (T)(it.cutoff < 0 ? 0.0 : it.cutoff);
            while (n > 1) { T s = (T)sig[perm[n - 1]]; T p = s * s; if (truncerr + p <= cut * scale) { truncerr += p; --n; } else break; }
            truncerr = truncerr / scale;
        }
        if (n > it.chi_cap) { status = 1; n = it.chi_cap; }
        double nrm = 0;
        for (int i = 0; i < n; ++i) nrm += sig[perm[i]] * sig[perm[i]];
        nrm = sqrt(nrm);
        if (part == 0) {
            for (int i = 0; i < n; ++i) {
                double s = sig[perm[i]];
                it.S[i] = (it.normalize && nrm > 0) ? (double)((T)s / (T)nrm) : (double)(T)s;
            }
            it.info[2] = n; it.info[3] = status; *it.truncerr = (double)truncerr;
        }
        s_keep = n;
    }
    // per-column factors of R^+ = W diag(lambda^-1/2) (and of the theta scaling), once per workgroup: a square root and a division per
    // INNER iteration were most of this kernel's time
    __shared__ double fa1[512], fa2[512];
    for (int a = threadIdx.x; a < r1; a += blockDim.x) fa1[a] = (wide ? 1.0 : tsc) / sqrt(it.lam1[a]);      // th holds the scaled U Sigma (tv, the recovered vectors, is scale free)
    for (int c = threadIdx.x; c < r2; c += blockDim.x) fa2[c] = (wide ? tsc : 1.0) / sqrt(it.lam2[c]);
    __syncthreads();
    const int nk = s_keep;
    const cx<double>* V1 = reinterpret_cast<const cx<double>*>(it.GW1);       // R^+ = W diag(lambda^-1/2)
    const cx<double>* V2 = reinterpret_cast<const cx<double>*>(it.GW2);
    cx<T>* X1 = reinterpret_cast<cx<T>*>(it.X1);
    cx<T>* X2 = reinterpret_cast<cx<T>*>(it.X2);
    const int n1 = it.n1, n2 = it.n2;
    // X1[(s,b),(s1',u)] = sum_a W1[(s,b),a] / sqrt(l1_a) * (U Sigma)[(a,s1'),pi(u)] / sqrt(sigma_u)
    // X2[(s,b),(s2',u)] = sum_c W2[(s,b),c] / sqrt(l2_c) * sqrt(sigma_u) conj(Vtheta[(c,s2'),pi(u)])
    // Two small complex products (64 x 64 x 64 at chi = 32) on the f64 matrix cores, one wave per 16 x 16 tile (ztile_mm); the tile's lanes
    // run along (s,b), the contiguous index of X.  (The scalar loops these replace chased idx -> W -> multiply-add through L2 once per term:
    // 0.23 ms per launch at chi = 32, most of it load latency.)
    const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
    const int wv = part * (blockDim.x >> 6) + (threadIdx.x >> 6), nwv = gridDim.y * (blockDim.x >> 6);
    const int N1 = d1 * nk, N2 = d2 * nk;
    const int t1r = (N1 + 15) >> 4, t1c = (n1 + 15) >> 4, t2r = (N2 + 15) >> 4, t2c = (n2 + 15) >> 4;
    for (int t = wv; t < t1r * t1c + t2r * t2c; t += nwv) {
        const bool second = t >= t1r * t1c;
        const int tt = second ? t - t1r * t1c : t;
        const int tr = second ? t2r : t1r;
        const int r0 = 16 * (tt % tr), c0 = 16 * (tt / tr);
        v4d_t cr = {0, 0, 0, 0}, ci = {0, 0, 0, 0};
        if (!second) {
            ztile_mm(r1, r0 + l15, c0 + l15,
                     [&](int nn, int a2) {
                         if (nn >= N1 || a2 >= r1) return cmake<double>(0, 0);
                         const int s1p = nn % d1, u = nn / d1, pu = perm[u];
                         const double su = sig[pu];
                         if (!(su > 0)) return cmake<double>(0, 0);
                         const cx<T> l = wide ? tv[(a2 + r1 * s1p) + (size_t)Mr * pu] : th[(a2 + r1 * s1p) + (size_t)Mr * pu];
                         const double f = fa1[a2] * (wide ? sqrt(su) : 1.0 / sqrt(su));              // L = U sqrt(S)
                         return cmake<double>(l.re * f, l.im * f);
                     },
                     [&](int a2, int kk) { return (kk < n1 && a2 < r1) ? V1[kk + (size_t)n1 * it.idx1[a2]] : cmake<double>(0, 0); }, cr, ci);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nn = r0 + kq + 4 * r, kk = c0 + l15;
                if (nn < N1 && kk < n1) X1[kk + (size_t)n1 * nn] = cmake<T>((T)cr[r], (T)ci[r]);
            }
        } else {
            ztile_mm(r2, r0 + l15, c0 + l15,
                     [&](int nn, int c2) {
                         if (nn >= N2 || c2 >= r2) return cmake<double>(0, 0);
                         const int s2p = nn % d2, u = nn / d2, pu = perm[u];
                         const double su = sig[pu];
                         if (wide && !(su > 0)) return cmake<double>(0, 0);
                         const cx<T> v = wide ? th[(c2 + r2 * s2p) + (size_t)Nc * pu] : tv[(c2 + r2 * s2p) + (size_t)Nc * pu];
                         const double f = fa2[c2] * (wide ? 1.0 / sqrt(su) : sqrt(su));              // R = sqrt(S) V^dagger
                         return cmake<double>(v.re * f, -v.im * f);
                     },
                     [&](int c2, int kk) { return (kk < n2 && c2 < r2) ? V2[kk + (size_t)n2 * it.idx2[c2]] : cmake<double>(0, 0); }, cr, ci);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nn = r0 + kq + 4 * r, kk = c0 + l15;
                if (nn < N2 && kk < n2) X2[kk + (size_t)n2 * nn] = cmake<T>((T)cr[r], (T)ci[r]);
            }
        }
    }
}
template <class T> void launch_gate_finish(hipStream_t s, const GateItem* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL((gate_finish_kernel<T>), dim3(nitems, 4), dim3(1024), 0, s, d_items);
    TNQS_CHECK_LAUNCH();
}
template void launch_gate_finish<float>(hipStream_t, const GateItem*, int);
template void launch_gate_finish<double>(hipStream_t, const GateItem*, int);

// ------------------------------------------------------------------------------------------------------------
// small utilities
// ------------------------------------------------------------------------------------------------------------
template <class T> __global__ void diag_kernel(const DiagItem* __restrict__ items) {
    const DiagItem it = items[blockIdx.x];
    cx<T>* out = reinterpret_cast<cx<T>*>(it.out);
    for (int e = threadIdx.x; e < it.chi * it.chi; e += blockDim.x) {
        int i = e % it.chi, j = e / it.chi;
        out[e] = cmake<T>(i == j ? (T)it.S[i] : (T)0, (T)0);
    }
}
template <class T> void launch_diag(hipStream_t s, const DiagItem* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL((diag_kernel<T>), dim3(nitems), dim3(256), 0, s, d_items); TNQS_CHECK_LAUNCH();
}
template void launch_diag<float>(hipStream_t, const DiagItem*, int);
template void launch_diag<double>(hipStream_t, const DiagItem*, int);

__global__ __launch_bounds__(256) void norm_factor_kernel(const NormFactorItem* __restrict__ items) {
    __shared__ double sh[17];
    const NormFactorItem it = items[blockIdx.x];
    double t = 0;
    for (int i = threadIdx.x; i < it.npart; i += 256) t += it.norm_partials[i];
    // fixed-order reduction would need a second pass; the block_sum order is deterministic for a given launch shape
    t = block_sum(t, sh);
    if (threadIdx.x == 0) *it.factor = (t > 0) ? 1.0 / sqrt(t) : 1.0;
}
void launch_norm_factor(hipStream_t s, const NormFactorItem* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL(norm_factor_kernel, dim3(nitems), dim3(256), 0, s, d_items); TNQS_CHECK_LAUNCH();
}
template <class T> __global__ __launch_bounds__(256) void scale_kernel(const ScaleItem* __restrict__ items) {
    const ScaleItem it = items[blockIdx.y];
    const T f = (T)(*it.factor);
    const cx<T>* __restrict__ p = reinterpret_cast<const cx<T>*>(it.src);
    cx<T>* __restrict__ q = reinterpret_cast<cx<T>*>(it.dst);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < it.n; i += (size_t)gridDim.x * 256) {
        cx<T> v = p[i]; q[i] = cmake<T>(v.re * f, v.im * f);
    }
}
template <class T> void launch_scale(hipStream_t s, const ScaleItem* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL((scale_kernel<T>), dim3(64, nitems), dim3(256), 0, s, d_items); TNQS_CHECK_LAUNCH();
}
template void launch_scale<float>(hipStream_t, const ScaleItem*, int);
template void launch_scale<double>(hipStream_t, const ScaleItem*, int);

// ------------------------------------------------------------------------------------------------------------
// BP normalisation (rescale!, beliefpropagationcache.jl:82-140)
// ------------------------------------------------------------------------------------------------------------
template <class T> __device__ __forceinline__ cx<double> msg_elem(const cx<T>* m, int e, int chi) {
    if (m) return cmake<double>((double)m[e].re, (double)m[e].im);
    return cmake<double>((e % chi) == (e / chi) ? 1.0 : 0.0, 0.0);
}
template <class T> __global__ __launch_bounds__(256) void msg_rescale_kernel(const MsgRescaleItem* __restrict__ items) {
    __shared__ double sh[17];
    const MsgRescaleItem it = items[blockIdx.x];
    const cx<T>* a = reinterpret_cast<const cx<T>*>(it.me); const cx<T>* b = reinterpret_cast<const cx<T>*>(it.mer);
    const int n2 = it.chi * it.chi;
    double na = 0, nb = 0, pr = 0, pi = 0;
    for (int e = threadIdx.x; e < n2; e += 256) {
        cx<double> x = msg_elem(a, e, it.chi), y = msg_elem(b, e, it.chi);
        na += x.re * x.re + x.im * x.im; nb += y.re * y.re + y.im * y.im;
        pr += x.re * y.re - x.im * y.im; pi += x.re * y.im + x.im * y.re;
    }
    na = block_sum(na, sh); nb = block_sum(nb, sh); pr = block_sum(pr, sh); pi = block_sum(pi, sh);
    const double ia = na > 0 ? 1.0 / sqrt(na) : 0.0, ib = nb > 0 ? 1.0 / sqrt(nb) : 0.0;
    double nr = pr * ia * ib, ni = pi * ia * ib;             // n = scalar(normalize(me) * normalize(mer))
    double sgn = 1.0;
    if (ni == 0.0) { sgn = (nr > 0) - (nr < 0); nr *= sgn; }  // isreal(n): me *= sign(n), n *= sign(n)
    // 1/sqrt(n), principal branch
    const double mod = sqrt(nr * nr + ni * ni), arg = atan2(ni, nr);
    const double r = mod > 0 ? 1.0 / sqrt(mod) : 0.0, ph = -0.5 * arg;
    const double fr = r * cos(ph), fi = r * sin(ph);
    cx<T>* ao = reinterpret_cast<cx<T>*>(it.me_out); cx<T>* bo = reinterpret_cast<cx<T>*>(it.mer_out);
    for (int e = threadIdx.x; e < n2; e += 256) {
        cx<double> x = msg_elem(a, e, it.chi), y = msg_elem(b, e, it.chi);
        x.re *= ia * sgn; x.im *= ia * sgn; y.re *= ib; y.im *= ib;
        ao[e] = cmake<T>((T)(x.re * fr - x.im * fi), (T)(x.re * fi + x.im * fr));
        bo[e] = cmake<T>((T)(y.re * fr - y.im * fi), (T)(y.re * fi + y.im * fr));
    }
}
template <class T> void launch_msg_rescale(hipStream_t s, const MsgRescaleItem* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL((msg_rescale_kernel<T>), dim3(nitems), dim3(256), 0, s, d_items); TNQS_CHECK_LAUNCH();
}
template void launch_msg_rescale<float>(hipStream_t, const MsgRescaleItem*, int);
template void launch_msg_rescale<double>(hipStream_t, const MsgRescaleItem*, int);
template <class T> __global__ __launch_bounds__(256) void edge_scalar_kernel(const EdgeScalarItem* __restrict__ items) {
    __shared__ double sh[17];
    const EdgeScalarItem it = items[blockIdx.x];
    const cx<T>* a = reinterpret_cast<const cx<T>*>(it.me); const cx<T>* b = reinterpret_cast<const cx<T>*>(it.mer);
    const int n2 = it.chi * it.chi;
    double pr = 0, pi = 0;
    for (int e = threadIdx.x; e < n2; e += 256) {
        cx<double> x = msg_elem(a, e, it.chi), y = msg_elem(b, e, it.chi);
        pr += x.re * y.re - x.im * y.im; pi += x.re * y.im + x.im * y.re;
    }
    pr = block_sum(pr, sh); pi = block_sum(pi, sh);
    if (threadIdx.x == 0) { it.out[0] = pr; it.out[1] = pi; }
}
template <class T> void launch_edge_scalar(hipStream_t s, const EdgeScalarItem* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL((edge_scalar_kernel<T>), dim3(nitems), dim3(256), 0, s, d_items); TNQS_CHECK_LAUNCH();
}
template void launch_edge_scalar<float>(hipStream_t, const EdgeScalarItem*, int);
template void launch_edge_scalar<double>(hipStream_t, const EdgeScalarItem*, int);
template <class T> __global__ __launch_bounds__(256) void cscale_kernel(const CScaleItem* __restrict__ items) {
    const CScaleItem it = items[blockIdx.y];
    const double fr = it.re, fi = it.im;
    const cx<T>* __restrict__ p = reinterpret_cast<const cx<T>*>(it.src);
    cx<T>* __restrict__ q = reinterpret_cast<cx<T>*>(it.dst);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < it.n; i += (size_t)gridDim.x * 256) {
        cx<T> v = p[i]; q[i] = cmake<T>((T)(v.re * fr - v.im * fi), (T)(v.re * fi + v.im * fr));
    }
}
template <class T> void launch_cscale(hipStream_t s, const CScaleItem* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL((cscale_kernel<T>), dim3(64, nitems), dim3(256), 0, s, d_items); TNQS_CHECK_LAUNCH();
}
template void launch_cscale<float>(hipStream_t, const CScaleItem*, int);
template void launch_cscale<double>(hipStream_t, const CScaleItem*, int);

// ------------------------------------------------------------------------------------------------------------
// symmetric gauge (src/symmetric_gauge.jl)
// ------------------------------------------------------------------------------------------------------------
template <class T> __global__ __launch_bounds__(256) void symg_build_kernel(const SymGaugeItem* __restrict__ items) {
    __shared__ double lx[256], ly[256];
    const SymGaugeItem it = items[blockIdx.x];
    const int n = it.n;
    const cx<double>* AX = reinterpret_cast<const cx<double>*>(it.AX); const cx<double>* VX = reinterpret_cast<const cx<double>*>(it.VX);
    const cx<double>* AY = reinterpret_cast<const cx<double>*>(it.AY); const cx<double>* VY = reinterpret_cast<const cx<double>*>(it.VY);
    for (int j = threadIdx.x; j < n; j += 256) {
        double a = 0, b = 0;            // Rayleigh quotients v_j^dagger H v_j
        for (int i = 0; i < n; ++i) { cx<double> v = VX[i + n * j], w = AX[i + n * j]; a += v.re * w.re + v.im * w.im;
                                      cx<double> p = VY[i + n * j], q = AY[i + n * j]; b += p.re * q.re + p.im * q.im; }
        a += it.reg; b += it.reg;       // map_diag(x -> x + regularization) (:15-16)
        if (a < 0 || b < 0) *it.flag = 1;     // sqrt of a negative real: DomainError in the reference
        lx[j] = a; ly[j] = b;
    }
    __syncthreads();
    cx<double>* rx = reinterpret_cast<cx<double>*>(it.rx); cx<double>* ry = reinterpret_cast<cx<double>*>(it.ry);
    cx<double>* irx = reinterpret_cast<cx<double>*>(it.irx); cx<double>* iry = reinterpret_cast<cx<double>*>(it.iry);
    // ITensors.eigen without index sets diagonalises M^T (the primed index is the row index), so every function of the message
    // enters as f(M)^T = conj(f(M)) [l, l']  (:13-24)
    for (int e = threadIdx.x; e < n * n; e += 256) {
        int i = e % n, l = e / n;
        cx<double> sx = cmake<double>(0, 0), ix = sx, sy = sx, iy = sx;
        for (int j = 0; j < n; ++j) {
            cx<double> vi = VX[i + n * j], vl = VX[l + n * j];
            cx<double> o = cmake<double>(vi.re * vl.re + vi.im * vl.im, -(vi.im * vl.re - vi.re * vl.im));      // conj(vi conj(vl))
            double r = lx[j] > 0 ? sqrt(lx[j]) : 0.0, ir = lx[j] > 0 ? 1.0 / r : 0.0;
            sx.re += r * o.re; sx.im += r * o.im; ix.re += ir * o.re; ix.im += ir * o.im;
            cx<double> wi = VY[i + n * j], wl = VY[l + n * j];
            cx<double> p = cmake<double>(wi.re * wl.re + wi.im * wl.im, -(wi.im * wl.re - wi.re * wl.im));
            double q = ly[j] > 0 ? sqrt(ly[j]) : 0.0, iq = ly[j] > 0 ? 1.0 / q : 0.0;
            sy.re += q * p.re; sy.im += q * p.im; iy.re += iq * p.re; iy.im += iq * p.im;
        }
        rx[e] = sx; irx[e] = ix; ry[e] = sy; iry[e] = iy;
    }
    __syncthreads();
    __threadfence_block();
    cx<T>* Ce = reinterpret_cast<cx<T>*>(it.Ce); cx<T>* Ce0 = reinterpret_cast<cx<T>*>(it.Ce0);
    for (int e = threadIdx.x; e < n * n; e += 256) {          // Ce[l, c] = sum_l' rootX[l, l'] rootY[c, l']   (:29-30)
        int l = e % n, c = e / n;
        cx<double> acc = cmake<double>(0, 0);
        for (int k = 0; k < n; ++k) cfma(acc, rx[l + n * k], ry[c + n * k]);
        cx<T> v = cmake<T>((T)acc.re, (T)acc.im);
        Ce[e] = v; Ce0[e] = v;
    }
}
template <class T> void launch_symg_build(hipStream_t s, const SymGaugeItem* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL((symg_build_kernel<T>), dim3(nitems), dim3(256), 0, s, d_items); TNQS_CHECK_LAUNCH();
}
template void launch_symg_build<float>(hipStream_t, const SymGaugeItem*, int);
template void launch_symg_build<double>(hipStream_t, const SymGaugeItem*, int);
template <class T> __global__ __launch_bounds__(256) void symg_finish_kernel(const SymGaugeItem* __restrict__ items) {
    __shared__ double sig[256];
    __shared__ int perm[256];
    const SymGaugeItem it = items[blockIdx.x];
    const int n = it.n;
    const cx<T>* A = reinterpret_cast<const cx<T>*>(it.Ce);       // U Sigma
    const cx<T>* V = reinterpret_cast<const cx<T>*>(it.Vsvd);
    for (int u = threadIdx.x; u < n; u += 256) {
        double s2 = 0;
        for (int i = 0; i < n; ++i) { cx<T> v = A[i + (size_t)n * u]; s2 += (double)v.re * v.re + (double)v.im * v.im; }
        sig[u] = (s2 == s2 && s2 < 1e300) ? sqrt(s2) : 0.0;
    }
    __syncthreads();
    for (int u = threadIdx.x; u < n; u += 256) {                  // descending order, stable
        int rk = 0; double su = sig[u];
        for (int v = 0; v < n; ++v) rk += (sig[v] > su) || (sig[v] == su && v < u);
        perm[rk] = u;
    }
    __syncthreads();
    for (int u = threadIdx.x; u < n; u += 256) it.S[u] = (double)(T)sig[perm[u]];
    const cx<double>* irx = reinterpret_cast<const cx<double>*>(it.irx); const cx<double>* iry = reinterpret_cast<const cx<double>*>(it.iry);
    cx<T>* Xs = reinterpret_cast<cx<T>*>(it.Xs); cx<T>* Xd = reinterpret_cast<cx<T>*>(it.Xd);
    for (int e = threadIdx.x; e < n * n; e += 256) {
        int l = e % n, u = e / n; int pu = perm[u]; double su = sig[pu];
        cx<double> a = cmake<double>(0, 0), b = cmake<double>(0, 0);
        if (su > 0) {
            for (int k = 0; k < n; ++k) {
                cx<T> x = A[k + (size_t)n * pu], y = V[k + (size_t)n * pu];
                cfma(a, irx[l + n * k], cmake<double>((double)x.re, (double)x.im));
                cfma(b, iry[l + n * k], cmake<double>((double)y.re, -(double)y.im));      // V^T of ITensors = conj of the right singular vectors
            }
            const double f = 1.0 / sqrt(su), gq = sqrt(su);       // U = A / sigma, times sqrt(sigma)
            a.re *= f; a.im *= f; b.re *= gq; b.im *= gq;
        }
        Xs[e] = cmake<T>((T)a.re, (T)a.im); Xd[e] = cmake<T>((T)b.re, (T)b.im);
    }
}
template <class T> void launch_symg_finish(hipStream_t s, const SymGaugeItem* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL((symg_finish_kernel<T>), dim3(nitems), dim3(256), 0, s, d_items); TNQS_CHECK_LAUNCH();
}
template void launch_symg_finish<float>(hipStream_t, const SymGaugeItem*, int);
template void launch_symg_finish<double>(hipStream_t, const SymGaugeItem*, int);

__global__ __launch_bounds__(256) void record_pack_kernel(const RecordPackItem* __restrict__ items) {
    const RecordPackItem it = items[blockIdx.x];
    char* dst = reinterpret_cast<char*>(it.dst);
    if (threadIdx.x == 0) { double* h = reinterpret_cast<double*>(dst); h[0] = (double)it.info[2]; h[1] = (double)it.info[3]; h[2] = *it.terr; h[3] = 0.0; }
    double* sd = reinterpret_cast<double*>(dst + 32);
    for (int i = threadIdx.x; i < it.nS; i += 256) sd[i] = it.S[i];
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(it.X2);
    unsigned long long* xd = reinterpret_cast<unsigned long long*>(dst + it.x2_off);
    for (long long i = threadIdx.x; i < it.x2_words; i += 256) xd[i] = src[i];
}
void launch_record_pack(hipStream_t s, const RecordPackItem* d_items, int nitems) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL(record_pack_kernel, dim3(nitems), dim3(256), 0, s, d_items); TNQS_CHECK_LAUNCH();
}
__global__ void header_gather_kernel(const void* const* __restrict__ srcs, int n, double* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 4 * n) out[i] = reinterpret_cast<const double*>(srcs[i >> 2])[i & 3];
}
void launch_header_gather(hipStream_t s, const void* const* d_srcs, int n, double* d_out) {
    if (n <= 0) return;
    hipLaunchKernelGGL(header_gather_kernel, dim3((4 * n + 255) / 256), dim3(256), 0, s, d_srcs, n, d_out); TNQS_CHECK_LAUNCH();
}

template <class T> __global__ __launch_bounds__(256) void permute_kernel(PermItem it) {
    const cx<T>* in = reinterpret_cast<const cx<T>*>(it.in);
    cx<T>* out = reinterpret_cast<cx<T>*>(it.out);
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < it.n; e += (size_t)gridDim.x * 256) {
        size_t rem = e; long long off = 0;
        for (int k = 0; k < it.ndim; ++k) { int idx = (int)(rem % it.dims_out[k]); rem /= it.dims_out[k]; off += idx * it.stride_in[k]; }
        out[e] = in[off];     // pure data movement: bit-exact
    }
}
template <class T> void launch_permute(hipStream_t s, const PermItem& item) {
    if (item.n == 0) return;
    int blocks = (int)((item.n + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((permute_kernel<T>), dim3(blocks), dim3(256), 0, s, item); TNQS_CHECK_LAUNCH();
}
template void launch_permute<float>(hipStream_t, const PermItem&);
template void launch_permute<double>(hipStream_t, const PermItem&);

template <class T> __global__ void identity_kernel(cx<T>* out, int n) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n * n; e += gridDim.x * blockDim.x)
        out[e] = cmake<T>((e % n) == (e / n) ? (T)1 : (T)0, (T)0);
}
// iid standard-normal (re, im) pairs from a counter-based generator: entry e <- splitmix64(seed + e) -> two uniforms -> Box-Muller
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
template <class T> __global__ __launch_bounds__(256) void random_fill_kernel(cx<T>* out, size_t n, unsigned long long seed, double scale, int real_only) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long r = splitmix64(seed + 0xD1B54A32D192ED03ull * (unsigned long long)e);
        const double u1 = ((double)(r >> 32) + 1.0) * (1.0 / 4294967296.0), u2 = (double)(r & 0xffffffffull) * (1.0 / 4294967296.0);
        const double rad = sqrt(-2.0 * log(u1)) * scale; double sn, cs; sincospi(2.0 * u2, &sn, &cs);
        out[e] = cmake<T>((T)(rad * cs), real_only ? (T)0 : (T)(rad * sn));
    }
}
template <class T> void launch_random_fill(hipStream_t s, void* out, size_t n, unsigned long long seed, double scale, bool real_only) {
    if (!n) return;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 65536);
    hipLaunchKernelGGL((random_fill_kernel<T>), dim3(blocks), dim3(256), 0, s, reinterpret_cast<cx<T>*>(out), n, seed, scale, real_only ? 1 : 0); TNQS_CHECK_LAUNCH();
}
template void launch_random_fill<float>(hipStream_t, void*, size_t, unsigned long long, double, bool);
template void launch_random_fill<double>(hipStream_t, void*, size_t, unsigned long long, double, bool);
template <class T> void launch_identity(hipStream_t s, void* out, int n) {
    hipLaunchKernelGGL((identity_kernel<T>), dim3((n * n + 255) / 256), dim3(256), 0, s, reinterpret_cast<cx<T>*>(out), n); TNQS_CHECK_LAUNCH();
}
template void launch_identity<float>(hipStream_t, void*, int);
template void launch_identity<double>(hipStream_t, void*, int);

// one-site gate, d = 2, ComplexF32: out[s'] = sum_s G[s',s] in[s] on 16-byte (s=0,1) pairs; K11 of SURVEY.md 2
__global__ __launch_bounds__(256) void site1_c64_kernel(const Site1Item* __restrict__ items, double* __restrict__ norm_partials) {
    __shared__ double sh[17];
    const Site1Item it = items[blockIdx.y];
    const float4* __restrict__ in = reinterpret_cast<const float4*>(it.in);
    float4* __restrict__ out = reinterpret_cast<float4*>(it.out);
    const float g00r = it.g[0], g00i = it.g[1], g01r = it.g[2], g01i = it.g[3], g10r = it.g[4], g10i = it.g[5], g11r = it.g[6], g11i = it.g[7];
    double nrm = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < it.npairs; i += (size_t)gridDim.x * 256) {
        float4 a = in[i];                       // (a0.re, a0.im, a1.re, a1.im)
        float4 o;
        o.x = g00r * a.x - g00i * a.y + g01r * a.z - g01i * a.w;
        o.y = g00r * a.y + g00i * a.x + g01r * a.w + g01i * a.z;
        o.z = g10r * a.x - g10i * a.y + g11r * a.z - g11i * a.w;
        o.w = g10r * a.y + g10i * a.x + g11r * a.w + g11i * a.z;
        out[i] = o;
        nrm += (double)o.x * o.x + (double)o.y * o.y + (double)o.z * o.z + (double)o.w * o.w;
    }
    if (norm_partials) {
        double t = block_sum(nrm, sh);
        if (threadIdx.x == 0) norm_partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
    }
}
void launch_site1_c64(hipStream_t s, const Site1Item* d_items, int nitems, int nbx, double* d_norm_partials) {
    if (nitems <= 0) return;
    hipLaunchKernelGGL(site1_c64_kernel, dim3(nbx, nitems), dim3(256), 0, s, d_items, d_norm_partials); TNQS_CHECK_LAUNCH();
}

__global__ void sum_doubles_kernel(const double* in, int n, double* out) {
    __shared__ double sh[17];
    double t = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) t += in[i];
    t = block_sum(t, sh);
    if (threadIdx.x == 0) *out = t;
}
void launch_sum_doubles(hipStream_t s, const double* in, int n, double* out) {
    hipLaunchKernelGGL(sum_doubles_kernel, dim3(1), dim3(256), 0, s, in, n, out); TNQS_CHECK_LAUNCH();
}

}  // namespace tnqs
