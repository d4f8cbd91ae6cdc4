// int8_gram_probe.hip -- EXPERIMENT (not part of the library): the f64 Gram of the gate path (G = psi~^dagger psi~, 64 complex columns, csrc/kernels_gate.hip)
// on the INTEGER matrix cores instead of v_mfma_f64_16x16x4_f64.
//
// Why f64 at all: the thin QR of simple_update.jl:45-48 is taken as the Cholesky factor of G, which squares the condition number -- G has to be the Gram matrix of the
// f32 tensor to (much) better than f32 accuracy.  The f64 matrix pipe delivers that at 78 TFLOP/s; the int8 pipe runs at ~3900 TOPS.  Fixed point makes the integer
// pipe EXACT: with one power-of-two scale per column, x -> q = rint(x 2^(30 - E_c)) is a 32-bit integer, q = sum_i d_i 256^i with four signed digits d_i in
// [-128, 127], and  sum_rows q q' = sum_ij 256^(i+j) sum_rows d_i d'_j  -- every digit product is exact in the i32 accumulator (2^14 per product, < 2^31 over 16384 rows
// x 4 pairs).  The pairs with i + j <= 1 weigh < 2^-34 of the result and are dropped (13 of 16 products, five accumulators by i + j).  What is computed is the EXACT
// Gram matrix of the tensor rounded to 2^(E_c - 30) per element, E_c >= log2 max |column c|: a perturbation below f32 rounding of the column's largest element --
// and ONLY of entries within 2^6 of it: the rounding is absolute, f32's is relative to each entry, so columns whose entries span many decades (the tensors of a truncated
// evolution do) are represented 2^(range - 6) times more coarsely than the f32 data they come from.  DESIGN.md 8.3 says why this stays an experiment.
// The digits cost 5 + 2 vector instructions per value:  t = q + 0x00808080;  w = t ^ 0x00808080  -- the four bytes of w ARE the signed digits (adding 128 per lower
// byte and flipping its top bit back is the balanced-digit conversion) -- then a 4 x 4 byte transpose (v_perm_b32) packs four rows per digit plane.
//
// The probe measures the kernel a gate batch would launch (380 sites x 32768 rows x 64 complex columns, ComplexF32 planes [re | im][column][row]: what the LDS image of
// mfma_gauge_gram64_kernel holds after its transform) and checks site 0 against an f64 Gram on the host.  Build + run on the GPU box:
//     hipcc -O3 --offload-arch=gfx950 profiles/int8_gram_probe.hip -o /tmp/int8_gram_probe && /tmp/int8_gram_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int NCOL = 128;                 // real columns: 64 re + 64 im
constexpr int TR = 64;                    // rows per tile
constexpr int CS = 80;                    // bytes per column of a digit plane in LDS (64 rows + pad: the 16 lanes of a b128 read group hit 16 different 16-byte slots)
constexpr int PLANE = NCOL * CS;          // one digit plane
constexpr int BUF = 4 * PLANE;            // four digit planes = one tile

__device__ __forceinline__ unsigned digits_of(float x, float scale) {
    const int q = __float2int_rn(x * scale);
    return (unsigned)(q + 0x00808080) ^ 0x00808080u;
}
// 4 x 4 byte transpose: in[r] = digits (d0 d1 d2 d3, d0 lowest byte) of row r  ->  out[i] = digit i of rows 0..3 (row 0 lowest byte)
__device__ __forceinline__ void transpose4(const unsigned (&in)[4], unsigned (&out)[4]) {
    const unsigned a = __builtin_amdgcn_perm(in[1], in[0], 0x05010400u);   // (r0.d0, r1.d0, r0.d1, r1.d1)
    const unsigned b = __builtin_amdgcn_perm(in[1], in[0], 0x07030602u);   // (r0.d2, r1.d2, r0.d3, r1.d3)
    const unsigned c = __builtin_amdgcn_perm(in[3], in[2], 0x05010400u);
    const unsigned d = __builtin_amdgcn_perm(in[3], in[2], 0x07030602u);
    out[0] = __builtin_amdgcn_perm(c, a, 0x05040100u);                     // (r0.d0, r1.d0, r2.d0, r3.d0)
    out[1] = __builtin_amdgcn_perm(c, a, 0x07060302u);
    out[2] = __builtin_amdgcn_perm(d, b, 0x05040100u);
    out[3] = __builtin_amdgcn_perm(d, b, 0x07060302u);
}

struct Block { int acol, bcol, part2; };   // A columns acol .. acol + 31, B columns bcol .. bcol + 31; part2: also the same block 64 columns further (re + im parts of Gr)

// one workgroup = 4 waves; X: [site][128 columns][nrows] f32; scale: [site][128] = 2^(30 - E_c); out: [site][chunk][7 blocks][5 accumulators][1024] i32
template <int MODE>     // 0: full; 1: no matrix instructions; 2: no quantisation (digit planes written once)
__global__ __launch_bounds__(256, 2) void int8_gram_kernel(const float* __restrict__ X, const float* __restrict__ scale, int* __restrict__ out, int nrows, int tiles_per_chunk, int nchunks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int site = blockIdx.x / nchunks, chunk = blockIdx.x % nchunks;
    const float* __restrict__ Xs = X + (size_t)site * NCOL * nrows;
    // mover: column tid >> 1, rows 32 (tid & 1) .. + 31 of the tile: eight 16-byte loads
    const int mc = tid >> 1, mr = 32 * (tid & 1);
    const float sc = scale[site * NCOL + mc];
    const float* __restrict__ src = Xs + (size_t)mc * nrows + mr;
    v4f px[8];
    auto issue = [&](int t) {
        const float* p = src + (size_t)t * TR;
#pragma unroll
        for (int j = 0; j < 8; ++j) px[j] = *reinterpret_cast<const v4f*>(p + 4 * j);
    };
    auto commit = [&](int buf) {
        unsigned char* base = smem + buf * BUF + mc * CS + mr;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned in[4], o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) in[e] = digits_of(px[j][e], sc);
            transpose4(in, o);
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<unsigned*>(base + i * PLANE + 4 * j) = o[i];
        }
    };
    // blocks of this wave.  Gr = Xr^T Xr + Xi^T Xi (upper blocks), P1 = Xr^T Xi (all four blocks; Gi(I,J) = P1(I,J) - P1(J,I)^T)
    //   wave 0: Gr(0,0), P1(0,0)   wave 1: Gr(1,1), P1(1,1)   wave 2: Gr(0,1)   wave 3: P1(0,1), P1(1,0)
    // (the role is a compile-time constant of the loop: accumulators indexed by a run-time block number would live in scratch memory)
    const int li = lane & 31, lh = lane >> 5;
    const int t_begin = chunk * tiles_per_chunk, t_end = min(nrows / TR, t_begin + tiles_per_chunk);
    if (t_begin < t_end) { issue(t_begin); commit(0); if (t_begin + 1 < t_end) issue(t_begin + 1); }
    __syncthreads();
    auto run = [&](auto role_c) {
        constexpr int ROLE = decltype(role_c)::value;
        constexpr int NB = ROLE == 2 ? 1 : 2;
        constexpr int A0 = ROLE == 1 ? 32 : 0, B0 = ROLE == 0 ? 0 : ROLE == 1 ? 32 : ROLE == 2 ? 32 : 96;            // first block
        constexpr bool PART2 = ROLE != 3;                                                                              // first block: re and im parts (Gr)
        constexpr int A1 = ROLE == 0 ? 0 : 32, B1 = ROLE == 0 ? 64 : ROLE == 1 ? 96 : 64;                              // second block (a P1 block)
        v16i acc0[5], acc1[5];
#pragma unroll
        for (int d = 0; d < 5; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[d][r] = 0; acc1[d][r] = 0; }
        auto mac = [&](const unsigned char* T, int acol, int bcol, v16i (&a5)[5]) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                v4i A[4], B[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    A[i] = *reinterpret_cast<const v4i*>(T + i * PLANE + (acol + li) * CS + 32 * ks + 16 * lh);
                    B[i] = *reinterpret_cast<const v4i*>(T + i * PLANE + (bcol + li) * CS + 32 * ks + 16 * lh);
                }
                if (MODE == 1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) a5[i][0] += A[i][0] ^ A[i][1] ^ A[i][2] ^ A[i][3] ^ B[i][0] ^ B[i][1] ^ B[i][2] ^ B[i][3];
                    continue;
                }
                // small digits first; consecutive instructions go to different accumulators
#pragma unroll
                for (int sum = 2; sum <= 6; ++sum)
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const int j = sum - i; if (j < 0 || j > 3) continue; a5[sum - 2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[i], B[j], a5[sum - 2], 0, 0, 0); }
            }
        };
        for (int t = t_begin; t < t_end; ++t) {
            const int cur = (t - t_begin) & 1;
            if (t + 1 < t_end) {
                if (MODE != 2 || t == t_begin) commit(cur ^ 1);
                if (t + 2 < t_end) issue(t + 2);
            }
            const unsigned char* T = smem + cur * BUF;
            mac(T, A0, B0, acc0);
            if (PART2) mac(T, A0 + 64, B0 + 64, acc0);
            if (NB == 2) mac(T, A1, B1, acc1);
            __syncthreads();
        }
        // accumulator register r of lane (j = li, h = lh): row i = (r & 3) + 8 (r >> 2) + 4 h, column j
        constexpr int slot0 = ROLE == 0 ? 0 : ROLE == 1 ? 2 : ROLE == 2 ? 4 : 5;
        int* o = out + (((size_t)site * nchunks + chunk) * 7 + slot0) * 5 * 1024;
#pragma unroll
        for (int d = 0; d < 5; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
                o[d * 1024 + i * 32 + li] = acc0[d][r];
                if (NB == 2) o[5 * 1024 + d * 1024 + i * 32 + li] = acc1[d][r];
            }
    };
    if (w == 0) run(std::integral_constant<int, 0>{}); else if (w == 1) run(std::integral_constant<int, 1>{}); else if (w == 2) run(std::integral_constant<int, 2>{}); else run(std::integral_constant<int, 3>{});
}

int main(int argc, char** argv) {
    const int nsites = argc > 1 ? std::atoi(argv[1]) : 380, nrows = 32768, tpc = argc > 2 ? std::atoi(argv[2]) : 64;
    const int ntiles = nrows / TR, nchunks = (ntiles + tpc - 1) / tpc;
    const size_t per_site = (size_t)NCOL * nrows;
    std::vector<float> h((size_t)per_site);                 // site 0 on the host; the other sites are copies of it on the device (the timing does not care)
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
    for (int c = 0; c < NCOL; ++c) {
        const double cs = std::pow(10.0, -3.0 * rnd());       // columns of very different size, elements over several orders of magnitude inside a column
        for (int r = 0; r < nrows; ++r) { const double g = std::sqrt(-2.0 * std::log(rnd() + 1e-300)) * std::cos(6.283185307179586 * rnd()); h[(size_t)c * nrows + r] = (float)(cs * g * std::pow(10.0, -2.0 * rnd())); }
    }
    std::vector<float> sc(NCOL); std::vector<int> E(NCOL);
    for (int c = 0; c < NCOL; ++c) { float m = 0.f; for (int r = 0; r < nrows; ++r) m = std::fmax(m, std::fabs(h[(size_t)c * nrows + r])); int e; std::frexp(m, &e); E[c] = e; sc[c] = std::ldexp(1.0f, 30 - e); }
    float* dX; float* dS; int* dO;
    CK(hipMalloc(&dX, per_site * sizeof(float) * nsites)); CK(hipMalloc(&dS, sizeof(float) * NCOL * nsites));
    const size_t out_elems = (size_t)nsites * nchunks * 7 * 5 * 1024;
    CK(hipMalloc(&dO, out_elems * sizeof(int)));
    for (int s = 0; s < nsites; ++s) { CK(hipMemcpy(dX + per_site * s, h.data(), per_site * sizeof(float), hipMemcpyHostToDevice)); CK(hipMemcpy(dS + NCOL * s, sc.data(), NCOL * sizeof(float), hipMemcpyHostToDevice)); }
    const size_t lds = 2 * BUF;
    auto run = [&](int mode) {
        const void* f = mode == 0 ? (const void*)int8_gram_kernel<0> : mode == 1 ? (const void*)int8_gram_kernel<1> : (const void*)int8_gram_kernel<2>;
        CK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e30f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(int8_gram_kernel<0>, dim3(nsites * nchunks), dim3(256), lds, 0, dX, dS, dO, nrows, tpc, nchunks);
            else if (mode == 1) hipLaunchKernelGGL(int8_gram_kernel<1>, dim3(nsites * nchunks), dim3(256), lds, 0, dX, dS, dO, nrows, tpc, nchunks);
            else hipLaunchKernelGGL(int8_gram_kernel<2>, dim3(nsites * nchunks), dim3(256), lds, 0, dX, dS, dO, nrows, tpc, nchunks);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) best = std::fmin(best, ms);
        }
        return best;
    };
    const float t_full = run(0);
    // ---- check site 0: assemble G from the integer partials, compare with the f64 Gram of the f32 data on the host ----------------------------------------
    std::vector<int> ho((size_t)nchunks * 7 * 5 * 1024);
    CK(hipMemcpy(ho.data(), dO, ho.size() * sizeof(int), hipMemcpyDeviceToHost));
    // block slots: 0 Gr(0,0)  1 P1(0,0)  2 Gr(1,1)  3 P1(1,1)  4 Gr(0,1)  5 P1(0,1)  6 P1(1,0)
    auto blockval = [&](int slot, int i, int j) {          // integer value sum_d 256^(d+2) acc_d summed over the chunks, as long double
        long double v = 0;
        for (int ch = 0; ch < nchunks; ++ch) for (int d = 0; d < 5; ++d) v += std::ldexp((long double)ho[(((size_t)ch * 7 + slot) * 5 + d) * 1024 + i * 32 + j], 8 * (d + 2));
        return v;
    };
    double worst_q = 0, worst_f = 0, gmax = 0;
    std::vector<double> Gr(64 * 64), P1(64 * 64);
    for (int I = 0; I < 2; ++I) for (int J = 0; J < 2; ++J) for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        const int c = 32 * I + i, cp = 32 * J + j;
        const int ps = (I == 0 && J == 0) ? 1 : (I == 1 && J == 1) ? 3 : (I == 0) ? 5 : 6;
        P1[c * 64 + cp] = (double)std::ldexp(blockval(ps, i, j), E[c] + E[64 + cp] - 60);
        if (I <= J) { const int gs = (I == 0 && J == 0) ? 0 : (I == 1 && J == 1) ? 2 : 4; Gr[c * 64 + cp] = (double)std::ldexp(blockval(gs, i, j), E[c] + E[cp] - 60); }
    }
    // (Gr uses ONE scale per complex column pair in a real kernel; here re and im columns have their own, so the re and im parts are assembled separately for the check)
    for (int c = 0; c < 64; ++c) for (int cp = c; cp < 64; ++cp) {
        long double rr = 0, ii = 0, ri = 0, qrr = 0, qii = 0, qri = 0;
        for (int r = 0; r < nrows; ++r) {
            const float a = h[(size_t)c * nrows + r], b = h[(size_t)cp * nrows + r], ai = h[(size_t)(64 + c) * nrows + r], bi = h[(size_t)(64 + cp) * nrows + r];
            rr += (long double)a * b; ii += (long double)ai * bi; ri += (long double)a * bi;
            const long double qa = std::nearbyint((long double)a * sc[c]), qb = std::nearbyint((long double)b * sc[cp]), qai = std::nearbyint((long double)ai * sc[64 + c]), qbi = std::nearbyint((long double)bi * sc[64 + cp]);
            qrr += qa * qb; qii += qai * qbi; qri += qa * qbi;
        }
        (void)qrr; (void)qii;
        const double got_p1 = P1[c * 64 + cp];
        const double exact_q = (double)std::ldexp(qri, E[c] + E[64 + cp] - 60);
        // against the f64 Gram of the unquantised data, relative to the geometric mean of the two columns' squared norms
        long double na = 0, nb = 0; for (int r = 0; r < nrows; ++r) { na += (long double)h[(size_t)c * nrows + r] * h[(size_t)c * nrows + r]; nb += (long double)h[(size_t)(64 + cp) * nrows + r] * h[(size_t)(64 + cp) * nrows + r]; }
        worst_f = std::fmax(worst_f, std::fabs(got_p1 - (double)ri) / std::sqrt((double)(na * nb)));
        worst_q = std::fmax(worst_q, std::fabs(got_p1 - exact_q) / std::sqrt((double)(na * nb)));
        gmax = std::fmax(gmax, std::fabs((double)ri));
        (void)rr; (void)ii;
    }
    const float t_nomm = run(1), t_noq = run(2);
    const double bytes = (double)nsites * per_site * 4;
    std::printf("{\"sites\": %d, \"rows\": %d, \"tiles_per_chunk\": %d, \"ms_int8_gram\": %.3f, \"TBps\": %.2f, \"ms_without_matrix_instructions\": %.3f, \"ms_without_quantisation\": %.3f, "
                "\"P1_vs_exact_integer_gram_rel\": %.3e, \"P1_vs_f64_gram_of_the_f32_data_normwise\": %.3e, \"note\": \"the library's f64 kernel (gauge transform + Gram): 4.4 ms per 380 sites\"}\n",
                nsites, nrows, tpc, t_full, bytes / t_full * 1e-9, t_nomm, t_noq, worst_q, worst_f);
    return 0;
}
