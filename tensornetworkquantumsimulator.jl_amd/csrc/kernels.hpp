// kernels.hpp -- host-side interface of the HIP kernels (gfx950).  All tensors live in the canonical layout
// [site d][leg_0]...[leg_{z-1}] (column-major, interleaved complex).  Every heavy kernel works on "fiber tiles":
// the tensor is viewed as  element(s, a, k, b) at  s + D*(a + PA*(k + K*b)),  where k is the leg being
// contracted / kept, (s) the site index when it takes part (D = d) or folded into a (D = 1), and (a, b) the
// remaining indices before / after leg k.  No explicit transposes are ever materialised.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime_api.h>
#include <cstddef>
#include <cstdint>

namespace tnqs {

// ---- batched item descriptors (POD, uploaded to the device per launch) -------------------------------------
struct FiberItem {        // out[(s',n),(a,b)] = sum_{(s,k)} in[(s,k),(a,b)] * X[(s,k),(s',n)]
    const void* in; void* out; const void* X;   // X: (D*K) x (Do*No) column-major complex
    int D, PA, K, PB;     // input addressing
    int Do, No;           // output addressing: s' + Do*(a + PA*(n + No*b))
    int TA, TB, nta, ntb; // tile = TA x TB fibers, nta x ntb tiles
    int tile_begin;       // first global tile id of this item (MFMA path: first workgroup id)
    int tpw;              // MFMA path: consecutive tiles walked by one workgroup (generic kernel: 1)
    int want_norm;        // 1: write sum |out|^2 of each tile to norm_partials[global tile id]
};

struct GramItem {         // partial[c][i + KK*j] = sum_{(a,b) in chunk c} X[i,(a,b)] * conj(Y[j,(a,b)]),  i,j = (s,k)
    const void* X; const void* Y; void* partial;
    int D, PA, K, PB;
    int TA, TB, nta, ntb;
    int chunk_begin, nchunks, tiles_per_chunk;
    const void* M;        // fused variant only: 32 x 32 message absorbed on the first row leg of X before the Gram
};

struct ReduceItem {       // out[i + n*j] = sum_c partial[c][..] (optionally conjugated)
    const void* partial; void* out; int n2; int nchunks; int conj; int elem_begin;
};

struct MsgFinalItem {     // BP epilogue: reduce partials, m /= sum(m), message_diff against the previous message
    const void* partial; int nchunks; int chi;
    const void* old_msg;  // may be null => identity
    void* new_msg;
    double* diff_out;     // one double
    int normalize;
};

// BP message of a small site (bp_small_site_kernel): psi in the canonical layout [d][chi_0]..[chi_{z-1}] (<= 8192 elements, every chi <= 32, z <= 8), the message
// entering through leg k (null: unset = identity; M[jo] is ignored), out = the raw chi_jo x chi_jo message [ket + chi bra]
struct SmallMsgItem { const void* psi; void* out; const void* M[8]; int chi[8]; int d, z, jo; int mfma;      // mfma: every leg 16-dimensional and the host allows the matrix-core form
    // new_msg != null (matrix-core form only): the kernel is the message's epilogue as well -- m /= sum(m), message_diff against old_msg (null: identity), as
    // msg_finalize_kernel does from the raw message; `out` is not written then
    const void* old_msg; void* new_msg; double* diff_out; int normalize; };
inline bool bp_small_site_covers(int d, int z, const int* chi, size_t nelem) {
    if (z < 1 || z > 8 || nelem > 8192 || d < 1) return false;
    for (int k = 0; k < z; ++k) if (chi[k] > 32) return false;
    return true;
}
void launch_bp_small_site(hipStream_t s, const SmallMsgItem* d_items, int nitems, int max_elems);
struct JacobiItem {       // one-sided Jacobi on A (m x n, col-major, ld = m); V (n x n) accumulates the rotations (null: not wanted)
    void* A; void* V; int m; int n; int* sweeps_out;
    // dyn != null: the dimensions are decided ON THE DEVICE by an earlier kernel of the same stream (the theta of a gate: ranks of the two
    // R factors) and read from its info array (GateItem::info) -- m, n above are then upper bounds the host sized the launch with.  This
    // is what lets a gate batch run from the Gram matrices to the truncated factors without a host round trip in between.
    const int* dyn; int dm, dn;
    int nhint;            // host only: the column count the item is EXPECTED to have when dyn decides it (0: n)
    const int* only_if;   // global-memory kernel only: skip the item unless *only_if != 0 (null: always run)
    // Preconditioned route of a low-rank theta (theta_svd_pre_kernel): QB = the orthonormal factor Q of theta = M Q^T ((r2 d2) x K, complex128), Vout = where the
    // right singular vectors of theta go ((r2 d2) x K, data precision).  pre != 0: the item has ALSO been handed to that kernel, which takes it when the
    // dimensions found on the device fit (theta_pre_takes); the plain Jacobi kernel and the V recovery then skip it
    const void* QB; void* Vout; int pre;     // pre: 1 = offered, theta as it stands; 2 = offered, low-rank factor with Q
    int cap;                                 // columns that can survive the truncation (0: all): U Sigma and V are only formed for the `cap` largest singular values
};
// device-side decision shared by theta_svd_pre_kernel, jacobi_lds_kernel / jacobi_kernel and the V-recovery kernels: the matrix the SVD runs on -- the low-rank
// factor when that route survived on the device (info[7] = K > 0; needs Q), theta itself otherwise -- fits the kernel
#ifndef TNQS_PRE_MIN_COLS
#define TNQS_PRE_MIN_COLS 16      // the preconditioned kernel takes factors of MORE than this many columns (csrc/build.sh EXTRA_FLAGS=-DTNQS_PRE_MIN_COLS=32: the round-5 start)
#endif
__host__ __device__ inline bool theta_pre_takes(const int* info, int d1, int d2, bool have_q) {
    if (!info) return false;
    int Mr = info[0] * d1, Nc = info[1] * d2;
    if (info[7] > 0) {                                  // low-rank route alive: the K-column factor, V through Q
        const int K = info[7];
        return have_q && Mr >= Nc && K > TNQS_PRE_MIN_COLS && K <= 64 && Mr >= K && Mr <= 128;
    }
    if (Mr < Nc) { const int t = Mr; Mr = Nc; Nc = t; }   // theta itself (stored as its adjoint when wide): V = U_L
    return Nc > TNQS_PRE_MIN_COLS && Nc <= 64 && Mr <= 128;
    // (a round of the sweeps costs ~0.7 us whatever the row count -- latency of the load / reduce / rotate / barrier chain.  On ISOLATED 64 x 32 factors of the
    //  benchmark state the plain kernel's 7-8 sweeps of 31 rounds, 0.14-0.16 ms, beat 6-7 preconditioned sweeps plus 37 us of Gram, Cholesky and products (0.18-0.21 ms,
    //  profiles/svd_bench.py), which is why the limit was 32 columns at first; inside the heavy-hex layer (64 x 32 factors of every gate, 0.2-0.3 ms per batch in the
    //  plain kernel) the preconditioned kernel wins: Jacobi class 1.17 -> 0.96 ms per layer, layer 3.8 -> 3.6 ms, and the 7 x 7 lattice 15.9 -> 15.8.  At 64 columns:
    //  0.42-0.47 ms against 0.27-0.31)
}
// dimensions of a gate's theta SVD from its info array (gate_theta_kernel): rows, columns of theta, columns the Jacobi runs on
__host__ __device__ inline void theta_dims(const int* info, int d1, int d2, int& m, int& nfull, int& ncol) {
    int Mr = info[0] * d1, Nc = info[1] * d2;
    if (Mr < Nc) { const int t = Mr; Mr = Nc; Nc = t; }        // a wide theta is stored as its adjoint
    m = Mr; nfull = Nc; ncol = info[7] > 0 ? info[7] : Nc;     // low-rank route: the SVD runs on M (Mr x K)
}

struct EnvItem {          // env message -> Hermitian f64 matrix H (= (M + M^dagger)/2), V = I
    const void* msg; void* H; void* V; int n;
};
struct EnvFinishItem {    // (H rotated, V) -> M^{1/2} and P = M^{1/2} M^{-1/2} in data precision
    const void* A; const void* V; void* msqrt; void* proj; int n; double cutoff; int* flags; // flags[0]: full rank, flags[1]: error
};

struct GateItem {         // everything the per-gate small-algebra kernels need (pointers into the workspace)
    // Gram factors: Jacobi output for G1, G2 (f64): A = G V, V
    const void* GA1; const void* GV1; const void* GA2; const void* GV2;
    // pseudo-inverse factor W with R^+ = W diag(lambda^-1/2): the eigenvectors again, or (Cholesky sites) the conjugate-transposed
    // inverse triangle; chol != 0: lambda = 1 and every column is kept (GA unused, GV = L with R = L^dagger)
    const void* GW1; const void* GW2; int chol1, chol2;
    int n1, n2;           // d1*chi, d2*chi
    int d1, d2, chi;      // bond dim before the gate
    const double* gate;   // (d1 d2) x (d1 d2) complex128 column-major, first vertex most significant
    // outputs / scratch
    double* lam1; double* lam2;     // kept eigenvalues (n1 / n2 doubles)
    int* idx1; int* idx2;           // kept eigen-column indices
    void* theta; void* thetaV;      // (r1 d1) x (r2 d2) in data precision (+ V of its SVD), ld = r1*d1
    void* theta0;                   // optional second copy of theta (kept unrotated for the V recovery), may be null
    void* X1; void* X2;             // n1 x (d1 chi') , n2 x (d2 chi') in data precision (allocated for chi' <= chi_cap)
    double* S;                      // chi_cap reals
    int* info;                      // [0]=r1 [1]=r2 [2]=chi' [3]=status [4]=svd sweeps [5]=wide [6]=bit 0 / 1: site 1 / 2 is ill-conditioned [7]=columns of the SVD input (low-rank route) or 0
    double* truncerr;               // one double
    int maxdim; double cutoff; int normalize; int chi_cap;
    // per-site rank threshold of the eigen route (rank_tau; negative: shifted first pass of a site that a second factorisation pass
    // can follow, gate_eigs); chol == 2: the site carries an
    // explicit factor from that second pass (GV = R^dagger, GW = R^+, lambda = 1, *rk columns) -- see Qr2ComposeItem
    double tau1, tau2; const int* rk1; const int* rk2;
    // low-rank route of the theta SVD (ComplexF32): the gate as an operator sum  g = sum_k a_k (x) b_k  (kappa terms; opA: kappa x d1 x d1,
    // opB: kappa x d2 x d2 complex128, [k][s' + d s]) makes theta = A B^T with K = kappa chi columns in A ((r1 d1) x K) and B ((r2 d2) x K).
    // When K < r2 d2, theta is not wide and chi_cap <= K, gate_theta also writes A, B and G = B^dagger B (lowA / lowB / lowG, complex128);
    // chol_kernel factors G = L L^dagger and lowrank_m_kernel overwrites the first K columns of theta with M = A conj(L), whose left
    // singular vectors and singular values are theta's (theta = M Q^T, Q = B L^-dagger orthonormal).  info[7] = columns the SVD runs on.
    int kappa; const double* opA; const double* opB; void* lowA; void* lowB; void* lowG; const void* lowL; const int* lowfail;
    // theta (or its low-rank factor) and theta0 are scaled by 2^(*texp) so that their largest entry is O(1) (theta_scale_kernel): the f32
    // SVD pipeline squares and multiplies these entries; gate_finish puts the factor back into the singular values
    int* texp;
    // low-rank route, preconditioned theta SVD (round 5): W = L^-dagger of the Cholesky factor of B^dagger B, and Q = B W (written by lowrank_m_kernel)
    const void* lowW; void* lowQ;
};
template <class T> void launch_theta_scale(hipStream_t s, const GateItem* d_items, int nitems);
void launch_lowrank_g(hipStream_t s, const GateItem* d_items, int nitems);
template <class T> void launch_lowrank_m(hipStream_t s, const GateItem* d_items, int nitems);
// ComplexF64 low-rank route: B (Nc x K) is orthogonalised by CholeskyQR2 -- a single Gram / Cholesky pass squares the condition number, which
// f32 data do not notice and f64 data do.  Pass 1: G1 = B^dagger B = L1 L1^dagger, B1 = B W1 with W1 = L1^-dagger; pass 2: G2 = B1^dagger B1 =
// L2 L2^dagger; then B = Q (L1 L2)^dagger with Q orthonormal to eps and theta = A B^T = (A conj(L1 L2)) Q^T.  bw: B1 = B W1;  ll: Lc = L1 L2 and the
// failure flag of pass 2 folded into the gate's (a failed pass sends the gate back to the SVD of the full theta).
struct LowQr2Item { const void* B; const void* W1; void* B1; const void* L1; const void* L2; void* Lc; const int* info; int d2; int* fail1; const int* fail2; };
void launch_lowrank_bw(hipStream_t s, const LowQr2Item* d_items, int nitems);
void launch_lowrank_ll(hipStream_t s, const LowQr2Item* d_items, int nitems);
// Second factorisation pass of an ill-conditioned ComplexF64 site (CholeskyQR2).  With the first-pass factor R1 (interface of
// GateItem: R1[a,(s,b)] = sqrt(l_a) conj(GV[(s,b), idx_a]), R1^+[:,a] = GW[:, idx_a] / sqrt(l_a), r kept columns):
//   Qr2RinvItem:    X1 = R1^+ as an explicit n x n matrix (columns >= r zero), so that Q1 = psi~ x_(s,b) X1 can be formed;
//   Qr2ComposeItem: with the Jacobi eigen factorisation (A2 = G2 V2, V2) of G2 = Q1^dagger Q1 = W2 L2 W2^dagger, R2 = L2^1/2 W2^dagger:
//                   GVout = (R2 R1)^dagger, GWout = R1^+ R2^+ (n x n, the first *rk columns valid), *rk = number of l2 above tau.
struct Qr2RinvItem { const void* GW; const double* lam; const int* idx; const int* r; int n; void* X1; };
struct Qr2ComposeItem { const void* A2; const void* V2; const void* X1; const void* GV1; const double* lam1; const int* idx1; const int* r1;
                        int n; double tau; void* GVout; void* GWout; int* rk; };
void launch_qr2_rinv(hipStream_t s, const Qr2RinvItem* d_items, int nitems);
void launch_qr2_compose(hipStream_t s, const Qr2ComposeItem* d_items, int nitems);

struct Site1Item { const void* in; void* out; float g[8]; size_t npairs; };   // d = 2 one-site gate: g = (g00, g01, g10, g11) re/im
struct DiagItem { void* out; const double* S; int chi; };          // dense diag(S) message
struct ScaleItem { const void* src; void* dst; size_t n; const double* factor; };      // dst = src * (*factor)
struct NormFactorItem { const double* norm_partials; int npart; double* factor; };      // *factor = 1/sqrt(sum partials)  (1 when the sum is not positive)
// rescale_messages! (beliefpropagationcache.jl:127-140) for one edge: both messages normalised to unit Frobenius norm, then divided by
// sqrt(n), n = sum_ij me[i,j] mer[i,j] (sign folded into me when n is exactly real); null inputs = identity (tensornetworkstate.jl:72-75)
struct MsgRescaleItem { const void* me; const void* mer; void* me_out; void* mer_out; int chi; };
template <class T> void launch_msg_rescale(hipStream_t s, const MsgRescaleItem* d_items, int nitems);
// edge_scalar (beliefpropagationcache.jl:47-49): out[e] = sum_ij me[i,j] mer[i,j] (complex128)
struct EdgeScalarItem { const void* me; const void* mer; int chi; double* out; };
template <class T> void launch_edge_scalar(hipStream_t s, const EdgeScalarItem* d_items, int nitems);
// dst = src * (re + i im)
struct CScaleItem { const void* src; void* dst; size_t n; double re, im; };
template <class T> void launch_cscale(hipStream_t s, const CScaleItem* d_items, int nitems);
// symmetric_gauge! (src/symmetric_gauge.jl:1-62) per edge, n = chi.  build: from the f64 Jacobi factorisations (A = H V, V) of both
// messages: irx, iry = conj((M + reg)^-1/2) (f64, n x n) and Ce = conj((X+reg)^1/2) conj((Y+reg)^1/2)^T in data precision (two copies:
// Ce is rotated by the SVD, Ce0 stays).  finish: from U Sigma (in Ce) and V: S (descending), Xs = irx U S^1/2, Xd = iry conj(V) S^1/2.
struct SymGaugeItem { const void* AX; const void* VX; const void* AY; const void* VY; void* rx; void* ry; void* irx; void* iry;
                      void* Ce; void* Ce0; void* Vsvd; void* Xs; void* Xd; double* S; int n; double reg; int* flag; };
template <class T> void launch_symg_build(hipStream_t s, const SymGaugeItem* d_items, int nitems);
template <class T> void launch_symg_finish(hipStream_t s, const SymGaugeItem* d_items, int nitems);
// sharded gate batches: the (chi', status, truncerr | S | X2) record of a gate packed into its exchange slot, and the 32-byte
// headers of all records gathered into one contiguous array (one launch each instead of three copies per gate)
struct RecordPackItem { void* dst; const int* info; const double* terr; const double* S; int nS; const void* X2; long long x2_words; long long x2_off; };
void launch_record_pack(hipStream_t s, const RecordPackItem* d_items, int nitems);
void launch_header_gather(hipStream_t s, const void* const* d_srcs, int n, double* d_out);
struct PermItem { const void* in; void* out; int ndim; int dims_out[8]; long long stride_in[8]; size_t n; };

// ---- launchers (T = float or double; data are complex<T>) ------------------------------------------------------
template <class T> void launch_fiber_gemm(hipStream_t s, const FiberItem* d_items, int nitems, int total_tiles,
                                          int TR, int KKmax, double* d_norm_partials);
template <class T, class Acc> void launch_gram(hipStream_t s, const GramItem* d_items, int nitems, int total_chunks,
                                               int TR, int KKmax);
template <class Acc, class Out> void launch_reduce(hipStream_t s, const ReduceItem* d_items, int nitems, int total_elems);
template <class T> void launch_msg_finalize(hipStream_t s, const MsgFinalItem* d_items, int nitems);
struct RecoverItem { const void* A0; const void* A; void* V; int m; int n; int nu; const int* dyn; int dm, dn; int pre; /* JacobiItem::pre of the same gate: skip when theta_pre_takes */ };   // dyn: as in JacobiItem   // V (n x nu) = A0^dagger (U Sigma) Sigma^-2; A0: m x n, U Sigma: m x nu
// LDS bytes the LDS-resident Jacobi needs for an m x n matrix (columns padded by 2 elements)
inline size_t jacobi_lds_bytes(int m, int n, bool withV, size_t esz) { return ((size_t)(m + 2) * n + (withV ? (size_t)(n + 2) * n : 0)) * esz; }
template <class T> void launch_jacobi(hipStream_t s, const JacobiItem* d_items, int nitems, int max_sweeps, size_t lds_bytes, int mmax, int ncols = 0);   // ncols: expected columns (sizes the workgroup of the LDS kernel)
// Preconditioned theta SVD in one kernel (kernels.hip theta_svd_pre_kernel): ComplexF32, no V, m >= n, n <= 64, m <= 128 (upper bounds when JacobiItem::dyn
// decides the dimensions on the device).  LDS: the sorted A (f32) + the n x n Gram / Cholesky array (f64)
inline bool theta_svd_pre_covers(int m, int n) { return n >= 2 && n <= 64 && m >= n && m <= 128; }
inline size_t theta_svd_pre_lds_bytes(int m, int n) { return ((((size_t)(m + 2) * n * 8) + 15) & ~(size_t)15) + (size_t)(n + 1) * n * 16; }
void launch_theta_svd_pre(hipStream_t s, const JacobiItem* d_items, int nitems, int max_sweeps, int mmax, int nmax);
// sites with fewer fibers than columns (N < n = d*chi): the R factor comes from a one-sided Jacobi SVD of the n x N matrix
// M[(s,b), outer] = conj(psi~[outer,(s,b)]) (f64) instead of the eigen factorisation of the rank-deficient n x n Gram matrix
struct SmallSvdItem { const void* src; void* M; void* GA; void* GV; int d, low, chi_b, hi; };   // low = pre(b)/d, hi = post(b); n = d*chi_b, N = low*hi
template <class T> void launch_small_svd_prepare(hipStream_t s, const SmallSvdItem* d_items, int nitems);
void launch_small_svd_finish(hipStream_t s, const SmallSvdItem* d_items, int nitems);
// Cholesky of a Hermitian positive definite n x n f64 matrix (column-major): L (lower, G = L L^dagger) and Winv = (L^-1)^dagger;
// *fail is set when a pivot drops below 1e-12 * max diagonal (the caller falls back to the eigen factorisation)
// Rank threshold of the Gram-matrix factorisations: a direction of psi~ whose Gram eigenvalue (Cholesky pivot) is below tau * max is
// treated as null.  tau is what the f64 Gram matrix can resolve -- not more: for ComplexF32 states the f32 rounding of psi~ itself
// (6e-8 in amplitude, 4e-15 n in the eigenvalue) sits under 1e-12; for ComplexF64 states the f64 accumulation leaves ~n eps, and the
// reference's QR keeps everything above that (a flat 1e-12 here dropped singular directions of relative size < 1e-6 and cost 1e-9
// per layer in log Z on the reference's thermal-state example, cutoff = 1e-14).
__host__ __device__ inline double rank_tau(bool f32_state, int n) { return f32_state ? 1e-12 : 4.0 * (double)n * 2.220446049250313e-16; }
struct CholItem { const void* G; void* L; void* Winv; int n; int* fail; double tau; double shift; };      // shift (packed kernel only): G + shift * max diag * I is factorised
// Cholesky-QR preprocessing of a tall theta for the LDS-resident Jacobi (kernels_chi64.hip): A m x n (ComplexF32, ld = m)
struct TallSvdItem { const void* A; void* G; const void* L; void* R0; void* Rrot; int m, n; };
struct SmallGemmItem { const void* A; const void* B; void* C; int m, n, k; };           // C (m x n) = A (m x k) B (k x n), ComplexF32
struct CopyItem { const void* src; void* dst; size_t n16; };                             // n16 16-byte words
// register-direct MFMA fiber GEMM for chi = 64 sites (kernels_chi64.hip): rowgemm_tiles() sets the tile grid of an item
bool rowgemm_covers(const FiberItem& it);
void rowgemm_tiles(FiberItem& it);
void launch_mfma_rowgemm(hipStream_t s, const FiberItem* d_items, int nitems, int total_wgs, int D, int K, double* d_norm_partials);   // all items: the same K and D
bool launch_x3_gram64(hipStream_t s, const GramItem* d_items, int nitems, int total_chunks, int KKmax);                               // kernels_x3.hip: mfma_gram64_kernel on the bf16 matrix cores
void launch_x3_rowgemm64(hipStream_t s, const FiberItem* d_items, int nitems, int total_wgs, int D, double* d_norm_partials);              // kernels_x3.hip: D K = 64 on the bf16 matrix cores
void launch_tall_gram(hipStream_t s, const TallSvdItem* d_items, int nitems, int nmax);
void launch_tall_rt(hipStream_t s, const TallSvdItem* d_items, int nitems);
void launch_tall_w(hipStream_t s, const TallSvdItem* d_items, int nitems, int nmax);      // R0 slot (out, complex128) = [L slot: R^-1, complex128] x Rrot
void launch_tall_mj(hipStream_t s, const SmallGemmItem* d_items, int nitems);      // C (m x n, ComplexF32) = A (m x k, ComplexF32) B (k x n, complex128), f64 matrix cores
void launch_copy_items(hipStream_t s, const CopyItem* d_items, int nitems);
void launch_chol(hipStream_t s, const CholItem* d_items, int nitems, int nmax);
void launch_chol_packed(hipStream_t s, const CholItem* d_items, int nitems, int nmax);      // n <= 128 (packed triangle in LDS); Winv may be null
template <class T> void launch_recover_v(hipStream_t s, const RecoverItem* d_items, int nitems, int nmax);
void launch_recover_v_mfma(hipStream_t s, const RecoverItem* d_items, int nitems, int nmax);    // ComplexF32, MFMA (kernels_mfma.hip)
template <class T> void launch_env_prepare(hipStream_t s, const EnvItem* d_items, int nitems);
template <class T> void launch_env_finish(hipStream_t s, const EnvFinishItem* d_items, int nitems);
template <class T> void launch_gate_theta(hipStream_t s, const GateItem* d_items, int nitems);
template <class T> void launch_gate_theta_mm(hipStream_t s, const GateItem* d_items, int nitems);      // theta = A B^T from the operator-sum factors (after gate_theta)
template <class T> void launch_gate_finish(hipStream_t s, const GateItem* d_items, int nitems);
template <class T> void launch_diag(hipStream_t s, const DiagItem* d_items, int nitems);
template <class T> void launch_scale(hipStream_t s, const ScaleItem* d_items, int nitems);
void launch_norm_factor(hipStream_t s, const NormFactorItem* d_items, int nitems);
template <class T> void launch_permute(hipStream_t s, const PermItem& item);
template <class T> void launch_identity(hipStream_t s, void* out, int n);
template <class T> void launch_random_fill(hipStream_t s, void* out, size_t n, unsigned long long seed, double scale, bool real_only);   // iid normal entries, counter-based
void launch_sum_doubles(hipStream_t s, const double* in, int n, double* out);
// one-site gates on d = 2, ComplexF32: streaming 2x2 apply, norm partials [item][nbx]
void launch_site1_c64(hipStream_t s, const Site1Item* d_items, int nitems, int nbx, double* d_norm_partials);

// geometry shared by the two "pair" kernels: a 32 x 32 plane over two legs (strides sx, sy) for 16 companion elements
// = 8 companion PAIRS of 2 contiguous elements, pair f at offset f*cstr.  Slices are enumerated by three counters:
// slice sl -> a0 = sl % n0, a1 = (sl / n0) % n1, a2 = sl / (n0*n1), base = a0*t0 + a1*t1 + a2*t2   (elements).
//  * both legs above the site index: the companions are 16 CONTIGUOUS elements (cstr = 2, one 128-byte run)
//  * leg 0 involved (only the site index below it): the 2 site components are the contiguous pair and the 8 pairs step
//    along another leg a (cstr = pre(a)); then consecutive plane rows ix of leg 0 are contiguous (8 x 16 B = 128-byte runs)
struct PairGeom { long long cstr, sx, sy, t0, t1, t2; int n0, n1, n2; };
// geometry for the legs (lx -> plane index ix, ly -> plane index iy) of a site tensor [d][chi_0]..[chi_{z-1}];
// false when the pair kernels do not cover the shape (both legs must have dimension 32)
inline bool pair_geometry(int d, int z, const int* chi, int lx, int ly, PairGeom& g) {
    if (lx == ly || lx < 0 || ly < 0 || lx >= z || ly >= z || chi[lx] != 32 || chi[ly] != 32) return false;
    auto pre = [&](int j) { long long p = d; for (int i = 0; i < j; ++i) p *= chi[i]; return p; };
    const int p = lx < ly ? lx : ly, q = lx < ly ? ly : lx;
    g.sx = pre(lx); g.sy = pre(ly);
    if (pre(p) % 16 == 0) {            // 16 contiguous companions below the lower leg
        g.cstr = 2; g.n0 = (int)(pre(p) / 16); g.t0 = 16;
        g.n1 = (int)(pre(q) / (pre(p) * 32)); g.t1 = pre(p) * 32;
        long long post = 1; for (int i = q + 1; i < z; ++i) post *= chi[i];
        g.n2 = (int)post; g.t2 = pre(q) * 32;
        return true;
    }
    if (pre(p) != 2) return false;     // leg 0 above a 2-dimensional site index: companions = (s) x 8 values of another leg
    int a = -1;
    for (int i = 0; i < z; ++i) if (i != p && i != q && chi[i] % 8 == 0) { a = i; break; }
    if (a < 0) return false;
    g.cstr = pre(a); g.n0 = chi[a] / 8; g.t0 = 8 * pre(a);
    g.n1 = 1; g.t1 = 0; g.n2 = 1; g.t2 = 0;
    int k = 0;
    for (int i = 0; i < z; ++i) {
        if (i == p || i == q || i == a) continue;
        if (k == 0) { g.n1 = chi[i]; g.t1 = pre(i); } else if (k == 1) { g.n2 = chi[i]; g.t2 = pre(i); } else return false;
        ++k;
    }
    return true;
}
struct PairItem {         // out = in x_x Mx x_y My for two 32-dimensional legs
    const void* in; void* out; const void* Mx; const void* My;
    PairGeom g;
    int slice_begin;      // first workgroup id of this item (a multiple of 16)
    int spw;              // slices (16 companions x 32 x 32) walked by one PAIR of workgroups (each takes 8 of the 16 companions)
};
struct PairGramItem {     // partial[b,b'] = sum_{rest, jx} (sum_ix X[.. ix .. b ..] M[ix, jx]) conj(Y[.. jx .. b' ..]);  x = absorbed leg, y = kept leg
    const void* X; const void* Y; const void* M; void* partial;   // 8 partials (one per wave) per workgroup, 32*32 complex each
    PairGeom g;
    int wg_begin;         // first workgroup id of this item
    int spw;              // slices per workgroup
};

// ---- 16-dimensional planes (kernels_plane.hip): the per-site shape of BASELINE configs[3] (degree 6, chi = 16) and of heavy-hex ------
// Same idea as PairGeom for two legs of dimension DIM = 16, with one more slice counter (a degree-6 site has four legs besides the
// plane): slice sl -> a0 = sl % n0, a1 = (sl / n0) % n1, a2 = (sl / (n0 n1)) % n2, a3 = sl / (n0 n1 n2), base = sum a_i t_i.
// A slice holds 16 companion elements = 8 (re, im) PAIRS of 2 contiguous elements, pair f at f * cstr, and is processed as two
// half slices of 8 companions (64-byte runs) by two waves of one workgroup at the same time.
struct PlaneGeom { long long cstr, sx, sy, t0, t1, t2, t3; int n0, n1, n2, n3; int nslices() const { return n0 * n1 * n2 * n3; } };
inline bool plane_geometry(int d, int z, const int* chi, int lx, int ly, int DIM, PlaneGeom& g) {
    if (lx == ly || lx < 0 || ly < 0 || lx >= z || ly >= z || chi[lx] != DIM || chi[ly] != DIM) return false;
    auto pre = [&](int j) { long long p = d; for (int i = 0; i < j; ++i) p *= chi[i]; return p; };
    const int p = lx < ly ? lx : ly, q = lx < ly ? ly : lx;
    g.sx = pre(lx); g.sy = pre(ly);
    g.n1 = g.n2 = g.n3 = 1; g.t1 = g.t2 = g.t3 = 0;
    if (pre(p) % 16 == 0) {            // 16 contiguous companions below the lower leg
        g.cstr = 2; g.n0 = (int)(pre(p) / 16); g.t0 = 16;
        g.n1 = (int)(pre(q) / (pre(p) * DIM)); g.t1 = pre(p) * DIM;
        long long post = 1; for (int i = q + 1; i < z; ++i) post *= chi[i];
        g.n2 = (int)post; g.t2 = pre(q) * DIM;
        return true;
    }
    if (pre(p) != 2) return false;     // leg 0 above a 2-dimensional site index: companions = (s) x 8 values of another leg
    int a = -1;
    for (int i = 0; i < z; ++i) if (i != p && i != q && chi[i] % 8 == 0) { a = i; break; }
    if (a < 0) return false;
    g.cstr = pre(a); g.n0 = chi[a] / 8; g.t0 = 8 * pre(a);
    int k = 0;
    for (int i = 0; i < z; ++i) {
        if (i == p || i == q || i == a) continue;
        if (k == 0) { g.n1 = chi[i]; g.t1 = pre(i); } else if (k == 1) { g.n2 = chi[i]; g.t2 = pre(i); } else if (k == 2) { g.n3 = chi[i]; g.t3 = pre(i); } else return false;
        ++k;
    }
    return true;
}
struct Pair16Item {       // out = in x_x Mx x_y My for two 16-dimensional legs
    const void* in; void* out; const void* Mx; const void* My;
    PlaneGeom g;
    int wg_begin;         // first workgroup id of this item
    int spw;              // slices walked by one workgroup (a multiple of 4: the 8 waves take 4 slices x 2 halves at a time)
};
// both messages a site sends into one linear forest from one pass over (X = shared partial product, Y = psi), 16 x 16 plane (lx, ly):
//   partial_y[b,b'] = sum (X x_lx Mx)[.. b on ly ..] conj(Y[.. b' on ly ..]),   partial_x[d,d'] = sum (X x_ly My)[.. d on lx ..] conj(Y[.. d' on lx ..])
// one 16 x 16 partial per workgroup each
struct PairGram2x16Item { const void* X; const void* Y; const void* Mx; const void* My; void* partial_y; void* partial_x; PlaneGeom g; int wg_begin; int spw; };
// two kernels: a wave pair sharing every 128-byte line (planes that contain leg 0: a wave's lanes read 256 contiguous bytes anyway) and one
// wave owning whole lines (the 16 companions are 16 consecutive elements); a launch holds items of one kind
bool pair16_whole_lines(const PlaneGeom& g);
void launch_mfma_pair16(hipStream_t s, const Pair16Item* d_items, int nitems, int total_wgs, bool whole_lines);
int pair_gram2x16_slices_at_a_time();
void launch_mfma_pair_gram2x16(hipStream_t s, const PairGram2x16Item* d_items, int nitems, int total_wgs);

// ---- MFMA fast paths (ComplexF32 only; kernels_mfma.hip) -----------------------------------------------------------
int mfma_fiber_tile_rows(int KK, int NN);     // fibers per tile for the shape, 0 = not covered
bool launch_mfma_fiber_gemm(hipStream_t s, const FiberItem* d_items, int nitems, int total_tiles, int KKmax, int NNmax,
                            double* d_norm_partials);
bool launch_mfma_gram32(hipStream_t s, const GramItem* d_items, int nitems, int total_chunks, int KKmax);  // tiles of 64 fibers; writes 4 partials per chunk
bool launch_mfma_gram64(hipStream_t s, const GramItem* d_items, int nitems, int total_chunks, int KKmax);  // 32 < KK <= 64 (kernels_chi64.hip): ONE partial per chunk
// fused (X x_r M) then Gram with Y: tiles of 64 fibers = (s:2) x (first row leg: 32); writes 4 partials per chunk
void launch_mfma_gram32_fused(hipStream_t s, const GramItem* d_items, int nitems, int total_chunks);
// Gram with f64 accumulation on the f64 matrix cores (gate path: G = psi~^dagger psi~, D*K == 64, X == Y); tiles of 64 fibers
bool launch_mfma_gram64_f64(hipStream_t s, const GramItem* d_items, int nitems, int total_chunks, int KKmax, bool all_kk64 = false);   // all_kk64: every item has D * K = 64
// the third gauge leg absorbed inside the f64 Gram (kernels_gate.hip): GramItem::M = the 32 x 32 matrix of the fastest outer leg `rleg`; same
// tiles and partial layout as launch_mfma_gram64_f64 (2 partials per chunk)
// ComplexF64 mode products on the f64 matrix cores (kernels_f64.hip): one wave per tile of 16 fibers, FiberItem::tpw tiles per wave-quartet
bool fiber_gemm_f64_covers(const FiberItem& it);
void fiber_gemm_f64_tiles(FiberItem& it);
void launch_mfma_fiber_gemm_f64(hipStream_t s, const FiberItem* d_items, int nitems, int total_wgs, int Kmax, int Nmax, double* d_norm_partials, bool general);
bool gram_f64in_covers(int D, int K);          // ComplexF64 Gram over tiles of 32 fibers (one partial per chunk)
void launch_mfma_gram_f64in(hipStream_t s, const GramItem* d_items, int nitems, int total_chunks);
bool gauge_gram64_covers(int d, int z, const int* chi, int bleg, int rleg);
bool gauge_gram32_covers(int d, int z, const int* chi, int bleg, int rleg);      // the same fusion for 16-dimensional legs (32 columns)
int gauge_gram32_units(int z, const int* chi, int bleg);                         // units (one fiber of r = 16 rows) of a site
void launch_mfma_gauge_gram32(hipStream_t s, const GramItem* d_items, int nitems, int total_chunks);
void launch_mfma_gauge_gram64(hipStream_t s, const GramItem* d_items, int nitems, int total_chunks);
// the same for 64 < D*K <= 128 (chi = 64 sites; kernels_chi64.hip); writes ONE partial per chunk
bool launch_mfma_gram128_f64(hipStream_t s, const GramItem* d_items, int nitems, int total_chunks, int KKmax, bool all_kk128 = false);   // all_kk128: every item has D * K = 128
// fused pair of mode products on two slow 32-dim legs (16 companions = 128-byte runs per workgroup)
void launch_mfma_pair(hipStream_t s, const PairItem* d_items, int nitems, int total_wgs);
void launch_x3_pair(hipStream_t s, const PairItem* d_items, int nitems, int total_wgs);                  // kernels_x3.hip: the same pass on the bf16 matrix cores
// workgroups of one PairItem: ceil(nslices / spw) slice ranges in groups of 8, two workgroups (one per 8-companion half) per range
inline int pair_wgs(int nslices, int spw) { const int np = (nslices + spw - 1) / spw; return 16 * ((np + 7) / 8); }
// slices per workgroup for a batch of `total_slices`: the largest power of two <= 16 that still gives >= 1024 workgroups
inline int pair_spw(double total_slices) {
#ifdef TNQS_EXPERIMENTS
    static const int forced = [] { const char* e = std::getenv("TNQS_PAIR_SPW"); return e ? std::atoi(e) : 0; }();
    if (forced > 0) return forced;
#endif
    int spw = 16; while (spw > 1 && 2.0 * total_slices / spw < 1024.0) spw >>= 1; return spw;
}
// both messages a site sends into one linear forest in ONE pass over (X, Y): legs (lx, ly) span the plane;
//   partial_y[b,b'] = sum (X x_lx Mx)[.. b on ly ..] conj(Y[.. b' on ly ..])      (message leaving through ly: lx absorbed with Mx)
//   partial_x[d,d'] = sum (X x_ly My)[.. d on lx ..] conj(Y[.. d' on lx ..])      (message leaving through lx: ly absorbed with My)
struct PairGram2Item { const void* X; const void* Y; const void* Mx; const void* My; void* partial_y; void* partial_x; PairGeom g; int wg_begin; int spw; };
void launch_mfma_pair_gram2(hipStream_t s, const PairGram2Item* d_items, int nitems, int total_wgs);
void launch_x3_pair_gram2(hipStream_t s, const PairGram2Item* d_items, int nitems, int total_wgs);      // kernels_x3.hip: the same pass on the bf16 matrix cores
void launch_x3_pair_gram1(hipStream_t s, const PairGram2Item* d_items, int nitems, int total_wgs);     // kernels_x3.hip: ONE message per item (My = partial_x = null)
int pair_gram2_group();      // workgroups of one group of 8 slice ranges: 32 (a workgroup walks one quarter of each slice)
// last absorption + Gram on two arbitrary 32-dim legs (absorbed leg x, kept leg y), reading a (cached) pair product X and psi = Y
void launch_mfma_pair_gram(hipStream_t s, const PairGramItem* d_items, int nitems, int total_wgs);

}  // namespace tnqs
