/* tnqs_debug.h -- kernel-level test entry points of libtnqs_hip.so (used by tests/ only; not part of the drop-in
 * boundary).  Host pointers, column-major interleaved complex, synchronous. */
#ifndef TNQS_DEBUG_H
#define TNQS_DEBUG_H
#include <stdint.h>
#include "tnqs.h"
#ifdef __cplusplus
extern "C" {
#endif
/* one-sided Jacobi on A (m x n): on return A = U*Sigma (columns), V (n x n) with A_in = (U Sigma) V^dagger. dtype: 0 c64, 1 c128 */
int tnqs_dbg_jacobi(int dtype, int m, int n, void* A_inout, void* V_out, int* sweeps_out);
/* preconditioned theta SVD kernel (kernels.hip theta_svd_pre_kernel) on one ComplexF32 factor A (m x n, 2 <= n <= 64, n <= nq <= m <= 128) of theta = A Q^T with Q
 * (nq x n complex128, orthonormal columns): on return A = U Sigma (columns, any order), V (nq x n ComplexF32) = right singular vectors of theta in the same column
 * order.  reps > 0: also the average duration (ms, HIP events) of a launch over `copies` device-resident copies */
int tnqs_dbg_theta_svd_pre(int m, int n, int nq, void* A_inout, const void* Q, void* V_out, int* sweeps_out, int copies, int reps, double* ms_out, double* phase_us_out /* 6 doubles or NULL: load, Gram, Cholesky + conversion, sweeps, U Sigma, V */,
                           int cap /* > 0: only the cap largest singular triplets are formed, the other columns of A leave as sigma_j e_0; Q == NULL: A is theta itself, V (n x n) = its right singular vectors */);
/* timing of the plain LDS-resident ComplexF32 Jacobi (no V) on `copies` copies of A (m x n): average launch duration and the sweeps it took */
int tnqs_dbg_time_jacobi_f32(int m, int n, const void* A, int copies, int reps, double* ms_out, int* sweeps_out);
/* Cholesky of a Hermitian positive definite n x n complex128 matrix (n <= 128): L lower with G = L L^dagger, W = (L^-1)^dagger; *fail = 1 when a
 * pivot fell to tau * max diagonal or below */
int tnqs_dbg_chol(int n, const void* G, void* L_out, void* W_out, int* fail_out, double tau);
/* out[(s',n),(a,b)] = sum_{(s,k)} in[(s,k),(a,b)] X[(s,k),(s',n)] with in element (s,a,k,b) at s + D*(a + PA*(k + K*b)) */
int tnqs_dbg_fiber_gemm(int dtype, int D, int PA, int K, int PB, int Do, int No, const void* in, const void* X, void* out,
                        double* norm2_out, int use_mfma);
/* out[i + KK*j] = sum_{(a,b)} X[i,(a,b)] conj(Y[j,(a,b)]), i,j = (s,k), KK = D*K; acc64: accumulate in double (out complex128) */
int tnqs_dbg_gram(int dtype, int D, int PA, int K, int PB, const void* X, const void* Y, void* out, int acc64, int use_mfma);
/* c64 only: out[i + K*j] = sum ( X x_r M )[i,.] conj(Y[j,.]) with D = 1 and r = the first row leg (chi_r = 32, d = 2) */
int tnqs_dbg_gram_fused(int PA, int K, int PB, const void* X, const void* Y, const void* M, void* out);
/* the gate path's fused gauge + f64 Gram kernels on one ComplexF32 site tensor [2, chi...] (column-major): out (complex128, KK x KK, KK = 2 chi_b)
   = G[i + KK j] = sum X'[i, fibers] conj(X'[j, fibers]), X' = X x_r M rounded to f32, r = the lowest leg that is not bleg; chi_b = chi_r = 32 or 16 */
int tnqs_dbg_gauge_gram(int z, const int* chi, int bleg, const void* X, const void* M, void* out);
/* c64 only: out[c,jx,mid,jy,hi] = sum in[c,ix,mid,iy,hi] Mx[ix,jx] My[iy,jy]; element at c + C0*(ix + 32*(mid + NMID*(iy + 32*hi))) */
int tnqs_dbg_pair(int C0, int NMID, int NHI, const void* in, const void* Mx, const void* My, void* out);
/* c64 only, site tensor [d][chi_0]..[chi_{z-1}] column-major: out = in x_lx Mx x_ly My (chi_lx = chi_ly = 32; leg 0 allowed) */
int tnqs_dbg_pair_legs(int d, int z, const int* chi, int lx, int ly, const void* in, const void* Mx, const void* My, void* out);
/* c64 only: out[b + 32*b'] = sum (X x_lx M)[.., b on leg ly, ..] conj(Y[.., b' on leg ly, ..]) */
int tnqs_dbg_pair_gram(int d, int z, const int* chi, int lx, int ly, const void* X, const void* Y, const void* M, void* out);
/* c64 only: both Grams of the plane (lx, ly) in one pass: out_y keeps ly (lx absorbed with Mx), out_x keeps lx (ly absorbed with My) */
int tnqs_dbg_pair_gram2(int d, int z, const int* chi, int lx, int ly, const void* X, const void* Y, const void* Mx, const void* My, void* out_y, void* out_x);
/* timing only: average launch duration (ms, HIP events) of a chi = 32 plane kernel over `nsites` device-resident tensors [2][32]^4;
 * which = 0 pair product on legs (lx, ly), 1 both-messages pair-Gram */
int tnqs_dbg_bench_plane(int which, int nsites, int lx, int ly, int reps, double* ms);
/* c64 only, d = 2, chi_b = 32: out[s',b',rest] = sum in[s,b,rest] X[(s + 2 b) + 64 (s' + 2 b')]; *norm2 = |out|^2 */
/* the BP sweep order bp_update uses when no edge_sequence is given, as (src[i] -> dst[i]) vertex indices; *n_out = its length (2 ne) */
int tnqs_dbg_default_sequence(tnqs_handle h, int* src, int* dst, int cap, int* n_out);
/* the same order from the graph alone (nv vertices, ne undirected edges esrc[e] - edst[e]) together with the dependency level bp_update runs every message in
 * (messages of one level are launched together; which value a message reads is decided by the positions).  HOST ONLY: no device is touched, so the scheduling
 * logic of the BP update -- linear forests, edge sets that close cycles on periodic lattices -- is testable without a GPU (tests/test_bp_schedule.py) */
int tnqs_dbg_default_sequence_graph(int nv, int ne, const int32_t* esrc, const int32_t* edst, int* src, int* dst, int* level, int cap, int* n_out);
#ifdef __cplusplus
}
#endif
#endif
