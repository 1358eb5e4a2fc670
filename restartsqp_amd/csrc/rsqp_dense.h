// rsqp_dense.h -- dense f64 building blocks of the HBM-resident engine (dense_la.hip):
// MFMA GEMM, blocked Householder QR (compact WY), explicit Q, triangular inverse and blocked
// Cholesky. They replace the one-constraint-at-a-time GEMV/GER construction of the TQ
// factorisation and of the projected Hessian when a solve starts from a non-empty working set
// (qpOASES setupTQfactorisation / computeProjectedCholesky inside SQProblem::hotstart(H, A, ...),
// reference call sites src/qpOASESInterface.cpp:184,197,204-206).
#pragma once
#include <hip/hip_runtime.h>

// C (m x n, ldc) = alpha * op(A) * op(B) + beta * C, all column-major device pointers.
// transA: op(A) = A' (A is k x m, lda >= k), else A is m x k. transB likewise (B is n x k / k x n).
hipError_t rsqp_dgemm(bool transA, bool transB, int m, int n, int k, double alpha, const double *A, long long lda,
                      const double *B, long long ldb, double beta, double *C, long long ldc, hipStream_t st);

// Workspace of the blocked factorisations for problems with up to `mmax` rows.
struct RsqpDenseWork {
    double *V = nullptr;     // mmax x OB   explicit reflectors of an outer block (four panels side by side, unit lower trapezoidal)
    double *T2 = nullptr;    // scratch of the aggregation: V'V of an outer block and a product block
    double *Tout = nullptr;  // OB x OB     aggregated triangular factor of every outer block (rsqp_dgeqrf -> rsqp_dorgqr)
    double *T = nullptr;     // NB x NB     triangular factors of every panel, nb_panels * NB*NB
    double *W = nullptr;     // OB x mmax   V' C
    double *W2 = nullptr;    // OB x mmax   T' W / T W
    double *tau = nullptr;   // mmax
    double *norm2 = nullptr; // mmax        squared norms of the original columns
    double *dblk = nullptr;  // NB x NB     diagonal block scratch
    double *ws = nullptr;    // split-K slabs of the tall-skinny products (V'C with a long inner dimension)
    long long ws_cap = 0;
    int *flag = nullptr;     // [0] = number of dependent columns found, [1] = not positive definite, [2] = a panel was too ill-conditioned
                             //       for the Cholesky-QR panel factorisation: repeat rsqp_dgeqrf with panel_cholqr = false
    double *hr = nullptr;    // 3 NB x NB   R1^-1, R1, (U R2)^-1 of the panel being factorised
    bool panel_cholqr = true;   // panels by Cholesky-QR + Householder reconstruction (a few launches) instead of one launch per column
    long long mmax = 0;
};
hipError_t rsqp_dense_work_alloc(RsqpDenseWork *w, long long mmax);
void rsqp_dense_work_free(RsqpDenseWork *w);

// Householder QR of B (m x n, m >= n, ldb): on exit R in the upper triangle, the reflectors below
// it, tau[n], T factors per panel in w->T. Column j counts as linearly dependent on the previous
// ones when its remaining norm is <= eps_li * its original norm (w->flag[0] is incremented; the
// factorisation is then unusable and the caller falls back to the sequential construction).
hipError_t rsqp_dgeqrf(int m, int n, double *B, long long ldb, double eps_li, RsqpDenseWork *w, hipStream_t st);
// Q (m x m, ldq) = H_1 H_2 ... H_n from the output of rsqp_dgeqrf
hipError_t rsqp_dorgqr(int m, int n, const double *B, long long ldb, double *Q, long long ldq, RsqpDenseWork *w,
                       hipStream_t st);
// C (n x n, symmetric result) = alpha op(A) op(B) + beta C, upper triangle only (tiles strictly below the diagonal are skipped and
// undefined); rsqp_mirror_upper copies the upper triangle into the lower one
hipError_t rsqp_dgemm_upper(bool transA, bool transB, int n, int k, double alpha, const double *A, long long lda,
                            const double *B, long long ldb, double beta, double *C, long long ldc, hipStream_t st);
hipError_t rsqp_mirror_upper(int n, double *M, long long ld, hipStream_t st);
// C (n x n, upper tiles only) = alpha X X' for an UPPER TRIANGULAR X whose strict lower triangle holds zeros (U^-1 U^-T behind
// rsqp_dtrtri_upper): the zero part of the inner dimension is skipped -- a third of the multiply-adds of rsqp_dgemm_upper
// rsqp_dgemm with a triangular operand (ktri: 2 = op(A) upper triangular, A not transposed; 3 = op(A) = A', A upper triangular;
// exact zeros in the other triangle): the zero part of the inner dimension is skipped
hipError_t rsqp_dgemm_tri(bool transA, bool transB, int m, int n, int k, double alpha, const double *A, long long lda,
                          const double *B, long long ldb, double beta, double *C, long long ldc, int ktri, hipStream_t st);
hipError_t rsqp_dtrmmt_upper(int n, double alpha, const double *X, long long ldx, double *C, long long ldc, hipStream_t st);
// X (n x n, ldx) = R^-1 for the upper triangular R (n x n, ldr); X is upper triangular, its strict
// lower part is zeroed
hipError_t rsqp_dtrtri_upper(int n, const double *R, long long ldr, double *X, long long ldx, RsqpDenseWork *w,
                             hipStream_t st);
// Cholesky G = U'U in place (upper triangle of the symmetric n x n G, ldg; the lower triangle is
// not referenced on entry and zeroed on exit). A pivot d with d <= pd_rel*(|g_jj| + sum) + pd_abs
// sets w->flag[1] (not positive definite).
hipError_t rsqp_dpotrf_upper(int n, double *G, long long ldg, double pd_rel, double pd_abs, RsqpDenseWork *w,
                             hipStream_t st);
