// rsqp_sparse.h -- launchers of the sparse / certificate kernels (sparse.hip)
#pragma once
#include "rsqp_internal.h"

#include "rsqp_kkt.h"

// arguments of the fused certificate kernel. Either (nV, nC, offV, offC) arrays for a
// batch, or nV1 / nC1 with null arrays for one problem.
struct RsqpKktArgs {
    const int *nV, *nC;
    const long long *offV, *offC;
    int nV1, nC1;
    const double *x, *y, *g, *lb, *ub, *lbA, *ubA, *Ax, *ATy, *Hx;
    const int *ws_b, *ws_c;
    int *W_b, *W_c;
    double *out;  // 6 doubles per problem: primal, dual, compl, stat, KKT_error, invalid
    int *done_flag;   // single-QP certificate: host-mapped word that receives done_val behind the results (or nullptr)
    int done_val;
};

// row/column blocks for csx_stream_spmv: consecutive majors with at most `chunk`
// entries per block (a longer single major gets a block of its own)
int rsqp_spmv_chunk(void);

// epilogue of the single-matrix product out = A'dy_C on the diagonal range-space path of the HBM-resident engine (qp_large.hip: what
// was the kernel k_dual_dx): dx on the free variables from stationarity, D dx + dg = A'dy_C; H dx of every variable
struct RsqpSpmvDualDx {
    const int *Sb = nullptr; const double *hinv = nullptr, *Hval = nullptr, *gN = nullptr, *g = nullptr; double hreg = 0.0;
    double *dx = nullptr, *Hdx = nullptr;
};
hipError_t rsqp_launch_spmv_dualdx(const int4 *blkinfo, int nblk, const int *ptr, const int *idx, const double *val, const double *in,
                                   double *out, const RsqpSpmvDualDx &epi, hipStream_t stream);
hipError_t rsqp_launch_spmv(const int4 *blkinfo, int nblk, const int *ptr, const int *idx, const double *val,
                            const double *in, double *out, int nbatch, long long ptr_stride,
                            long long nnz_stride, long long in_stride, long long out_stride,
                            hipStream_t stream);
// batched product with the input vector staged in LDS (variant selects lanes per major / unroll)
hipError_t rsqp_launch_spmv_ldsvec(int variant, int nminor, int nslices, const int *slice, const int *ptr,
                                   const int *idx, const unsigned short *idx16, const double *val, const double *in,
                                   double *out, int nbatch, long long ptr_stride, long long nnz_stride, long long in_stride,
                                   long long out_stride, hipStream_t stream);
// entry-parallel variant (40): chunks of whole majors with <= 512 entries, one start bit per entry (16 words per chunk
// and member), list of the non-empty majors, list of the empty ones
hipError_t rsqp_launch_spmv_segscan(int nminor, const int4 *chunks, int nchunks, const int *nzlist, const int *empties,
                                    int nempty, const unsigned *sbits, const unsigned short *idx16, const double *val,
                                    const double *in, double *out, int nbatch, long long nnz_stride, long long in_stride,
                                    long long out_stride, hipStream_t stream);
hipError_t rsqp_launch_scatter(int n, const int *order, const int *tmap, const double *tv, double *val,
                               hipStream_t stream);
hipError_t rsqp_launch_gather(int n, const int *perm, const double *src, double *dst, hipStream_t stream);
// the same for a matrix that stores every entry (CSC = column-major, CSR = row-major): a tiled transpose
hipError_t rsqp_launch_gather_dense(int nrow, int ncol, const double *src, double *dst, hipStream_t stream);
hipError_t rsqp_launch_scatter_csc_csr(int n, const int *order, const int *rorder, const double *tv, double *val, double *rval,
                                       hipStream_t stream);
hipError_t rsqp_launch_densify(int nrow, int ncol, const int *jc, const int *ir, const double *val, double *dense,
                               hipStream_t stream);
hipError_t rsqp_launch_kkt(const RsqpKktArgs &a, int nq, hipStream_t stream);
hipError_t rsqp_launch_small_certificate(const QPPools &p, const RsqpKktArgs &a, int nq, double *Ax, double *ATy,
                                         double *Hx, hipStream_t stream);
