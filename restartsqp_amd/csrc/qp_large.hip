// qp_large.hip -- HBM-resident online active-set QP engine for problems that do not fit the
// LDS-resident kernel (qp_small.hip): dense n=2048 x m=4096, sparse n=10k x m=20k.
//
// Same algorithm and decisions as qp_small.hip / the CPU restatement (homotopy, ratio tests
// with lowest-candidate-id tie break, exchange on linear dependence, bound flipping), but the
// null-space machinery is re-derived for a chip whose strength is wide bandwidth-bound
// kernels, not long dependent chains:
//
//   Z  (nV x nZ)   orthonormal basis of the null space of the active rows on the free variables
//   Y  (nV x nAC)  orthonormal basis of its complement
//   Minv (nAC x nAC) = (A_AC,FR * Y)^-1          (row i <-> column i of Y, column j <-> AC[j])
//   Wz   (nZ x nZ)   = (Z' H Z)^-1
//
// Every working-set change is a Householder reflection (rank-1 update) of Z or Y plus
// Sherman-Morrison / bordering updates of the two explicit inverses; every solve of the step
// direction is a GEMV. No triangular factor, no Givens chain: each operation is a handful of
// GEMV / GER-shaped kernels that stream Z, Y, Minv, Wz once at HBM speed. Deleting a column
// never moves a matrix: the reflection is aimed at the LAST column, which is then dropped;
// deleting constraint k of Minv moves one column. The step directions are independent of the
// choice of basis, so iterates and decisions equal those of the Givens formulation up to
// rounding. Control flow runs on the host: it needs a few scalars per working-set change
// (ratio-test winner, independence / definiteness tests); the kernels publish them into a
// host-mapped block behind a sequence number the host spins on (publish() / Impl::wait_ctl),
// and whatever does not depend on the verdict is launched before the host looks.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <vector>

#include "rsqp_large.h"
#include "rsqp_dense.h"

namespace {

constexpr int NT = 256;

// ---------------------------------------------------------------------------------
// dense building blocks (column-major, leading dimension ld)
// ---------------------------------------------------------------------------------
__device__ inline double block_sum(double v, double *sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// sum over i = threadIdx.x, threadIdx.x + NT, ... < n of f(i) for ONE workgroup, four independent accumulators: a single
// accumulator makes every iteration wait for the previous one's load (40 iterations x ~0.5 us at n = 10 000: the 17 - 24 us of
// the one-workgroup kernels in profiles/r03_h_kernel_stats_large_sparse.csv); every kernel that forms the same sum uses this
// routine, so the bits agree between them (house_tail)
template <class F> __device__ __forceinline__ double lane_sum4(int n, F f) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int i = threadIdx.x;
    for (; i + 3 * NT < n; i += 4 * NT) { s0 += f(i); s1 += f(i + NT); s2 += f(i + 2 * NT); s3 += f(i + 3 * NT); }
    for (; i < n; i += NT) s0 += f(i);
    return (s0 + s1) + (s2 + s3);
}

// out[c] = sum_r M[c*ld + r] * x[r]. A workgroup owns GT_COLS consecutive columns: every lane
// keeps GT_COLS independent 16-byte loads in flight per step and x is read once for all of them.
// (ld and the column bases are even for every matrix of this engine, so the double2 loads are
// 16-byte aligned; an odd tail row is handled separately.)
// The decision block `ctl` is host-mapped memory. A kernel that publishes into it ends with publish(): a system-scope
// fence, then the sequence number of this publication in ctl[63]. The host spins on that number (Impl::wait_ctl)
// instead of sleeping in hipStreamSynchronize: a blocking wait costs ~11 us per round trip on this platform (measured:
// one more of them per working-set change = +5 % on the dense 2048 x 4096 solve), and a change has two or three.
__device__ __forceinline__ void publish(double *ctl, double seqv) {
    __threadfence_system();
    reinterpret_cast<volatile double *>(ctl)[63] = seqv;
}
constexpr int GT_COLS = 4;
// one transposed product with optional fused element-wise work:
//   x-prologue  x[r] := x[r] + (xa[r] - xb[r])       (the residual H dx + (gN - g) of the multiplier step)
//   epilogue    out[c] := acc + addv[c]               (tmpg + H xY), written to out[omap[c]] when omap is given
//               (scatter of the active-constraint multipliers to their constraint indices)
struct GtTask {
    const double *M; long long ld; int nrows, ncols;
    const double *x, *xa, *xb, *addv; const int *omap; double *out;
};
__device__ inline void gemv_t_body(const GtTask &t, int blk, double *sh) {
    const double *__restrict__ M = t.M;
    const double *__restrict__ x = t.x;
    const long long ld = t.ld;
    const int nrows = t.nrows, ncols = t.ncols;
    const int c0 = blk * GT_COLS;
    double s[GT_COLS];
#pragma unroll
    for (int k = 0; k < GT_COLS; k++) s[k] = 0.0;
    const bool vec = ((ld & 1) == 0) && ((reinterpret_cast<unsigned long long>(M) & 15) == 0) &&
                     ((reinterpret_cast<unsigned long long>(x) & 15) == 0) &&
                     (!t.xa || (((reinterpret_cast<unsigned long long>(t.xa) | reinterpret_cast<unsigned long long>(t.xb)) & 15) == 0));
    if (vec) {
        const int n2 = nrows >> 1;
#pragma unroll 2
        for (int r = threadIdx.x; r < n2; r += NT) {
            double2 xv = reinterpret_cast<const double2 *>(x)[r];
            if (t.xa) {
                const double2 a = reinterpret_cast<const double2 *>(t.xa)[r], b = reinterpret_cast<const double2 *>(t.xb)[r];
                xv.x += a.x - b.x; xv.y += a.y - b.y;
            }
#pragma unroll
            for (int k = 0; k < GT_COLS; k++) {
                const int c = c0 + k < ncols ? c0 + k : ncols - 1;
                const double2 m = reinterpret_cast<const double2 *>(M + c * ld)[r];
                s[k] += m.x * xv.x + m.y * xv.y;
            }
        }
        if ((nrows & 1) && threadIdx.x == 0) {
            const int r = nrows - 1;
            const double xv = x[r] + (t.xa ? t.xa[r] - t.xb[r] : 0.0);
#pragma unroll
            for (int k = 0; k < GT_COLS; k++) {
                const int c = c0 + k < ncols ? c0 + k : ncols - 1;
                s[k] += M[c * ld + r] * xv;
            }
        }
    } else {
        for (int r = threadIdx.x; r < nrows; r += NT) {
            const double xv = x[r] + (t.xa ? t.xa[r] - t.xb[r] : 0.0);
#pragma unroll
            for (int k = 0; k < GT_COLS; k++) {
                const int c = c0 + k < ncols ? c0 + k : ncols - 1;
                s[k] += M[c * ld + r] * xv;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < GT_COLS; k++) {
        const double v = block_sum(s[k], sh);
        if (threadIdx.x == 0 && c0 + k < ncols) {
            const int c = c0 + k;
            t.out[t.omap ? t.omap[c] : c] = t.addv ? v + t.addv[c] : v;
        }
    }
}
__global__ void __launch_bounds__(NT) k_gemv_t(GtTask t) {
    __shared__ double sh[4];
    gemv_t_body(t, blockIdx.x, sh);
}
// two independent transposed products in one launch (blocks [0, nb0) work on t0, the rest on t1)
__global__ void __launch_bounds__(NT) k_gemv_t2(GtTask t0, int nb0, GtTask t1) {
    __shared__ double sh[4];
    if ((int)blockIdx.x < nb0) gemv_t_body(t0, blockIdx.x, sh);
    else gemv_t_body(t1, blockIdx.x - nb0, sh);
}

// out = beta*base + alpha * M w, column-major M. A workgroup owns 64 consecutive rows (one
// 512-byte run per column) and a chunk of columns, its four waves stride the chunk; the four
// wave sums meet in LDS in fixed order. With one chunk the result is final; otherwise the
// partials [chunk][row] are summed in chunk order by k_gemv_n_reduce (deterministic).
template <bool VEC>
__global__ void __launch_bounds__(NT) k_gemv_n_part(const double *__restrict__ M, long long ld, int nrows, int ncols,
                                                    int chunk, const double *__restrict__ w, double *__restrict__ part,
                                                    double alpha, double beta, const double *__restrict__ base,
                                                    double *__restrict__ out) {
    // VEC: a workgroup owns 128 rows, each lane two consecutive ones (16-byte loads, 1 KiB per
    // wave-instruction); otherwise 64 rows, one per lane (odd leading dimension / unaligned)
    constexpr int RPL = VEC ? 2 : 1;
    __shared__ double sh[4][64 * RPL];
    const int lane = threadIdx.x & 63, cl = threadIdx.x >> 6;
    const int r = (blockIdx.x * 64 + lane) * RPL;
    const int c0 = blockIdx.y * chunk, c1 = min(c0 + chunk, ncols);
    double a0 = 0.0, a1 = 0.0;
    if (VEC) {
        if (r + 1 < nrows) {
            double2 s0 = {0, 0}, s1 = {0, 0}, s2 = {0, 0}, s3 = {0, 0};
            int c = c0 + cl;
            for (; c + 12 < c1; c += 16) {
                const double2 m0 = *reinterpret_cast<const double2 *>(M + c * ld + r);
                const double2 m1 = *reinterpret_cast<const double2 *>(M + (c + 4) * ld + r);
                const double2 m2 = *reinterpret_cast<const double2 *>(M + (c + 8) * ld + r);
                const double2 m3 = *reinterpret_cast<const double2 *>(M + (c + 12) * ld + r);
                const double w0 = w[c], w1 = w[c + 4], w2 = w[c + 8], w3 = w[c + 12];
                s0.x += m0.x * w0; s0.y += m0.y * w0; s1.x += m1.x * w1; s1.y += m1.y * w1;
                s2.x += m2.x * w2; s2.y += m2.y * w2; s3.x += m3.x * w3; s3.y += m3.y * w3;
            }
            for (; c < c1; c += 4) {
                const double2 m0 = *reinterpret_cast<const double2 *>(M + c * ld + r);
                s0.x += m0.x * w[c]; s0.y += m0.y * w[c];
            }
            a0 = (s0.x + s1.x) + (s2.x + s3.x);
            a1 = (s0.y + s1.y) + (s2.y + s3.y);
        } else if (r < nrows) {
            for (int c = c0 + cl; c < c1; c += 4) a0 += M[c * ld + r] * w[c];
        }
        sh[cl][2 * lane] = a0;
        sh[cl][2 * lane + (RPL - 1)] = RPL == 2 ? a1 : a0;
    } else {
        if (r < nrows) {
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int c = c0 + cl;
            for (; c + 12 < c1; c += 16) {
                s0 += M[c * ld + r] * w[c];
                s1 += M[(c + 4) * ld + r] * w[c + 4];
                s2 += M[(c + 8) * ld + r] * w[c + 8];
                s3 += M[(c + 12) * ld + r] * w[c + 12];
            }
            for (; c < c1; c += 4) s0 += M[c * ld + r] * w[c];
            a0 = (s0 + s1) + (s2 + s3);
        }
        sh[cl][lane] = a0;
    }
    __syncthreads();
    if (cl == 0) {
#pragma unroll
        for (int k = 0; k < RPL; k++) {
            const int rr = r + k, li = RPL * lane + k;
            if (rr < nrows) {
                const double t = (sh[0][li] + sh[1][li]) + (sh[2][li] + sh[3][li]);
                if (gridDim.y == 1) out[rr] = (base ? beta * base[rr] : 0.0) + alpha * t;
                else part[(long long)blockIdx.y * nrows + rr] = t;
            }
        }
    }
}
// out[r] = beta * base[r] + alpha * sum_chunks part. 64 rows per workgroup, the four waves take
// every fourth partial; combined in LDS in fixed order.
__global__ void __launch_bounds__(NT) k_gemv_n_reduce(const double *__restrict__ part, int nrows, int nchunks,
                                                      double alpha, double beta, const double *__restrict__ base,
                                                      double *__restrict__ out) {
    __shared__ double sh[4][64];
    const int lane = threadIdx.x & 63, pg = threadIdx.x >> 6;
    const int r = blockIdx.x * 64 + lane;
    double s = 0.0;
    if (r < nrows)
        for (int k = pg; k < nchunks; k += 4) s += part[(long long)k * nrows + r];
    sh[pg][lane] = s;
    __syncthreads();
    if (pg == 0 && r < nrows)
        out[r] = (base ? beta * base[r] : 0.0) + alpha * ((sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]));
}

// Single-launch form of out = beta*base + alpha * M w for matrices that are launch-bound rather than HBM-bound
// (nrows <= 8192: the two-kernel form above costs a second ~4 us launch for the reduction). A workgroup owns 16
// consecutive rows (8 lanes x 16-byte loads = one 128-byte line per column) and ALL columns, dealt to 32 column
// groups; the 32 partial sums of a row meet in LDS and are added in group order (deterministic). 128 workgroups
// x 256 threads x 8 loads in flight keep ~4 MB on the wire for a 2048 x 2048 matrix.
// Optional epilogue (the step direction's "dx on the free variables", k_merge_free): mdst[r] = out[r] where mSb[r] == 0.
// SKIP0: a column whose weight is exactly zero is not read -- for products whose vector lives on a subset (dx on the
// fixed variables): the pass costs the live columns only (0 * m is dropped, not added)
template <bool SKIP0> __device__ __forceinline__ double2 ld_live(const double *p, double wc) {
    if (SKIP0 && wc == 0.0) return make_double2(0.0, 0.0);
    return *reinterpret_cast<const double2 *>(p);
}
template <bool VEC, int NTH, bool SKIP0>
__device__ __forceinline__ void gemv_n1_body(const int bx, const double *__restrict__ M, long long ld, int nrows, int ncols,
                                             const double *__restrict__ w, double alpha, double beta,
                                             const double *__restrict__ base, double *__restrict__ out,
                                             const int *__restrict__ mSb, double *__restrict__ mdst) {
    constexpr int RPL = VEC ? 2 : 1;
    constexpr int NG = NTH / 8;          // column groups
    __shared__ double sh[NG][8 * RPL + 1];
    const int rl = threadIdx.x & 7, cg = threadIdx.x >> 3;
    const int r = (bx * 8 + rl) * RPL;
    double a0 = 0.0, a1 = 0.0;
    if (VEC) {
        if (r + 1 < nrows) {
            double2 s0 = {0, 0}, s1 = {0, 0}, s2 = {0, 0}, s3 = {0, 0};
            int c = cg;
            for (; c + 3 * NG < ncols; c += 4 * NG) {
                const double w0 = w[c], w1 = w[c + NG], w2 = w[c + 2 * NG], w3 = w[c + 3 * NG];
                const double2 m0 = ld_live<SKIP0>(M + c * ld + r, w0);
                const double2 m1 = ld_live<SKIP0>(M + (c + NG) * ld + r, w1);
                const double2 m2 = ld_live<SKIP0>(M + (c + 2 * NG) * ld + r, w2);
                const double2 m3 = ld_live<SKIP0>(M + (c + 3 * NG) * ld + r, w3);
                s0.x += m0.x * w0; s0.y += m0.y * w0; s1.x += m1.x * w1; s1.y += m1.y * w1;
                s2.x += m2.x * w2; s2.y += m2.y * w2; s3.x += m3.x * w3; s3.y += m3.y * w3;
            }
            for (; c < ncols; c += NG) {
                const double w0 = w[c];
                const double2 m0 = ld_live<SKIP0>(M + c * ld + r, w0);
                s0.x += m0.x * w0; s0.y += m0.y * w0;
            }
            a0 = (s0.x + s1.x) + (s2.x + s3.x);
            a1 = (s0.y + s1.y) + (s2.y + s3.y);
        } else if (r < nrows) {
            for (int c = cg; c < ncols; c += NG) a0 += M[c * ld + r] * w[c];
        }
        sh[cg][2 * rl] = a0; sh[cg][2 * rl + 1] = a1;
    } else {
        if (r < nrows) {
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int c = cg;
            for (; c + 3 * NG < ncols; c += 4 * NG) {
                s0 += M[c * ld + r] * w[c];
                s1 += M[(c + NG) * ld + r] * w[c + NG];
                s2 += M[(c + 2 * NG) * ld + r] * w[c + 2 * NG];
                s3 += M[(c + 3 * NG) * ld + r] * w[c + 3 * NG];
            }
            for (; c < ncols; c += NG) s0 += M[c * ld + r] * w[c];
            a0 = (s0 + s1) + (s2 + s3);
        }
        sh[cg][rl] = a0;
    }
    __syncthreads();
    if ((int)threadIdx.x < 8 * RPL) {
        const int rr = bx * 8 * RPL + threadIdx.x;
        if (rr < nrows) {
            double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll
            for (int g = 0; g < NG; g += 4) {
                t0 += sh[g][threadIdx.x]; t1 += sh[g + 1][threadIdx.x]; t2 += sh[g + 2][threadIdx.x]; t3 += sh[g + 3][threadIdx.x];
            }
            const double v = (base ? beta * base[rr] : 0.0) + alpha * ((t0 + t1) + (t2 + t3));
            out[rr] = v;
            if (mSb && mSb[rr] == 0) mdst[rr] = v;
        }
    }
}
template <bool VEC, int NTH, bool SKIP0 = false>
__global__ void __launch_bounds__(NTH) k_gemv_n1(const double *__restrict__ M, long long ld, int nrows, int ncols,
                                                const double *__restrict__ w, double alpha, double beta,
                                                const double *__restrict__ base, double *__restrict__ out,
                                                const int *__restrict__ mSb, double *__restrict__ mdst) {
    gemv_n1_body<VEC, NTH, SKIP0>(blockIdx.x, M, ld, nrows, ncols, w, alpha, beta, base, out, mSb, mdst);
}
// two independent products out = M w of that form in ONE launch (the first nb0 workgroups take the first): t = Z v and s = Wz v of
// an incoming constraint's reflection, A dx_FX and H dx_FX of the step direction -- launch-bound sizes, where a launch costs more
// than the product (same body: the same bits as two launches)
template <int NTH, bool SKIP0>
__global__ void __launch_bounds__(NTH) k_gemv_n1_pair(const double *__restrict__ M0, long long ld0, int nrows0, int ncols0,
                                                     const double *__restrict__ w0, double *__restrict__ out0, int nb0,
                                                     const double *__restrict__ M1, long long ld1, int nrows1, int ncols1,
                                                     const double *__restrict__ w1, double *__restrict__ out1) {
    if ((int)blockIdx.x < nb0) gemv_n1_body<true, NTH, SKIP0>(blockIdx.x, M0, ld0, nrows0, ncols0, w0, 1.0, 0.0, nullptr, out0, nullptr, nullptr);
    else gemv_n1_body<true, NTH, SKIP0>(blockIdx.x - nb0, M1, ld1, nrows1, ncols1, w1, 1.0, 0.0, nullptr, out1, nullptr, nullptr);
}

// The rank-1 update M += coef t v' and out = beta * base + alpha * M w of the UPDATED matrix in one pass, in the single-launch
// form of k_gemv_n1 (a workgroup owns 16 consecutive rows and all columns, 16-byte accesses; leading dimension even, M 16-byte
// aligned): the deferred reflection of Z riding on the step direction's LAST product when its first one is carried (k_carry_add).
template <int NTH>
__global__ void __launch_bounds__(NTH) k_ger_gemv_n1(double *__restrict__ M, long long ld, int nrows, int ncols,
                                                     const double *__restrict__ ut, const double *__restrict__ uv,
                                                     const double *__restrict__ scal, int ci, double cs,
                                                     const double *__restrict__ w, double alpha, double beta,
                                                     const double *__restrict__ base, double *__restrict__ out,
                                                     const int *__restrict__ mSb, double *__restrict__ mdst) {
    constexpr int NG = NTH / 8;          // column groups
    __shared__ double sh[NG][17];
    const int rl = threadIdx.x & 7, cg = threadIdx.x >> 3;
    const int r = (blockIdx.x * 8 + rl) * 2;
    const double coef = cs * scal[ci];
    double a0 = 0.0, a1 = 0.0;
    if (r + 1 < nrows) {
        const double2 tv = *reinterpret_cast<const double2 *>(ut + r);
        const double t0 = coef * tv.x, t1 = coef * tv.y;
        double2 s0 = {0, 0}, s1 = {0, 0};
        int c = cg;
        for (; c + NG < ncols; c += 2 * NG) {
            double *p0 = M + c * ld + r, *p1 = M + (c + NG) * ld + r;
            double2 m0 = *reinterpret_cast<double2 *>(p0), m1 = *reinterpret_cast<double2 *>(p1);
            const double v0 = uv[c], v1 = uv[c + NG], w0 = w[c], w1 = w[c + NG];
            m0.x += t0 * v0; m0.y += t1 * v0; m1.x += t0 * v1; m1.y += t1 * v1;
            *reinterpret_cast<double2 *>(p0) = m0; *reinterpret_cast<double2 *>(p1) = m1;
            s0.x += m0.x * w0; s0.y += m0.y * w0; s1.x += m1.x * w1; s1.y += m1.y * w1;
        }
        for (; c < ncols; c += NG) {
            double *p0 = M + c * ld + r;
            double2 m0 = *reinterpret_cast<double2 *>(p0);
            const double v0 = uv[c], w0 = w[c];
            m0.x += t0 * v0; m0.y += t1 * v0;
            *reinterpret_cast<double2 *>(p0) = m0;
            s0.x += m0.x * w0; s0.y += m0.y * w0;
        }
        a0 = s0.x + s1.x; a1 = s0.y + s1.y;
    } else if (r < nrows) {
        const double t0 = coef * ut[r];
        for (int c = cg; c < ncols; c += NG) { double *p0 = M + c * ld + r; const double m = *p0 + t0 * uv[c]; *p0 = m; a0 += m * w[c]; }
    }
    sh[cg][2 * rl] = a0; sh[cg][2 * rl + 1] = a1;
    __syncthreads();
    if ((int)threadIdx.x < 16) {
        const int rr = blockIdx.x * 16 + threadIdx.x;
        if (rr < nrows) {
            double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll
            for (int g = 0; g < NG; g += 4) {
                t0 += sh[g][threadIdx.x]; t1 += sh[g + 1][threadIdx.x]; t2 += sh[g + 2][threadIdx.x]; t3 += sh[g + 3][threadIdx.x];
            }
            const double v = (base ? beta * base[rr] : 0.0) + alpha * ((t0 + t1) + (t2 + t3));
            out[rr] = v;
            if (mSb && mSb[rr] == 0) mdst[rr] = v;
        }
    }
}

// M[c*ld + r] += coef * t[r] * v[c]   (coef = scal[ci] * cs)
__global__ void __launch_bounds__(NT) k_ger(double *__restrict__ M, long long ld, int nrows, int ncols,
                                            const double *__restrict__ t, const double *__restrict__ v,
                                            const double *__restrict__ scal, int ci, double cs) {
    const int r = blockIdx.x * NT + threadIdx.x, c = blockIdx.y;
    if (r >= nrows || c >= ncols) return;
    M[c * ld + r] += cs * scal[ci] * t[r] * v[c];
}

// the same update, a lane owning two consecutive rows (16-byte accesses) of GER_COLS columns: 8 independent
// read-modify-writes per lane in flight instead of one 8-byte one (leading dimension even, M 16-byte aligned)
constexpr int GER_COLS = 4;
__global__ void __launch_bounds__(NT) k_ger_v(double *__restrict__ M, long long ld, int nrows, int ncols,
                                              const double *__restrict__ t, const double *__restrict__ v,
                                              const double *__restrict__ scal, int ci, double cs) {
    const int r = (blockIdx.x * NT + threadIdx.x) * 2, c0 = blockIdx.y * GER_COLS;
    if (r >= nrows) return;
    const double coef = cs * scal[ci];
    if (r + 1 < nrows) {
        const double2 tv = *reinterpret_cast<const double2 *>(t + r);
        const double a0 = coef * tv.x, a1 = coef * tv.y;
        double2 m[GER_COLS];
#pragma unroll
        for (int k = 0; k < GER_COLS; k++)
            if (c0 + k < ncols) m[k] = *reinterpret_cast<const double2 *>(M + (c0 + k) * ld + r);
#pragma unroll
        for (int k = 0; k < GER_COLS; k++)
            if (c0 + k < ncols) {
                const double vc = v[c0 + k];
                m[k].x += a0 * vc; m[k].y += a1 * vc;
                *reinterpret_cast<double2 *>(M + (c0 + k) * ld + r) = m[k];
            }
    } else {
        const double a0 = coef * t[r];
        for (int k = 0; k < GER_COLS && c0 + k < ncols; k++) M[(c0 + k) * ld + r] += a0 * v[c0 + k];
    }
}

// The rank-1 update M += coef t v' and the transposed product out = M'x of the UPDATED matrix in one pass (the
// reflection of Z by an incoming constraint followed by the step direction's Z'(H xY + g~): one read + one write of Z
// instead of read + write + read). Structure, element expression and summation order are those of k_ger_v and
// gemv_t_body's 16-byte path (leading dimension even, M / x / t 16-byte aligned: checked by the launcher).
__global__ void __launch_bounds__(NT) k_ger_gemv_t(double *__restrict__ M, long long ld, int nrows, int ncols,
                                                   const double *__restrict__ ut, const double *__restrict__ uv,
                                                   const double *__restrict__ scal, int ci, double cs,
                                                   const double *__restrict__ x, const double *__restrict__ addv,
                                                   double *__restrict__ out, const double *__restrict__ xa = nullptr,
                                                   const double *__restrict__ xb = nullptr, const int *__restrict__ omap = nullptr) {
    // (xa / xb: the x-prologue x + (xa - xb) of gemv_t_body; omap: its scatter epilogue)
    __shared__ double sh[4];
    const int c0 = blockIdx.x * GT_COLS;
    const double coef = cs * scal[ci];
    double s[GT_COLS], vc[GT_COLS];
#pragma unroll
    for (int k = 0; k < GT_COLS; k++) { s[k] = 0.0; vc[k] = c0 + k < ncols ? uv[c0 + k] : 0.0; }
    const int n2 = nrows >> 1;
#pragma unroll 2
    for (int r = threadIdx.x; r < n2; r += NT) {
        double2 xv = reinterpret_cast<const double2 *>(x)[r];
        if (xa) {
            const double2 a = reinterpret_cast<const double2 *>(xa)[r], b = reinterpret_cast<const double2 *>(xb)[r];
            xv.x += a.x - b.x; xv.y += a.y - b.y;
        }
        const double2 tv = reinterpret_cast<const double2 *>(ut)[r];
        const double a0 = coef * tv.x, a1 = coef * tv.y;
        double2 m[GT_COLS];
#pragma unroll
        for (int k = 0; k < GT_COLS; k++)
            if (c0 + k < ncols) m[k] = reinterpret_cast<const double2 *>(M + (c0 + k) * ld)[r];
#pragma unroll
        for (int k = 0; k < GT_COLS; k++)
            if (c0 + k < ncols) {
                m[k].x += a0 * vc[k]; m[k].y += a1 * vc[k];
                reinterpret_cast<double2 *>(M + (c0 + k) * ld)[r] = m[k];
                s[k] += m[k].x * xv.x + m[k].y * xv.y;
            }
    }
    if ((nrows & 1) && threadIdx.x == 0) {
        const int r = nrows - 1;
        const double a0 = coef * ut[r];
#pragma unroll
        for (int k = 0; k < GT_COLS; k++)
            if (c0 + k < ncols) {
                double *p = M + (c0 + k) * ld + r;
                const double m = *p + a0 * vc[k];
                *p = m;
                s[k] += m * (x[r] + (xa ? xa[r] - xb[r] : 0.0));
            }
    }
#pragma unroll
    for (int k = 0; k < GT_COLS; k++) {
        const double v = block_sum(s[k], sh);
        if (threadIdx.x == 0 && c0 + k < ncols) out[omap ? omap[c0 + k] : c0 + k] = addv ? v + addv[c0 + k] : v;
    }
}
// Everything behind the two products of a carried constraint REMOVAL (t = Y v, s' = v'Minv) in ONE launch over max(nV, n) entries
// (n = nAC before the removal, kk = n - 1): k_keep_removal, the reflection of the last column of Y (it becomes the new column of Z:
// written to both) and of the last column of Minv (which then fills the slot k of the removed constraint), the carried range-space
// part -- wY <- (1 - tau)(P wY)[0..kk), omega = (P wY)[kk], xY <- (1 - tau)(xY - omega z_new) -- and the working-set bookkeeping
// (k_ws_remove_c). Every workgroup forms the dot product v'wY itself (same loop, same tree: the same bits in all of them);
// workgroup 0 writes the new wY into wYout (the OTHER half of the double buffer: the others may still be reading wY).
__global__ void __launch_bounds__(NT) k_remove_tail(int n, int nV, int k, double om, const double *__restrict__ v,
                                                    const double *__restrict__ srow, const double *__restrict__ t,
                                                    double *__restrict__ keep_v, double *__restrict__ keep_s, double *__restrict__ scal,
                                                    int from, int to, int sl, double *__restrict__ ylast, double *__restrict__ znew,
                                                    double *__restrict__ mlast, double *__restrict__ mk, const double *__restrict__ wY,
                                                    double *__restrict__ wYout, double *__restrict__ xY, int *AC, int *posAC, int *Sc,
                                                    int r, double *y, int yidx) {
    __shared__ double sh[4];
    const int kk = n - 1;
    double d = lane_sum4(kk + 1, [&](int i) { return v[i] * wY[i]; });
    d = block_sum(d, sh);
    const double beta = scal[from], c = beta * d, omega = wY[kk] - c * v[kk], coef = -1.0 * beta;
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < kk; i += NT) wYout[i] = om * (wY[i] - c * v[i]);
        if (threadIdx.x == 0) {
            scal[sl] = omega;
            scal[to] = beta;
            if (k != kk) { const int rl = AC[kk]; AC[k] = rl; posAC[rl] = k; }
            posAC[r] = -1; Sc[r] = 0;
            y[yidx] = 0.0;
        }
    }
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i < n) {
        const double vi = v[i];
        keep_v[i] = vi;
        keep_s[i] = i == k ? 0.0 : srow[i];
        const double m = mlast[i] + (coef * vi) * srow[kk];
        mlast[i] = m;
        if (mk) mk[i] = m;
    }
    if (i < nV) {
        const double m = ylast[i] + (coef * t[i]) * v[kk];
        ylast[i] = m;
        znew[i] = m;
        xY[i] = om * (xY[i] - omega * m);
    }
}
// what the deferred reflections of Y and Minv behind a removed constraint need later (remove_constraint_tq): v twice (Y's column
// coefficients = Minv's row coefficients), s = v'Minv with the entry of the column that is refilled zeroed, beta
__global__ void k_keep_removal(int n, const double *__restrict__ v, const double *__restrict__ srow, int kzero, double *__restrict__ keep_v,
                               double *__restrict__ keep_s, double *__restrict__ scal, int from, int to) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { keep_v[i] = v[i]; keep_s[i] = i == kzero ? 0.0 : srow[i]; }
    if (i == 0) scal[to] = scal[from];
}
// what a deferred reflection needs later: the Householder vector and its beta out of the way of the next products
__global__ void k_keep_reflector(const double *__restrict__ v, int n, double *__restrict__ keep, double *__restrict__ scal, int from,
                                 int to) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keep[i] = v[i];
    if (i == 0) scal[to] = scal[from];
}

// ... and, in the same launch, the reflection of the ONE column that is needed at once (the last column of Z, which becomes the new
// column of Y: element expression of k_ger_v) written to both places: k_keep_reflector + k_ger_v on one column + k_copy
__global__ void k_keep_reflect_lastcol(const double *__restrict__ v, int n, double *__restrict__ keep, double *__restrict__ scal, int from,
                                       int to, int nrows, const double *__restrict__ t, double *__restrict__ zlast,
                                       double *__restrict__ ycol) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keep[i] = v[i];
    if (i == 0) scal[to] = scal[from];
    if (i < nrows) {
        const double coef = -1.0 * scal[from];
        const double m = zlast[i] + (coef * t[i]) * v[n - 1];
        zlast[i] = m;
        ycol[i] = m;
    }
}

// scal[slot] = sum_i a[i]*b[i]
__global__ void __launch_bounds__(NT) k_dot(const double *__restrict__ a, const double *__restrict__ b, int n,
                                            double *__restrict__ scal, int slot) {
    __shared__ double sh[4];
    double s = lane_sum4(n, [&](int i) { return a[i] * b[i]; });
    s = block_sum(s, sh);
    if (threadIdx.x == 0) scal[slot] = s;
}

__global__ void k_copy(const double *__restrict__ src, double *__restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
__global__ void k_fill(double *__restrict__ dst, int n, double v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = v;
}

// Householder vector for mapping w (length n) onto its LAST component:
//   alpha = |w|, v = w, v[n-1] += sgn(w[n-1]) alpha, beta = 1 / (alpha (alpha + |w[n-1]|)),
//   P = I - beta v v' maps w to  -sgn(w[n-1]) alpha e_last.
// scal[s0] = alpha, scal[s0+1] = beta, scal[s0+2] = -sgn(w_last) (sign of the image), scal[s0+3] = |w|^2
__global__ void __launch_bounds__(NT) k_house(const double *__restrict__ w, int n, double *__restrict__ v,
                                              double *__restrict__ scal, int s0) {
    __shared__ double sh[4];
    double s = lane_sum4(n, [&](int i) { return w[i] * w[i]; });
    s = block_sum(s, sh);
    const double alpha = sqrt(s), wl = w[n - 1], sg = wl >= 0.0 ? 1.0 : -1.0;
    for (int i = threadIdx.x; i < n; i += NT) v[i] = w[i] + (i == n - 1 ? sg * alpha : 0.0);
    if (threadIdx.x == 0) {
        scal[s0] = alpha;
        scal[s0 + 1] = alpha > 0.0 ? 1.0 / (alpha * (alpha + fabs(wl))) : 0.0;
        scal[s0 + 2] = -sg;
        scal[s0 + 3] = s;
    }
}

// the same Householder data for w = wz1 whose squared norm s a publishing kernel has just formed (same loop, same
// tree: the same bits as k_house would compute) -- what the reflection of Z needs if the independence test passes;
// written after the publication, i.e. while the host is still on its way (hv == nullptr or n == 0: nothing)
__device__ __forceinline__ void house_tail(const double *__restrict__ w, int n, double s, double *__restrict__ hv,
                                           double *__restrict__ scal) {
    if (!hv || n <= 0) return;
    __syncthreads();            // w may have been written by this workgroup (k_bound_products)
    const double alpha = sqrt(s), wl = w[n - 1], sg = wl >= 0.0 ? 1.0 : -1.0;
    for (int i = threadIdx.x; i < n; i += NT) hv[i] = w[i] + (i == n - 1 ? sg * alpha : 0.0);
    if (threadIdx.x == 0) {
        scal[0] = alpha;
        scal[1] = alpha > 0.0 ? 1.0 / (alpha * (alpha + fabs(wl))) : 0.0;
        scal[2] = -sg;
        scal[3] = s;
    }
}

// ---- Wz updates ------------------------------------------------------------------
// two-sided reflection + elimination of the last row/column (null space loses its last
// column after the reflection P = I - beta v v'):
//   Wt = P Wz P ;  Wz' = Wt_11 - Wt_12 Wt_12' / Wt_22
// with s = Wz v, theta = v's:  Wt[a][b] = Wz[a][b] - beta s_a v_b - beta v_a s_b + beta^2 theta v_a v_b
// col[a] = Wt[a][last] is prepared by k_wz_lastcol; then one pass over the leading block.
// (theta = v's is formed here, by every workgroup in the same order; workgroup 0 publishes it in scal[st])
__global__ void __launch_bounds__(NT) k_wz_lastcol(const double *__restrict__ Wz, long long ld, int nZ, const double *__restrict__ s,
                             const double *__restrict__ v, double *__restrict__ scal, int sb, int st,
                             double *__restrict__ col) {
    __shared__ double sh[4];
    double th = lane_sum4(nZ, [&](int i) { return v[i] * s[i]; });
    th = block_sum(th, sh);
    if (blockIdx.x == 0 && threadIdx.x == 0) scal[st] = th;
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= nZ) return;
    const int l = nZ - 1;
    const double beta = scal[sb], theta = th;
    col[a] = Wz[(long long)l * ld + a] - beta * s[a] * v[l] - beta * v[a] * s[l] + beta * beta * theta * v[a] * v[l];
}
__global__ void __launch_bounds__(NT) k_wz_shrink(double *__restrict__ Wz, long long ld, int nZ,
                                                  const double *__restrict__ s, const double *__restrict__ v,
                                                  const double *__restrict__ col, const double *__restrict__ scal, int sb,
                                                  int st) {
    const int a = blockIdx.x * NT + threadIdx.x, b = blockIdx.y;
    const int l = nZ - 1;
    if (a >= l || b >= l) return;
    const double beta = scal[sb], theta = scal[st], w22 = col[l];
    Wz[(long long)b * ld + a] += -beta * s[a] * v[b] - beta * v[a] * s[b] + beta * beta * theta * v[a] * v[b] -
                                 col[a] * col[b] / w22;
}
// the two Wz updates with 16-byte accesses: a lane owns two consecutive rows of WZ_COLS columns (the element
// expressions are those of the scalar kernels, term by term). Leading dimension even, Wz 16-byte aligned.
constexpr int WZ_COLS = 4;
__global__ void __launch_bounds__(NT) k_wz_shrink_v(double *__restrict__ Wz, long long ld, int nZ,
                                                    const double *__restrict__ s, const double *__restrict__ v,
                                                    const double *__restrict__ col, const double *__restrict__ scal, int sb,
                                                    int st) {
    const int a = (blockIdx.x * NT + threadIdx.x) * 2, b0 = blockIdx.y * WZ_COLS;
    const int l = nZ - 1;
    if (a >= l) return;
    const double beta = scal[sb], theta = scal[st], w22 = col[l];
    const bool two = a + 1 < l;
    const double sa0 = s[a], va0 = v[a], ca0 = col[a];
    const double sa1 = two ? s[a + 1] : 0.0, va1 = two ? v[a + 1] : 0.0, ca1 = two ? col[a + 1] : 0.0;
#pragma unroll
    for (int k = 0; k < WZ_COLS; k++) {
        const int b = b0 + k;
        if (b >= l) break;
        const double vb = v[b], sbv = s[b], cb = col[b];
        double *p = Wz + (long long)b * ld + a;
        if (two) {
            double2 m = *reinterpret_cast<double2 *>(p);
            m.x += -beta * sa0 * vb - beta * va0 * sbv + beta * beta * theta * va0 * vb - ca0 * cb / w22;
            m.y += -beta * sa1 * vb - beta * va1 * sbv + beta * beta * theta * va1 * vb - ca1 * cb / w22;
            *reinterpret_cast<double2 *>(p) = m;
        } else {
            p[0] += -beta * sa0 * vb - beta * va0 * sbv + beta * beta * theta * va0 * vb - ca0 * cb / w22;
        }
    }
}
// k_wz_shrink_v and the product out = alpha * Wz' w of the SHRUNK matrix in one pass (the elimination of the null-space
// column an incoming constraint takes, followed by the step direction's Wz (Z'g~)): one read + one write of Wz instead
// of read + write + read. A workgroup owns 128 rows (a lane two consecutive ones) and a chunk of columns, its four waves
// stride the chunk as in k_gemv_n_part; partial sums [chunk][row] go through k_gemv_n_reduce (gridDim.y > 1).
__global__ void __launch_bounds__(NT) k_wz_shrink_gemv(double *__restrict__ Wz, long long ld, int nZ, const double *__restrict__ s,
                                                       const double *__restrict__ v, const double *__restrict__ col,
                                                       const double *__restrict__ scal, int sb, int st, int chunk,
                                                       const double *__restrict__ w, double alpha, double *__restrict__ part,
                                                       double *__restrict__ out) {
    __shared__ double sh[4][128];
    const int lane = threadIdx.x & 63, cl = threadIdx.x >> 6;
    const int l = nZ - 1;
    const int a = (blockIdx.x * 64 + lane) * 2;
    const int c0 = blockIdx.y * chunk, c1 = min(c0 + chunk, l);
    const double beta = scal[sb], theta = scal[st], w22 = col[l];
    double acc0 = 0.0, acc1 = 0.0;
    if (a < l) {
        const bool two = a + 1 < l;
        const double sa0 = s[a], va0 = v[a], ca0 = col[a];
        const double sa1 = two ? s[a + 1] : 0.0, va1 = two ? v[a + 1] : 0.0, ca1 = two ? col[a + 1] : 0.0;
        if (two) {
            double2 t[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
            auto upd = [&](double2 &m, double vb, double sbv, double cb) {
                m.x += -beta * sa0 * vb - beta * va0 * sbv + beta * beta * theta * va0 * vb - ca0 * cb / w22;
                m.y += -beta * sa1 * vb - beta * va1 * sbv + beta * beta * theta * va1 * vb - ca1 * cb / w22;
            };
            int b = c0 + cl;
            for (; b + 12 < c1; b += 16) {
                double2 m[4];
                double vb[4], sbv[4], cb[4], wb[4];
#pragma unroll
                for (int k = 0; k < 4; k++) m[k] = *reinterpret_cast<double2 *>(Wz + (long long)(b + 4 * k) * ld + a);
#pragma unroll
                for (int k = 0; k < 4; k++) { vb[k] = v[b + 4 * k]; sbv[k] = s[b + 4 * k]; cb[k] = col[b + 4 * k]; wb[k] = w[b + 4 * k]; }
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    upd(m[k], vb[k], sbv[k], cb[k]);
                    *reinterpret_cast<double2 *>(Wz + (long long)(b + 4 * k) * ld + a) = m[k];
                    t[k].x += m[k].x * wb[k]; t[k].y += m[k].y * wb[k];
                }
            }
            for (; b < c1; b += 4) {
                double *p0 = Wz + (long long)b * ld + a;
                double2 m0 = *reinterpret_cast<double2 *>(p0);
                upd(m0, v[b], s[b], col[b]);
                *reinterpret_cast<double2 *>(p0) = m0;
                t[0].x += m0.x * w[b]; t[0].y += m0.y * w[b];
            }
            acc0 = (t[0].x + t[1].x) + (t[2].x + t[3].x); acc1 = (t[0].y + t[1].y) + (t[2].y + t[3].y);
        } else {
            for (int b = c0 + cl; b < c1; b += 4) {
                double *p0 = Wz + (long long)b * ld + a;
                const double m = *p0 + (-beta * sa0 * v[b] - beta * va0 * s[b] + beta * beta * theta * va0 * v[b] - ca0 * col[b] / w22);
                *p0 = m;
                acc0 += m * w[b];
            }
        }
    }
    sh[cl][2 * lane] = acc0; sh[cl][2 * lane + 1] = acc1;
    __syncthreads();
    if (cl == 0) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int rr = a + k, li = 2 * lane + k;
            if (rr < l) {
                const double t = (sh[0][li] + sh[1][li]) + (sh[2][li] + sh[3][li]);
                if (gridDim.y == 1) out[rr] = alpha * t;
                else part[(long long)blockIdx.y * l + rr] = t;
            }
        }
    }
}
__global__ void __launch_bounds__(NT) k_wz_grow_v(double *__restrict__ Wz, long long ld, int nZ,
                                                  const double *__restrict__ u, const double *__restrict__ scal, int sr) {
    const int a = (blockIdx.x * NT + threadIdx.x) * 2, b0 = blockIdx.y * WZ_COLS;
    if (a > nZ) return;
    const double r2 = scal[sr];
    if (!(r2 > scal[sr + 1])) return;      // not positive definite (the host takes the same decision from its copy): no growth
    auto elem = [&](int aa, int b, double old) -> double {
        if (aa < nZ && b < nZ) return old + u[aa] * u[b] / r2;
        if (aa == nZ && b == nZ) return 1.0 / r2;
        return -u[aa < nZ ? aa : b] / r2;
    };
#pragma unroll
    for (int k = 0; k < WZ_COLS; k++) {
        const int b = b0 + k;
        if (b > nZ) break;
        double *p = Wz + (long long)b * ld + a;
        if (a + 1 <= nZ) {
            double2 m = *reinterpret_cast<double2 *>(p);     // (row / column nZ hold stale values: read, not used)
            m.x = elem(a, b, m.x);
            m.y = elem(a + 1, b, m.y);
            *reinterpret_cast<double2 *>(p) = m;
        } else {
            p[0] = elem(a, b, p[0]);
        }
    }
}
// bordering (null space gains column nZ): u = Wz k, rho2 = kappa - k'u
//   Wz' = [[Wz + u u'/rho2, -u/rho2], [-u'/rho2, 1/rho2]]
__global__ void __launch_bounds__(NT) k_wz_grow(double *__restrict__ Wz, long long ld, int nZ,
                                                const double *__restrict__ u, const double *__restrict__ scal, int sr) {
    const int a = blockIdx.x * NT + threadIdx.x, b = blockIdx.y;
    if (a > nZ || b > nZ) return;
    const double r2 = scal[sr];
    if (!(r2 > scal[sr + 1])) return;      // not positive definite (the host takes the same decision from its copy): no growth
    double val;
    if (a < nZ && b < nZ) val = Wz[(long long)b * ld + a] + u[a] * u[b] / r2;
    else if (a == nZ && b == nZ) val = 1.0 / r2;
    else val = -u[a < nZ ? a : b] / r2;
    Wz[(long long)b * ld + a] = val;
}

// out[j] = A[AC[j]][v]   (column v restricted to the active rows; pos = position of a row in AC or -1)
__global__ void k_col_of_A_active(const int *__restrict__ jc, const int *__restrict__ ir, const double *__restrict__ val,
                                  int v, const int *__restrict__ pos, double *__restrict__ out) {
    for (int k = jc[v] + blockIdx.x * blockDim.x + threadIdx.x; k < jc[v + 1]; k += gridDim.x * blockDim.x) {
        const int p = pos[ir[k]];
        if (p >= 0) out[p] = val[k];
    }
}
// gather / scatter between constraint-indexed vectors and working-set positions
__global__ void k_gather_active(const double *__restrict__ full, const int *__restrict__ AC, int nAC,
                                double *__restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nAC) out[j] = full[AC[j]];
}
__global__ void k_scatter_active(const double *__restrict__ act, const int *__restrict__ AC, int nAC,
                                 double *__restrict__ full) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nAC) full[AC[j]] = act[j];
}

// ---- homotopy element-wise kernels --------------------------------------------------
__device__ inline double delta_of(double target, double cur) {
    return (fabs(target) >= RSQP_INFTY && fabs(cur) >= RSQP_INFTY) ? 0.0 : target - cur;
}

__global__ void k_axpby(int n, double a, const double *__restrict__ x, double b, const double *__restrict__ y,
                        double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a * x[i] + (y ? b * y[i] : 0.0);
}
// dx[v] = free ? xfree[v] : dx[v]
__global__ void k_merge_free(int nV, const int *__restrict__ Sb, const double *__restrict__ xfree, double *__restrict__ dx) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < nV && Sb[v] == 0) dx[v] = xfree[v];
}

// ratio tests: candidate ids as in qp_small.hip; stage 1 per workgroup, stage 2 one workgroup
__device__ inline void cand(double num, double den, int id, double &bt, int &bid) {
    if (den >= RSQP_EPS_DEN) {
        const double t = (num > 0.0 ? num : 0.0) / den;
        if (t < bt || (t == bt && id < bid)) { bt = t; bid = id; }
    }
}
__device__ inline void argmin_reduce(double &t, int &id, double *sht, int *shi) {
    for (int o = 32; o > 0; o >>= 1) {
        const double t2 = __shfl_xor(t, o);
        const int id2 = __shfl_xor(id, o);
        if (t2 < t || (t2 == t && id2 < id)) { t = t2; id = id2; }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sht[threadIdx.x >> 6] = t; shi[threadIdx.x >> 6] = id; }
    __syncthreads();
    t = sht[0]; id = shi[0];
    for (int w = 1; w < NT / 64; w++)
        if (sht[w] < t || (sht[w] == t && shi[w] < id)) { t = sht[w]; id = shi[w]; }
}
__global__ void __launch_bounds__(NT)
k_ratio1(int nV, int nC, const int *__restrict__ Sb, const int *__restrict__ Sc, const double *__restrict__ x,
         const double *__restrict__ y, const double *__restrict__ dx, const double *__restrict__ dy,
         const double *__restrict__ Ax, const double *__restrict__ dAx, const double *__restrict__ lb,
         const double *__restrict__ ub, const double *__restrict__ lbA, const double *__restrict__ ubA,
         const double *__restrict__ lbN, const double *__restrict__ ubN, const double *__restrict__ lbAN,
         const double *__restrict__ ubAN, double *__restrict__ pt, int *__restrict__ pid,
         double *__restrict__ ctl, double seqv, double *__restrict__ dev2, int *__restrict__ ticket,
         const double *__restrict__ Hdx, const double *__restrict__ gN, const double *__restrict__ g,
         const double *__restrict__ ATdy, double *__restrict__ dyw) {
    __shared__ double sht[4];
    __shared__ int shi[4];
    __shared__ int s_last;
    double bt = 1.0;
    int bid = 0x7fffffff;
    for (int i = blockIdx.x * NT + threadIdx.x; i < nC + nV; i += gridDim.x * NT) {
        if (i < nC) {
            const double Axi = Ax[i], dA = dAx[i];
            if (Sc[i] != 0) {
                const double yi = y[nV + i], d = dy[nV + i];
                if (Sc[i] == -1) cand(yi, -d, i, bt, bid); else cand(-yi, d, i, bt, bid);
            } else {
                if (lbAN[i] > -RSQP_INFTY) cand(Axi - lbA[i], delta_of(lbAN[i], lbA[i]) - dA, nC + nV + i, bt, bid);
                if (ubAN[i] < RSQP_INFTY) cand(ubA[i] - Axi, dA - delta_of(ubAN[i], ubA[i]), 2 * nC + nV + i, bt, bid);
            }
        } else {
            const int v = i - nC;
            if (Sb[v] != 0) {
                // the multiplier step of a fixed variable is formed here (it was the launch k_dy_fixed2) and stored for the
                // homotopy step; the free variables keep the zero written with dx_FX
                const double yi = y[v], d = (Hdx[v] + (gN[v] - g[v])) - ATdy[v];
                dyw[v] = d;
                if (Sb[v] == -1) cand(yi, -d, nC + v, bt, bid); else cand(-yi, d, nC + v, bt, bid);
            } else {
                if (lbN[v] > -RSQP_INFTY) cand(x[v] - lb[v], delta_of(lbN[v], lb[v]) - dx[v], 3 * nC + nV + v, bt, bid);
                if (ubN[v] < RSQP_INFTY) cand(ub[v] - x[v], dx[v] - delta_of(ubN[v], ub[v]), 3 * nC + 2 * nV + v, bt, bid);
            }
        }
    }
    if (!(bt < 1.0)) { bt = 1.0; bid = 0x7fffffff; }
    argmin_reduce(bt, bid, sht, shi);
    // The workgroup that draws the last ticket finishes the reduction (what k_argmin2 did in a second launch): partial
    // results are released at device scope before the ticket is drawn and read back with device-scope loads -- the
    // workgroups sit on different XCDs, whose L2s are not coherent for plain loads.
    if (threadIdx.x == 0) {
        pt[blockIdx.x] = bt; pid[blockIdx.x] = bid;
        __threadfence();
        s_last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    bt = RSQP_INFTY * 10.0; bid = 0x7fffffff;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += NT) {
        const double t2 = __hip_atomic_load(pt + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int id2 = __hip_atomic_load(pid + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t2 < bt || (t2 == bt && id2 < bid)) { bt = t2; bid = id2; }
    }
    argmin_reduce(bt, bid, sht, shi);
    if (threadIdx.x == 0) {
        *ticket = 0;                                             // next use (stream order: nobody draws before this kernel ends)
        if (dev2) { dev2[0] = bt; dev2[1] = (double)bid; }       // device copy: the homotopy step is launched before the host has looked
        ctl[0] = bt; ctl[1] = (double)bid; publish(ctl, seqv);   // ctl: host-mapped decision block
    }
}
__global__ void __launch_bounds__(NT) k_argmin2(int n, const double *__restrict__ pt, const int *__restrict__ pid,
                                                double *__restrict__ ctl, double seqv, double *__restrict__ dev2) {
    __shared__ double sht[4];
    __shared__ int shi[4];
    double bt = RSQP_INFTY * 10.0;
    int bid = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += NT)
        if (pt[i] < bt || (pt[i] == bt && pid[i] < bid)) { bt = pt[i]; bid = pid[i]; }
    argmin_reduce(bt, bid, sht, shi);
    if (threadIdx.x == 0) {
        if (dev2) { dev2[0] = bt; dev2[1] = (double)bid; }       // device copy: the homotopy step is launched before the host has looked
        ctl[0] = bt; ctl[1] = (double)bid; publish(ctl, seqv);   // ctl: host-mapped decision block
    }
}

__global__ void k_axpy(int n, double a, const double *__restrict__ x, double *__restrict__ y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += a * x[i];
}
// drift correction pieces
__global__ void k_fix_x(int nV, const int *__restrict__ Sb, const double *__restrict__ lb, const double *__restrict__ ub,
                        double *__restrict__ x) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < nV && Sb[v] != 0) x[v] = Sb[v] == -1 ? lb[v] : ub[v];
}
__global__ void k_set1(double *p, int i, double v) { p[i] = v; }
__global__ void k_copy1(double *dst, int di, const double *src, int si) { dst[di] = src[si]; }

// exchange partner search (ensure_LI): xiC over constraints (by index), xiB over variables
__global__ void __launch_bounds__(NT)
k_partner1(int nV, int nC, const int *__restrict__ Sb, const int *__restrict__ Sc, const double *__restrict__ y,
           const double *__restrict__ xiC, const double *__restrict__ xiB, double sgn, double *__restrict__ pt,
           int *__restrict__ pid) {
    __shared__ double sht[4];
    __shared__ int shi[4];
    double bt = RSQP_INFTY;
    int bid = 0x7fffffff;
    for (int i = blockIdx.x * NT + threadIdx.x; i < nC + nV; i += gridDim.x * NT) {
        const int s = i < nC ? Sc[i] : Sb[i - nC];
        if (s == 0) continue;
        const double xi = sgn * (i < nC ? xiC[i] : xiB[i - nC]), yi = i < nC ? y[nV + i] : y[i - nC];
        const double num = s == -1 ? yi : -yi, den = s == -1 ? xi : -xi;
        if (den > RSQP_EPS_DEN) {
            const double t = (num > 0.0 ? num : 0.0) / den;
            if (t < bt || (t == bt && i < bid)) { bt = t; bid = i; }
        }
    }
    argmin_reduce(bt, bid, sht, shi);
    if (threadIdx.x == 0) { pt[blockIdx.x] = bt; pid[blockIdx.x] = bid; }
}
// y -= t * sgn * xi on the active entries
__global__ void k_shift_duals(int nV, int nC, const int *__restrict__ Sb, const int *__restrict__ Sc, double t, double sgn,
                              const double *__restrict__ xiC, const double *__restrict__ xiB, double *__restrict__ y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nV + nC) return;
    if (i < nV) { if (Sb[i] != 0) y[i] -= t * sgn * xiB[i]; }
    else if (Sc[i - nV] != 0) y[i] -= t * sgn * xiC[i - nV];
}
// xiB[v] = fixed ? a[v] - ATxi[v] : 0
__global__ void k_xiB(int nV, const int *__restrict__ Sb, const double *__restrict__ a, const double *__restrict__ ATxi,
                      double *__restrict__ xiB) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < nV) xiB[v] = Sb[v] != 0 ? a[v] - ATxi[v] : 0.0;
}

// auxiliary-QP data (setup_aux tail)
__global__ void k_aux_v(int nV, const int *__restrict__ Sb, const double *__restrict__ x, const double *__restrict__ ATy,
                        const double *__restrict__ y, const double *__restrict__ Hx, const double *__restrict__ lbN,
                        const double *__restrict__ ubN, double *__restrict__ g, double *__restrict__ lb,
                        double *__restrict__ ub) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nV) return;
    const double xv = x[v];
    g[v] = ATy[v] + y[v] - Hx[v];
    lb[v] = Sb[v] == -1 ? xv : fmin(lbN[v], xv - RSQP_BOUND_RELAXATION);
    ub[v] = Sb[v] == 1 ? xv : fmax(ubN[v], xv + RSQP_BOUND_RELAXATION);
}
__global__ void k_aux_c(int nC, const int *__restrict__ Sc, const double *__restrict__ Ax, const double *__restrict__ lbAN,
                        const double *__restrict__ ubAN, double *__restrict__ lbA, double *__restrict__ ubA) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nC) return;
    const double ax = Ax[i];
    lbA[i] = Sc[i] == -1 ? ax : fmin(lbAN[i], ax - RSQP_BOUND_RELAXATION);
    ubA[i] = Sc[i] == 1 ? ax : fmax(ubAN[i], ax + RSQP_BOUND_RELAXATION);
}
__global__ void k_clamp_copy(int n, const double *__restrict__ src, double *__restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = fmin(fmax(src[i], -RSQP_INFTY), RSQP_INFTY);
}
__global__ void k_rerelax(int n, const int *__restrict__ S, const double *__restrict__ pos, const double *__restrict__ loN,
                          const double *__restrict__ hiN, double *__restrict__ lo, double *__restrict__ hi) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (S[i] != -1 && lo[i] <= -RSQP_INFTY && loN[i] > -RSQP_INFTY) lo[i] = fmin(loN[i], pos[i] - RSQP_BOUND_RELAXATION);
    if (S[i] != 1 && hi[i] >= RSQP_INFTY && hiN[i] < RSQP_INFTY) hi[i] = fmax(hiN[i], pos[i] + RSQP_BOUND_RELAXATION);
}

// ---- fused helpers (fewer launches per working-set change) -------------------------------
// At n <= 4096 the engine is bound by the NUMBER of launches (every kernel, however small, occupies the stream for
// 2.5 - 5 us: rocprofv3 of the dense 2048 x 4096 cold start showed 51 launches = 259 us per working-set change, 37 %
// of it in element-wise and single-thread kernels), so neighbouring element-wise steps share a launch.
// step direction: bA[j] = delta b(AC[j]) - (A dx_FX)[AC[j]]  and  tmpg = H dx_FX + (gN - g)
__global__ void k_sd_prep(int nAC, const int *__restrict__ AC, const int *__restrict__ Sc, const double *__restrict__ lbA,
                          const double *__restrict__ ubA, const double *__restrict__ lbAN, const double *__restrict__ ubAN,
                          const double *__restrict__ Adx, double *__restrict__ bA, int nV, const double *__restrict__ Hdxfx,
                          const double *__restrict__ gN, const double *__restrict__ g, double *__restrict__ tmpg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nAC) {
        const int r = AC[i];
        bA[i] = (Sc[r] == -1 ? delta_of(lbAN[r], lbA[r]) : delta_of(ubAN[r], ubA[r])) - Adx[r];
    }
    if (i < nV) tmpg[i] = Hdxfx[i] + (gN[i] - g[i]);
}
// The step direction behind an ADDED constraint, carried instead of recomputed, in ONE launch over max(nV, 1) entries
// (+ one workgroup). Range space: the right-hand sides of the constraints that were active before have (1 - tau) of their way left,
// the new row of Minv is -xi/eta | 1/eta (minv_append):
//   wY[0..k) *= (1 - tau);  wY[k] = (bA[k] - xi'bA[0..k)) / eta  (also into scal[sl]);  xY = (1 - tau) xY + wY[k] y_k
// with bA from k_sd_prep and y_k the column the constraint added to Y. Every workgroup forms the dot product xi'bA itself (same
// loop, same tree: the same bits in all of them) and then its slice of xY; the LAST workgroup has no slice (the grid is one
// workgroup larger than the slices need): it writes wY and carries the null-space part.
// (Forming bA inside this kernel as well -- five gathers through AC per entry, by every workgroup -- measured 13 us at nV = 2048
//  and 37 us at nV = 10 000 against 4.6 + ~6 us for the two kernels: profiles/r04_b_kernel_stats_large_*.csv.)
// Null space (l >= 0): with P = I - beta v v' the reflection of Z, wZ' = P wZ, col = the last column of P Wz P (k_wz_lastcol),
// kappa = col[0..l) / col[l]:
//   wZ[0..l) <- (1 - tau) (wZ'[0..l) - kappa wZ'[l]) + wY_k kappa
// (block elimination of the last null-space column from Z'HZ wZ = -Z'(g~ + H xY), whose right-hand side gained wY_k Z'H y_k).
__global__ void __launch_bounds__(NT) k_carry_add(int k, double om, const double *__restrict__ bA, const double *__restrict__ xi,
                                                  double *__restrict__ wY, double *__restrict__ scal, int se, int sl, int nV,
                                                  const double *__restrict__ yk, double *__restrict__ xY, int l,
                                                  const double *__restrict__ v, const double *__restrict__ col, double *__restrict__ wZ,
                                                  int sb) {
    __shared__ double sh[4];
    double s = lane_sum4(k, [&](int j) { return xi[j] * bA[j]; });
    const bool extra = blockIdx.x == gridDim.x - 1;
    double d = (extra && l >= 0) ? lane_sum4(l + 1, [&](int j) { return v[j] * wZ[j]; }) : 0.0;   // (its loads overlap the first sum's)
    s = block_sum(s, sh);
    const double w = (bA[k] - s) / scal[se];
    if (!extra) {
        const int i = blockIdx.x * NT + threadIdx.x;
        if (i < nV) xY[i] = om * xY[i] + w * yk[i];
        return;
    }
    // the extra workgroup (no slice of its own): wY, then wZ
    for (int j = threadIdx.x; j < k; j += NT) wY[j] *= om;
    if (threadIdx.x == 0) { wY[k] = w; scal[sl] = w; }
    if (l >= 0) {
        d = block_sum(d, sh);
        const double c = scal[sb] * d, wl = wZ[l] - c * v[l], rc = 1.0 / col[l];
        __syncthreads();
        for (int j = threadIdx.x; j < l; j += NT) {
            const double kap = col[j] * rc;
            wZ[j] = om * ((wZ[j] - c * v[j]) - kap * wl) + w * kap;
        }
    }
}
// ... and over a REMOVED constraint (remove_constraint_tq: Y <- Y P, Minv <- P Minv with P = I - beta v v', then the last column of
// Y moves to Z): wY <- (1 - tau) (P wY)[0..k), xY <- (1 - tau) (xY - omega z_new) with omega = (P wY)[k]  (one workgroup; omega
// into scal[sl])
__global__ void __launch_bounds__(NT) k_carry_remove_wY(int k, double om, const double *__restrict__ v, double *__restrict__ wY,
                                                        double *__restrict__ scal, int sb, int sl) {
    __shared__ double sh[4];
    double d = lane_sum4(k + 1, [&](int i) { return v[i] * wY[i]; });
    d = block_sum(d, sh);
    const double c = scal[sb] * d;
    if (threadIdx.x == 0) scal[sl] = wY[k] - c * v[k];
    for (int i = threadIdx.x; i < k; i += NT) wY[i] = om * (wY[i] - c * v[i]);
}
__global__ void k_carry_remove_xY(int n, double om, const double *__restrict__ znew, const double *__restrict__ scal, int sl,
                                  double *__restrict__ xY) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) xY[i] = om * (xY[i] - scal[sl] * znew[i]);
}
// ... and behind a REMOVED constraint (the null space gained the column z, Wz its border with u = Wz Z'Hz and rho^2, wz_grow): with
// s = z'(H dx_old + (gN - g)_old) -- the reduced gradient of the previous step along z, (gN - g)_old = (gN - g) / (1 - tau) -- and
// omega the component xY lost (k_carry_remove_wY):
//   wZ[0..n) <- (1 - tau) (wZ + (s / rho^2) u),   wZ[n] = (1 - tau) (omega - s / rho^2)
// (block solve of the bordered system). One workgroup.
__global__ void __launch_bounds__(NT) k_carry_wZ_grow(int n, int nV, double om, const double *__restrict__ z, const double *__restrict__ Hdx,
                                                      const double *__restrict__ gN, const double *__restrict__ g, const double *__restrict__ u,
                                                      double *__restrict__ wZ, const double *__restrict__ scal, int srho, int somega) {
    __shared__ double sh[4];
    const double iom = 1.0 / om;
    double d = lane_sum4(nV, [&](int i) { return z[i] * (Hdx[i] + (gN[i] - g[i]) * iom); });
    d = block_sum(d, sh);
    const double c = d / scal[srho];
    for (int i = threadIdx.x; i < n; i += NT) wZ[i] = om * (wZ[i] + c * u[i]);
    if (threadIdx.x == 0) wZ[n] = om * (scal[somega] - c);
}
// the whole homotopy step in one launch: variables, constraints, multipliers, and -- fix != 0 -- the blocking
// bound / constraint put exactly on its new side (fkind 3: constraint fidx, 4: variable fidx)
// (the winner of the ratio test is decoded HERE from the device copy k_argmin2 left in res[0..1], exactly as the host
// decodes its own copy: the launch does not wait for the host round trip)
__global__ void k_step_all(int nV, int nC, const double *__restrict__ res, int allow_fix, const int *__restrict__ Sb, double *__restrict__ x,
                           double *__restrict__ g, double *__restrict__ lb, double *__restrict__ ub,
                           const double *__restrict__ gN, const double *__restrict__ lbN, const double *__restrict__ ubN,
                           const double *__restrict__ dx, const double *__restrict__ ATdy, double *__restrict__ ATy,
                           const double *__restrict__ Hdx, double *__restrict__ Hx, double *__restrict__ lbA,
                           double *__restrict__ ubA, const double *__restrict__ lbAN, const double *__restrict__ ubAN,
                           const double *__restrict__ dAx, double *__restrict__ Ax, const double *__restrict__ dy,
                           double *__restrict__ y, double *__restrict__ rsp = nullptr, double *__restrict__ rsAp = nullptr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double tau = res[0];
    const int bid = (int)res[1];
    int fkind = 0, fidx = -1, fside = 0;
    if (bid != 0x7fffffff) {
        if (bid < nC) { fkind = 1; fidx = bid; }
        else if (bid < nC + nV) { fkind = 2; fidx = bid - nC; }
        else if (bid < 2 * nC + nV) { fkind = 3; fidx = bid - nC - nV; fside = -1; }
        else if (bid < 3 * nC + nV) { fkind = 3; fidx = bid - 2 * nC - nV; fside = 1; }
        else if (bid < 3 * nC + 2 * nV) { fkind = 4; fidx = bid - 3 * nC - nV; fside = -1; }
        else { fkind = 4; fidx = bid - 3 * nC - 2 * nV; fside = 1; }
    } else tau = 1.0;
    const int done = fkind == 0;
    const int fix = !done && allow_fix && (fkind == 3 || fkind == 4);
    if (i < nV) {
        const int v = i;
        if (done) {
            g[v] = gN[v]; lb[v] = lbN[v]; ub[v] = ubN[v];
            x[v] = Sb[v] == -1 ? lb[v] : (Sb[v] == 1 ? ub[v] : x[v] + tau * dx[v]);
        } else {
            x[v] += tau * dx[v];
            g[v] += tau * (gN[v] - g[v]);
            lb[v] += tau * delta_of(lbN[v], lb[v]);
            ub[v] += tau * delta_of(ubN[v], ub[v]);
            ATy[v] += tau * ATdy[v];
            Hx[v] += tau * Hdx[v];
            if (fix && fkind == 4 && v == fidx) { if (fside == -1) lb[v] = x[v]; else ub[v] = x[v]; }
            if (rsp) rsp[v] *= 1.0 - tau;      // (general range-space path: p = H^-1 (gN - g) shrinks like gN - g itself)
        }
    }
    if (i < nC) {
        if (done) { lbA[i] = lbAN[i]; ubA[i] = ubAN[i]; }
        else {
            lbA[i] += tau * delta_of(lbAN[i], lbA[i]);
            ubA[i] += tau * delta_of(ubAN[i], ubA[i]);
            Ax[i] += tau * dAx[i];
            if (fix && fkind == 3 && i == fidx) { if (fside == -1) lbA[i] = Ax[i]; else ubA[i] = Ax[i]; }
            if (rsAp) rsAp[i] *= 1.0 - tau;
        }
    }
    if (i < nV + nC) y[i] += tau * dy[i];
}
// products of variable v with the bases -- rows v of Z and Y -- and the norms of the independence test, one workgroup
__global__ void __launch_bounds__(NT) k_bound_products(const double *__restrict__ Z, const double *__restrict__ Y, long long ld,
                                                       int v, int nZ, int nAC, double *__restrict__ wz1, double *__restrict__ a1,
                                                       double *__restrict__ scal, int s1, int s2, double *__restrict__ ctl, double seqv,
                                                       double *__restrict__ hv) {
    __shared__ double sh[4];
    double q = lane_sum4(nZ, [&](int c) { const double t = Z[c * ld + v]; wz1[c] = t; return t * t; });
    for (int c = threadIdx.x; c < nAC; c += NT) a1[c] = Y[c * ld + v];
    q = block_sum(q, sh);
    if (threadIdx.x == 0) { scal[s1] = 1.0; scal[s2] = q; ctl[2] = 1.0; ctl[3] = q; publish(ctl, seqv); }
    house_tail(wz1, nZ, q, hv, scal);
}
// a variable joins the fixed set: rows v of both bases are cleared, the working set updated
__global__ void k_clean_bound(double *__restrict__ Y, double *__restrict__ Z, long long ld, int v, int nAC, int nZ,
                              int *__restrict__ Sb, int side) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < nAC) Y[c * ld + v] = 0.0;
    if (c < nZ) Z[c * ld + v] = 0.0;
    if (c == 0) Sb[v] = side;
}
// working-set bookkeeping of a removed constraint (position k, constraint r; the last position moves into k)
__global__ void k_ws_remove_c(int *AC, int *posAC, int *Sc, int k, int last, int r, double *y, int yidx) {
    if (k != last) { const int rl = AC[last]; AC[k] = rl; posAC[rl] = k; }
    posAC[r] = -1; Sc[r] = 0;
    if (y) y[yidx] = 0.0;
}
__global__ void k_free_bound_ws(int *Sb, int v, double *y) { Sb[v] = 0; if (y) y[v] = 0.0; }

// dx on the fixed variables (zero elsewhere) and dy := 0, one launch over nV + nC
__global__ void k_dx_fixed_zero_dy(int nV, int nC, const int *__restrict__ Sb, const double *__restrict__ lb,
                                   const double *__restrict__ ub, const double *__restrict__ lbN,
                                   const double *__restrict__ ubN, double *__restrict__ dx, double *__restrict__ dy) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nV) dx[i] = Sb[i] == -1 ? delta_of(lbN[i], lb[i]) : (Sb[i] == 1 ? delta_of(ubN[i], ub[i]) : 0.0);
    if (i < nV + nC) dy[i] = 0.0;
}
// drift correction without a refresh of the products: x on its bounds, active constraint sides on
// A x, gradient from stationarity -- one launch over max(nV, nC)
__global__ void k_drift_all(int nV, int nC, const int *__restrict__ Sb, const int *__restrict__ Sc,
                            const double *__restrict__ lb, const double *__restrict__ ub, double *__restrict__ x,
                            const double *__restrict__ Ax, double *__restrict__ lbA, double *__restrict__ ubA,
                            const double *__restrict__ ATy, const double *__restrict__ y, const double *__restrict__ Hx,
                            double *__restrict__ g, const double *__restrict__ lbN, const double *__restrict__ ubN,
                            double *__restrict__ dx, double *__restrict__ dy, const double *__restrict__ hinv = nullptr,
                            const double *__restrict__ gN = nullptr, double *__restrict__ rv = nullptr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nV) {
        if (Sb[i] != 0) x[i] = Sb[i] == -1 ? lb[i] : ub[i];
        const double gi = ATy[i] + y[i] - Hx[i];
        g[i] = gi;
        // first kernel of the next step direction (k_dx_fixed_zero_dy) rides along: it reads nothing this one writes
        const double dxi = Sb[i] == -1 ? delta_of(lbN[i], lb[i]) : (Sb[i] == 1 ? delta_of(ubN[i], ub[i]) : 0.0);
        dx[i] = dxi;
        // (range-space path: so does k_dual_rhs_vec -- the vector whose product with A gives the right-hand side of the multiplier step)
        if (rv) rv[i] = Sb[i] == 0 ? hinv[i] * (gN[i] - gi) : -dxi;
    }
    if (i < nC) { if (Sc[i] == -1) lbA[i] = Ax[i]; else if (Sc[i] == 1) ubA[i] = Ax[i]; }
    if (i < nV + nC) dy[i] = 0.0;
}
// a[v] = A[row][v] on the free variables (all != 0: on every variable), zero elsewhere -- from the dense row-major copy, where
// row `row` is contiguous: one entry per thread, any number of workgroups
__global__ void k_row_of_A_dense(const double *__restrict__ AT, int nV, int row, const int *__restrict__ Sb, int all,
                                 double *__restrict__ a) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < nV) a[v] = (all || Sb[v] == 0) ? AT[(long long)row * nV + v] : 0.0;
}
// zero a (length nV) and scatter row `row` of A into it: one workgroup
// (sc / as: optionally the row scaled entry by entry, as[c] = sc[c] a[c] -- D^-1 a_FR of the range-space path -- in the same launch)
__global__ void __launch_bounds__(NT) k_row_of_A_fused(const int *__restrict__ rp, const int *__restrict__ ci,
                                                       const double *__restrict__ rv, int row, const int *__restrict__ Sb,
                                                       int all, int nV, double *__restrict__ a, const double *__restrict__ sc = nullptr,
                                                       double *__restrict__ as = nullptr) {
    for (int v = threadIdx.x; v < nV; v += NT) { a[v] = 0.0; if (as) as[v] = 0.0; }
    __syncthreads();
    for (int k = rp[row] + threadIdx.x; k < rp[row + 1]; k += NT) {
        const int c = ci[k];
        if (all || Sb[c] == 0) { a[c] = rv[k]; if (as) as[c] = sc[c] * rv[k]; }
    }
}
// scal[s1] = |a|^2 (n1 entries; 1.0 if a is null), scal[s2] = |b|^2 (n2 entries); both published to ctl
__global__ void __launch_bounds__(NT) k_norms_publish(const double *__restrict__ a, int n1, const double *__restrict__ b,
                                                      int n2, double *__restrict__ scal, int s1, int s2,
                                                      double *__restrict__ ctl, double seqv, double *__restrict__ hv) {
    __shared__ double sh[4];
    double p = a ? lane_sum4(n1, [&](int i) { return a[i] * a[i]; }) : 0.0;
    double q = lane_sum4(n2, [&](int i) { return b[i] * b[i]; });
    p = a ? block_sum(p, sh) : 1.0;
    q = block_sum(q, sh);
    if (threadIdx.x == 0) { scal[s1] = p; scal[s2] = q; ctl[2] = p; ctl[3] = q; publish(ctl, seqv); }
    house_tail(b, n2, q, hv, scal);
}

// Z[:, k] = e_{free[k]} (columns zero-filled beforehand)
// products of the sparse row r of A with the columns of two bases: out0 = M0' a_r (n0 columns),
// out1 = M1' a_r (n1 columns). One thread per column, the row entries staged in LDS: the row has
// a few dozen entries, so this gathers nnz * 64 B per column instead of streaming nV * 8 B.
// (rows of fixed variables are zero in both bases: no masking needed)
__global__ void __launch_bounds__(256)
k_row_times_bases(const int *__restrict__ Arp, const int *__restrict__ Aci, const double *__restrict__ Arv, int r,
                  const double *__restrict__ M0, int n0, double *__restrict__ out0, const double *__restrict__ M1, int n1,
                  double *__restrict__ out1, long long l) {
    __shared__ int sidx[256];
    __shared__ double sval[256];
    const int nb0 = (n0 + 255) / 256;
    const bool first = (int)blockIdx.x < nb0;
    const double *Mx = first ? M0 : M1;
    const int ncols = first ? n0 : n1;
    double *out = first ? out0 : out1;
    const int c = (first ? blockIdx.x : blockIdx.x - nb0) * 256 + threadIdx.x;
    const int k0 = Arp[r], k1 = Arp[r + 1];
    const double *col = Mx + (long long)(c < ncols ? c : 0) * l;
    double s = 0.0;
    for (int base = k0; base < k1; base += 256) {
        const int cnt = min(256, k1 - base);
        __syncthreads();
        if ((int)threadIdx.x < cnt) { sidx[threadIdx.x] = Aci[base + threadIdx.x]; sval[threadIdx.x] = Arv[base + threadIdx.x]; }
        __syncthreads();
        for (int e = 0; e < cnt; e++) s += sval[e] * col[sidx[e]];
    }
    if (c < ncols) out[c] = s;
}

// blocked set-up: B[:, k] = free part of the row of candidate constraint cand[k] (compressed free coordinates)
__global__ void k_build_B(const int *__restrict__ Arp, const int *__restrict__ Aci, const double *__restrict__ Arv,
                          const int *__restrict__ cand, const int *__restrict__ fpos, double *__restrict__ B, long long ldb) {
    const int r = cand[blockIdx.x];
    double *col = B + (long long)blockIdx.x * ldb;
    for (int k = Arp[r] + threadIdx.x; k < Arp[r + 1]; k += blockDim.x) {
        const int p = fpos[Aci[k]];
        if (p >= 0) col[p] = Arv[k];
    }
}
// rows of the compressed orthogonal factor back to the variables: Y = Q[:, :n], Z = Q[:, n:]
__global__ void k_scatter_Q(int m, int n, const double *__restrict__ Q, long long ldq, const int *__restrict__ freev,
                            double *__restrict__ Y, double *__restrict__ Z, long long ld) {
    const int c = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const double v = Q[i + (long long)c * ldq];
    if (c < n) Y[freev[i] + (long long)c * ld] = v;
    else Z[freev[i] + (long long)(c - n) * ld] = v;
}
// dst (n x n, ldd) = src' (src n x n, lds)
__global__ void k_transpose(int n, const double *__restrict__ src, long long lds, double *__restrict__ dst, long long ldd) {
    __shared__ double tile[32][33];
    const int bi = blockIdx.x * 32, bj = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int i = bi + threadIdx.x, j = bj + r;
        tile[r][threadIdx.x] = (i < n && j < n) ? src[i + (long long)j * lds] : 0.0;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int i = bj + threadIdx.x, j = bi + r;    // dst(i, j) = src(j, i)
        if (i < n && j < n) dst[i + (long long)j * ldd] = tile[threadIdx.x][r];
    }
}
__global__ void k_add_scaled(long long n, double a, const double *__restrict__ x, double *__restrict__ y) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i < n) y[i] += a * x[i];
}

__global__ void k_unit_cols(double *__restrict__ Z, long long ld, const int *__restrict__ freev, int n) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) Z[k * ld + freev[k]] = 1.0;
}

inline dim3 g1(int n) { return dim3((unsigned)((n + NT - 1) / NT)); }

// ---- small single-thread / utility kernels -----------------------------------------------
// The border of Minv behind an added constraint. eta_from_house: eta = image sign * alpha of the last Householder vector
// (scal[2] * scal[0]), else scal[es]; thread nAC also records the new working-set entry (constraint r at position nAC, side).
// keep != nullptr: what the carried step direction needs from the border is kept on the way (the row xi in keep, eta in
// scal[se]); yidx >= 0: the multiplier of the incoming constraint is set (y[yidx] = yval) -- three launches in one
__global__ void k_minv_border_keep(double *Minv, long long ldm, int nAC, const double *row, double *scal, int es, int eta_from_house,
                                   int *AC, int *posAC, int *Sc, int r, int side, double *keep, int se, double *y, int yidx, double yval) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > nAC) return;
    const double eta = eta_from_house ? scal[2] * scal[0] : scal[es];
    if (j == nAC) {
        AC[nAC] = r; posAC[r] = nAC; Sc[r] = side;
        if (keep) scal[se] = eta;
        if (yidx >= 0) y[yidx] = yval;
    }
    if (j < nAC) {
        const double rj = row[j];
        if (keep) keep[j] = rj;
        Minv[(long long)j * ldm + nAC] = -rj / eta;       // new row nAC
        Minv[(long long)nAC * ldm + j] = 0.0;             // new column nAC
    } else {
        Minv[(long long)nAC * ldm + nAC] = 1.0 / eta;
    }
}
// a1[0..nAC) = qY, a1[nAC] = q*: unit vector. vt = q~ with last += sgn(q*); beta~ = 1/(1+|q*|),
// gamma = beta~/(1 - beta~ |qY|^2) = beta~/|q*|
__global__ void k_house_unit(double *a1, int nAC, double *scal, const double *zs, int v) {
    const double qs = zs[v], sg = qs >= 0.0 ? 1.0 : -1.0, aq = fabs(qs);   // q* = zs[v]
    const double beta = 1.0 / (1.0 + aq);
    a1[nAC] = qs + sg;
    scal[8] = beta;
    scal[9] = beta / aq;
    scal[10] = qs + sg;
}
__global__ void k_axpy_s(int n, const double *scal, int si, const double *x, double *y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += scal[si] * x[i];
}
// a1 = c (= Minv a_v) on entry; on exit a1 = vY = -c; nu = sqrt(1+|c|^2); vlast = 1 + nu; beta~ = 1/(nu(nu+1))
__global__ void k_house_free(double *a1, int nAC, double *scal) {
    __shared__ double sh[4];
    double s = lane_sum4(nAC, [&](int i) { return a1[i] * a1[i]; });
    s = block_sum(s, sh);
    for (int i = threadIdx.x; i < nAC; i += NT) a1[i] = -a1[i];
    if (threadIdx.x == 0) {
        const double nu = sqrt(1.0 + s);
        scal[8] = 1.0 / (nu * (nu + 1.0));
        scal[10] = 1.0 + nu;
    }
}
__global__ void k_add_scal_at(double *w, int v, const double *scal, int si) { w[v] += scal[si]; }
__global__ void k_newcol_free(int nV, int v, const double *t, const double *scal, double *znew) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nV) znew[i] = (i == v ? 1.0 : 0.0) - scal[8] * t[i] * scal[10];
}
__global__ void k_sm_coef(double *scal) { scal[16] = scal[8] / (1.0 - scal[8] * scal[15]); }
// kappa = z'(H z), ku = k'u (both dots formed here), rho2 = kappa - ku and its threshold
__global__ void __launch_bounds__(NT) k_rho2(double *scal, double *ctl, const double *__restrict__ z, const double *__restrict__ Hz,
                                             int nV, const double *__restrict__ kv, const double *__restrict__ uv, int nZ, double seqv) {
    __shared__ double sh[4];
    double p = lane_sum4(nV, [&](int i) { return z[i] * Hz[i]; });
    double q = lane_sum4(nZ, [&](int i) { return kv[i] * uv[i]; });
    p = block_sum(p, sh);
    q = block_sum(q, sh);
    if (threadIdx.x != 0) return;
    scal[11] = p; scal[12] = q;
    const double kappa = p, ku = q;
    scal[13] = kappa - ku;
    scal[14] = RSQP_EPS_PD_REL * (fabs(kappa) + fabs(ku)) + RSQP_EPS_PD_ABS;
    ctl[4] = scal[13]; ctl[5] = scal[14];
    publish(ctl, seqv);
}
// ---- the range-space (dual) path for a DIAGONAL positive Hessian (Impl::dual) ---------------------------------
// hinv = 1 / (diag(H) + hreg); flag[0] |= 1 when an entry is not positive (the caller then keeps the null-space path)
__global__ void k_dual_hinv(int nV, const double *__restrict__ Hval, double hreg, double *__restrict__ hinv, int *__restrict__ flag) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nV) return;
    const double d = Hval[v] + hreg;
    hinv[v] = d > 0.0 ? 1.0 / d : 0.0;
    if (!(d > 1e-300)) atomicOr(flag, 1);
}
// out = D^-1 a on the free variables, 0 elsewhere (a is zero on the fixed ones already)
__global__ void k_dual_scale(int nV, const double *__restrict__ hinv, const double *__restrict__ a, double *__restrict__ out) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < nV) out[v] = hinv[v] * a[v];
}
// the unit vector of variable v on the free variables (a_FR of a bound) and D^-1 of it
__global__ void k_dual_unit(int nV, int v, const int *__restrict__ Sb, const double *__restrict__ hinv, double *__restrict__ a,
                            double *__restrict__ da) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nV) return;
    const double e = (i == v && Sb[i] == 0) ? 1.0 : 0.0;
    a[i] = e; da[i] = hinv[i] * e;
}
// independence of an incoming row a from the active ones, decided on first-order quantities: with c = A_AC,FR D^-1 a_FR,
// u = Sinv c (the coefficients of a's representation by the active rows in the D^-1 metric) and atu = A_AC' u:
//   |a_FR|^2, |r|^2 with r = (a - atu) on the free variables (Z'r = Z'a, |Z'a| <= |r| <= cond(D) |Z'a|), and the pivot of the
//   bordering s = a_FR'D^-1 a_FR - c'u (second order: only its sign and size as a pivot are used). One workgroup; published.
__global__ void __launch_bounds__(1024) k_dual_li_publish(int nV, const int *__restrict__ Sb, const double *__restrict__ a,
                                                          const double *__restrict__ da, const double *__restrict__ atu, int k,
                                                          const double *__restrict__ c, const double *__restrict__ u,
                                                          double *__restrict__ scal, double *__restrict__ ctl, double seqv) {
    // (1024 lanes, two independent accumulators each: with 256 lanes the three sums over nV = 10 000 took 39 us)
    __shared__ double sh[4][16];
    double a2 = 0.0, r2 = 0.0, ad = 0.0, cu = 0.0, a2b = 0.0, r2b = 0.0, adb = 0.0, cub = 0.0;
    int i = threadIdx.x;
    for (; i + 1024 < nV; i += 2048) {
        const double x0 = a[i], x1 = a[i + 1024];
        const double q0 = (atu && Sb[i] == 0) ? x0 - atu[i] : 0.0, q1 = (atu && Sb[i + 1024] == 0) ? x1 - atu[i + 1024] : 0.0;
        a2 += x0 * x0; a2b += x1 * x1; r2 += q0 * q0; r2b += q1 * q1; ad += x0 * da[i]; adb += x1 * da[i + 1024];
    }
    if (i < nV) { const double x0 = a[i], q0 = (atu && Sb[i] == 0) ? x0 - atu[i] : 0.0; a2 += x0 * x0; r2 += q0 * q0; ad += x0 * da[i]; }
    int j = threadIdx.x;
    for (; j + 1024 < k; j += 2048) { cu += c[j] * u[j]; cub += c[j + 1024] * u[j + 1024]; }
    if (j < k) cu += c[j] * u[j];
    double v4[4] = {a2 + a2b, r2 + r2b, ad + adb, cu + cub};
#pragma unroll
    for (int q = 0; q < 4; q++) {
        double v = v4[q];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) sh[q][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double t[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { double v = 0.0; for (int w = 0; w < 16; w++) v += sh[q][w]; t[q] = v; }
    const double sp = t[2] - t[3];
    scal[6] = t[0]; scal[7] = t[1]; scal[5] = sp; scal[8] = sp != 0.0 ? 1.0 / sp : 0.0;
    ctl[2] = t[0]; ctl[3] = t[1]; ctl[4] = sp; ctl[5] = t[2];
    publish(ctl, seqv);
}
// the multiplier step CARRIED over a plain added constraint (the right-hand sides of the rows that were active before have
// (1 - tau) of their way left): Sinv_new rhs_new = [om dy - lam u; lam], lam = (rhs_k - om c'dy) / s -- O(nAC) instead of a pass over
// Sinv; k = the position of the new constraint r, Av = A (D^-1 dg_FR - dx_FX) of the new step. One workgroup.
// 1024 lanes (a dot product over up to 10 000 rows by 256 lanes is 40 dependent loads: 11 us); the ONE entry of A (D^-1 dg_FR - dx_FX)
// the new row needs is formed here from its sparse row instead of by a full product with A in front of the kernel (round 5: a
// carried step then runs two sparse products, not three)
__device__ __forceinline__ double block_sum_w(double v, double *sh16) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh16[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < 16; w++) s += sh16[w];
    return s;
}
__global__ void __launch_bounds__(1024) k_dual_carry_add_w(int k, double om, const double *__restrict__ c, const double *__restrict__ u,
                                                           double *__restrict__ dyv, const double *__restrict__ scal, int r,
                                                           const int *__restrict__ Sc, const double *__restrict__ lbA,
                                                           const double *__restrict__ ubA, const double *__restrict__ lbAN,
                                                           const double *__restrict__ ubAN, const int *__restrict__ rp,
                                                           const int *__restrict__ ci, const double *__restrict__ rv,
                                                           const double *__restrict__ wvec, const int *__restrict__ AC,
                                                           double *__restrict__ dyC) {
    __shared__ double sh[16];
    // (tried: everything the update loop reads fetched into registers before the two reductions -- 17.7 us against 11.5 us,
    //  profiles/r05_s_kernel_stats_large_sparse.csv: reverted)
    double t = 0.0, a = 0.0;
    for (int j = threadIdx.x; j < k; j += 1024) t += c[j] * dyv[j];
    for (int e = rp[r] + threadIdx.x; e < rp[r + 1]; e += 1024) a += rv[e] * wvec[ci[e]];
    t = block_sum_w(t, sh);
    a = block_sum_w(a, sh);
    const double rk = (Sc[r] == -1 ? delta_of(lbAN[r], lbA[r]) : delta_of(ubAN[r], ubA[r])) + a;
    const double lam = (rk - om * t) * scal[8];
    for (int j = threadIdx.x; j < k; j += 1024) { const double d = om * dyv[j] - lam * u[j]; dyv[j] = d; dyC[AC[j]] = d; }
    if (threadIdx.x == 0) { dyv[k] = lam; dyC[r] = lam; }
}
// ... and over a plain removed constraint at position j (v = column j of Sinv before the update): om (dy - (dy_j / v_j) v) with
// entry j dropped and the last one moved into its slot. One workgroup.
__global__ void __launch_bounds__(NT) k_dual_carry_remove(int k, int j, double om, const double *__restrict__ v, double *__restrict__ dyv) {
    const int last = k - 1;
    const double f = dyv[j] / v[j];
    const double tl = om * (dyv[last] - f * v[last]);
    __syncthreads();
    for (int i = threadIdx.x; i < last; i += NT)
        if (i != j) dyv[i] = om * (dyv[i] - f * v[i]);
    if (threadIdx.x == 0 && j != last) dyv[j] = tl;
}
// the border of Sinv behind an added constraint (after Sinv += u u' / s): row and column nAC = -u / s, corner 1 / s; thread nAC
// records the working-set entry and the multiplier of the incoming constraint (yidx >= 0)
__global__ void k_dual_border(double *Sinv, long long ldm, int nAC, const double *__restrict__ u, const double *__restrict__ scal,
                              int *AC, int *posAC, int *Sc, int r, int side, double *y, int yidx, double yval) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > nAC) return;
    const double is = scal[8];
    if (j == nAC) {
        AC[nAC] = r; posAC[r] = nAC; Sc[r] = side;
        if (yidx >= 0) y[yidx] = yval;
        Sinv[(long long)nAC * ldm + nAC] = is;
    } else {
        const double b = -u[j] * is;
        Sinv[(long long)j * ldm + nAC] = b;
        Sinv[(long long)nAC * ldm + j] = b;
    }
}
// the rank-1 part of a bordering is DEFERRED (it rides on the next product with Sinv, k_ger_gemv_n1): u and 1 / s out of the way
// of the next products, zero behind the block it applies to (the kernel then updates rows / columns 0..n) only)
__global__ void k_dual_keep_u(int n, const double *__restrict__ u, double *__restrict__ keep, double *__restrict__ scal, int to) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) keep[j] = u[j];
    if (j == n) { keep[n] = 0.0; keep[n + 1] = 0.0; scal[to] = scal[8]; }
}
// ---- Sinv is symmetric: stored in its UPPER triangle only (entry (i, j), i <= j, at M[j ld + i]), so that a rank-1 update and a
// product touch half the bytes. One workgroup per 64 x 64 tile (I <= J) of the triangle: optional update M += coef u u', optional
// contributions to y = M w -- the tile's rows against w_J (into P1[J][.]) and, transposed, its columns against w_I (into
// P2[I][.]; a diagonal tile holds i <= j only and gives its diagonal to the row part). k_sym_reduce adds the partials in tile
// order (deterministic). The tile passes through LDS once: coalesced column loads / stores, both products from there.
constexpr int SYT = 64;
__device__ __forceinline__ void sym_tile_of(int t, int &I, int &J) {
    J = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while (J * (J + 1) / 2 > t) J--;
    while ((J + 1) * (J + 2) / 2 <= t) J++;
    I = t - J * (J + 1) / 2;
}
template <bool UPD, bool MV>
__global__ void __launch_bounds__(256) k_sym_tile(double *__restrict__ M, long long ld, int n, const double *__restrict__ u,
                                                  const double *__restrict__ scal, int ci, double cs, const double *__restrict__ w,
                                                  double *__restrict__ P1, double *__restrict__ P2) {
    __shared__ double S[SYT][SYT + 1];
    __shared__ double wI[SYT], wJ[SYT], rp[2][SYT], cp[2][SYT];
    int I, J;
    sym_tile_of((int)blockIdx.x, I, J);
    const bool diag = I == J;
    const int ii = threadIdx.x & 63, wv = threadIdx.x >> 6, i = I * SYT + ii;
    const double ui = (UPD && i < n) ? cs * scal[ci] * u[i] : 0.0;
    if (MV && threadIdx.x < SYT) { wI[threadIdx.x] = i < n ? w[i] : 0.0; const int j = J * SYT + (int)threadIdx.x; wJ[threadIdx.x] = j < n ? w[j] : 0.0; }
    if (!UPD) {
        // read-only product (the lazy form of round 5): all loads of a lane are issued before the first LDS store -- with the
        // 4-deep unroll of the updating form a compute unit had ~32 KB in flight and the kernel read at 2.8 TB/s. 16 bytes per lane
        // (two rows of a column; 8-byte loads reach 0.54-0.70 of the 16-byte rate on this part, MI355X_MICROARCH.md): lane pair-of-
        // rows r2 = t & 31 of column 8 c + (t >> 5); the tile arrives in LDS as before, so the sums keep their order bit for bit
        if ((ld & 1) == 0 && (reinterpret_cast<unsigned long long>(M) & 15) == 0) {
            const int r2 = threadIdx.x & 31, cg = threadIdx.x >> 5, i0 = I * SYT + 2 * r2;
            double2 m2[8];
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const int j = J * SYT + c * 8 + cg;
                m2[c] = (i0 < n && j < n) ? *reinterpret_cast<const double2 *>(M + (long long)j * ld + i0) : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const int jj = c * 8 + cg, j = J * SYT + jj;
                const bool cj = j < n;
                S[jj][2 * r2] = (cj && i0 < n && (!diag || 2 * r2 <= jj)) ? m2[c].x : 0.0;
                S[jj][2 * r2 + 1] = (cj && i0 + 1 < n && (!diag || 2 * r2 + 1 <= jj)) ? m2[c].y : 0.0;
            }
        } else {
        double mm[16];
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const int jj = wv * 16 + c, j = J * SYT + jj;
            mm[c] = (i < n && j < n && (!diag || ii <= jj)) ? M[(long long)j * ld + i] : 0.0;
        }
#pragma unroll
        for (int c = 0; c < 16; c++) S[wv * 16 + c][ii] = mm[c];
        }
    } else {
#pragma unroll 4
    for (int c = 0; c < 16; c++) {
        const int jj = wv * 16 + c, j = J * SYT + jj;
        double m = 0.0;
        if (i < n && j < n && (!diag || ii <= jj)) {
            double *p = M + (long long)j * ld + i;
            m = *p;
            if (UPD) { m += ui * u[j]; *p = m; }
        }
        if (MV) S[jj][ii] = m;
    }
    }
    if (!MV) return;
    __syncthreads();
    const int h = (threadIdx.x >> 6) & 1, e = threadIdx.x & 63;
    double a = 0.0;
    if (threadIdx.x < 128) {
#pragma unroll 8
        for (int q = 0; q < 32; q++) { const int jj = 32 * h + q; a += S[jj][e] * wJ[jj]; }       // row e of the tile against w_J
        rp[h][e] = a;
    } else {
#pragma unroll 8
        for (int q = 0; q < 32; q++) { const int r = 32 * h + q; a += (diag && r == e) ? 0.0 : S[e][r] * wI[r]; }   // column e against w_I
        cp[h][e] = a;
    }
    __syncthreads();
    if (threadIdx.x < SYT) { if (i < n) P1[(long long)J * n + i] = rp[0][threadIdx.x] + rp[1][threadIdx.x]; }
    else if (threadIdx.x < 2 * SYT) { const int j = J * SYT + e; if (j < n) P2[(long long)I * n + j] = cp[0][e] + cp[1][e]; }
}
// (R / full: the result scattered by row id as well -- the multiplier step of the general range-space path)
__global__ void k_sym_reduce(int n, int nt, const double *__restrict__ P1, const double *__restrict__ P2, double *__restrict__ y,
                             const int *__restrict__ R = nullptr, double *__restrict__ full = nullptr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int Ti = i / SYT;
    // (four independent accumulators per sum: up to 2 x 120 partials per entry, one dependent load chain each measured 22 us)
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
    int J = Ti;
    for (; J + 3 < nt; J += 4) {
        a0 += P1[(long long)J * n + i]; a1 += P1[(long long)(J + 1) * n + i]; a2 += P1[(long long)(J + 2) * n + i]; a3 += P1[(long long)(J + 3) * n + i];
    }
    for (; J < nt; J++) a0 += P1[(long long)J * n + i];
    int I = 0;
    for (; I + 3 <= Ti; I += 4) {
        b0 += P2[(long long)I * n + i]; b1 += P2[(long long)(I + 1) * n + i]; b2 += P2[(long long)(I + 2) * n + i]; b3 += P2[(long long)(I + 3) * n + i];
    }
    for (; I <= Ti; I++) b0 += P2[(long long)I * n + i];
    const double val = ((a0 + a1) + (a2 + a3)) + ((b0 + b1) + (b2 + b3));
    y[i] = val;
    if (R) full[R[i]] = val;
}
// The same tile scheme for Wz = (Z'HZ)^-1 of the NULL-space path (symmetric as well; upper triangle from wz_sym_min variables on).
// UPD 2: the elimination step behind a reflected null space (k_wz_shrink's expression) on the leading l x l block; UPD 3: the
// rank-1 part of a bordering, u u' / rho^2, under the kernel's own definiteness guard (scal[i0] > scal[i0 + 1]: k_wz_grow's);
// UPD 0: product only. alpha scales the product in k_sym_reduce_a.
struct SymW { const double *u, *s, *v, *col, *scal; int i0, i1, l; };
template <int UPD, bool MV>
__global__ void __launch_bounds__(256) k_symw_tile(double *__restrict__ M, long long ld, int n, SymW q, const double *__restrict__ w,
                                                   double *__restrict__ P1, double *__restrict__ P2) {
    __shared__ double S[SYT][SYT + 1];
    __shared__ double wI[SYT], wJ[SYT], rp[2][SYT], cp[2][SYT];
    int I, J;
    sym_tile_of((int)blockIdx.x, I, J);
    const bool diag = I == J;
    const int ii = threadIdx.x & 63, wv = threadIdx.x >> 6, i = I * SYT + ii;
    double beta = 0.0, theta = 0.0, iw22 = 0.0, ir2 = 0.0;
    bool upd = UPD != 0;
    if (UPD == 2) { beta = q.scal[q.i0]; theta = q.scal[q.i1]; iw22 = 1.0 / q.col[q.l]; }
    if (UPD == 3) { const double r2 = q.scal[q.i0]; upd = r2 > q.scal[q.i0 + 1]; ir2 = 1.0 / r2; }
    const double si = (UPD == 2 && i < n) ? q.s[i] : 0.0, vi = (UPD == 2 && i < n) ? q.v[i] : 0.0, ci = (UPD == 2 && i < n) ? q.col[i] : 0.0;
    const double ui = (UPD == 3 && i < n) ? q.u[i] : 0.0;
    if (MV && threadIdx.x < SYT) { wI[threadIdx.x] = i < n ? w[i] : 0.0; const int j = J * SYT + (int)threadIdx.x; wJ[threadIdx.x] = j < n ? w[j] : 0.0; }
#pragma unroll 4
    for (int c = 0; c < 16; c++) {
        const int jj = wv * 16 + c, j = J * SYT + jj;
        double m = 0.0;
        if (i < n && j < n && (!diag || ii <= jj)) {
            double *p = M + (long long)j * ld + i;
            m = *p;
            if (UPD == 2) { const double vj = q.v[j]; m += -beta * si * vj - beta * vi * q.s[j] + beta * beta * theta * vi * vj - ci * q.col[j] * iw22; *p = m; }
            if (UPD == 3 && upd) { m += ui * q.u[j] * ir2; *p = m; }
        }
        if (MV) S[jj][ii] = m;
    }
    if (!MV) return;
    __syncthreads();
    const int h = (threadIdx.x >> 6) & 1, e = threadIdx.x & 63;
    double a = 0.0;
    if (threadIdx.x < 128) {
#pragma unroll 8
        for (int t = 0; t < 32; t++) { const int jj = 32 * h + t; a += S[jj][e] * wJ[jj]; }
        rp[h][e] = a;
    } else {
#pragma unroll 8
        for (int t = 0; t < 32; t++) { const int r = 32 * h + t; a += (diag && r == e) ? 0.0 : S[e][r] * wI[r]; }
        cp[h][e] = a;
    }
    __syncthreads();
    if (threadIdx.x < SYT) { if (i < n) P1[(long long)J * n + i] = rp[0][threadIdx.x] + rp[1][threadIdx.x]; }
    else if (threadIdx.x < 2 * SYT) { const int j = J * SYT + e; if (j < n) P2[(long long)I * n + j] = cp[0][e] + cp[1][e]; }
}
__global__ void k_sym_reduce_a(int n, int nt, const double *__restrict__ P1, const double *__restrict__ P2, double alpha, double *__restrict__ y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int Ti = i / SYT;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
    int J = Ti;
    for (; J + 3 < nt; J += 4) {
        a0 += P1[(long long)J * n + i]; a1 += P1[(long long)(J + 1) * n + i]; a2 += P1[(long long)(J + 2) * n + i]; a3 += P1[(long long)(J + 3) * n + i];
    }
    for (; J < nt; J++) a0 += P1[(long long)J * n + i];
    int I = 0;
    for (; I + 3 <= Ti; I += 4) {
        b0 += P2[(long long)I * n + i]; b1 += P2[(long long)(I + 1) * n + i]; b2 += P2[(long long)(I + 2) * n + i]; b3 += P2[(long long)(I + 3) * n + i];
    }
    for (; I <= Ti; I++) b0 += P2[(long long)I * n + i];
    y[i] = alpha * (((a0 + a1) + (a2 + a3)) + ((b0 + b1) + (b2 + b3)));
}
// the border of a grown Wz in upper storage: column nZ = -u / rho^2, corner 1 / rho^2 (same guard as the rank-1 part)
__global__ void k_wz_grow_col_sym(double *__restrict__ Wz, long long ld, int nZ, const double *__restrict__ u, const double *__restrict__ scal, int sr) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a > nZ) return;
    const double r2 = scal[sr];
    if (!(r2 > scal[sr + 1])) return;
    Wz[(long long)nZ * ld + a] = a == nZ ? 1.0 / r2 : -u[a] / r2;
}
// (upper-triangle storage) column j of the symmetric matrix and the coefficient of its removal
__global__ void k_dual_colcoef_sym(const double *__restrict__ Sinv, long long ldm, int k, int j, double *__restrict__ v, double *__restrict__ scal) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) v[i] = i <= j ? Sinv[(long long)j * ldm + i] : Sinv[(long long)i * ldm + j];
    if (i == 0) { const double d = Sinv[(long long)j * ldm + j]; scal[9] = d != 0.0 ? -1.0 / d : 0.0; }
}
// (keep != nullptr: k_dual_keep_u in the same launch -- u and 1 / s kept for the deferred rank-1 part, zero behind the block)
__global__ void k_dual_border_sym(double *Sinv, long long ldm, int nAC, const double *__restrict__ u, double *__restrict__ scal,
                                  int *AC, int *posAC, int *Sc, int r, int side, double *y, int yidx, double yval,
                                  double *__restrict__ keep, int to) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > nAC) return;
    const double is = scal[8];
    if (j == nAC) {
        AC[nAC] = r; posAC[r] = nAC; Sc[r] = side;
        if (yidx >= 0) y[yidx] = yval;
        Sinv[(long long)nAC * ldm + nAC] = is;
        if (keep) { keep[nAC] = 0.0; keep[nAC + 1] = 0.0; scal[to] = is; }
    } else {
        const double uj = u[j];
        Sinv[(long long)nAC * ldm + j] = -uj * is;
        if (keep) keep[j] = uj;
    }
}
__global__ void k_dual_move_last_sym(double *Sinv, long long ldm, int k, int j, int *AC, int *posAC, int *Sc, int r, double *y, int yidx) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int last = k - 1;
    if (c == 0) {
        if (j != last) { const int rl = AC[last]; AC[j] = rl; posAC[rl] = j; }
        posAC[r] = -1; Sc[r] = 0;
        y[yidx] = 0.0;
    }
    if (j == last || c >= last) return;
    const double val = c == j ? Sinv[(long long)last * ldm + last] : Sinv[(long long)last * ldm + c];
    const int lo_ = c < j ? c : j, hi_ = c < j ? j : c;
    Sinv[(long long)hi_ * ldm + lo_] = val;
}
// removal of the constraint at position j: v = column j of Sinv, coef = -1 / v_j  (Sinv += coef v v' zeroes row and column j)
__global__ void k_dual_colcoef(const double *__restrict__ Sinv, long long ldm, int k, int j, double *__restrict__ v, double *__restrict__ scal) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) v[i] = Sinv[(long long)j * ldm + i];
    if (i == 0) { const double d = Sinv[(long long)j * ldm + j]; scal[9] = d != 0.0 ? -1.0 / d : 0.0; }
}
// ... then the last row / column (k - 1) moves into the freed slot j (Sinv symmetric: both from column k - 1), working set follows
__global__ void k_dual_move_last(double *Sinv, long long ldm, int k, int j, int *AC, int *posAC, int *Sc, int r, double *y, int yidx) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int last = k - 1;
    if (c == 0) {
        if (j != last) { const int rl = AC[last]; AC[j] = rl; posAC[rl] = j; }
        posAC[r] = -1; Sc[r] = 0;
        y[yidx] = 0.0;
    }
    if (j == last || c >= last) return;
    const double val = c == j ? Sinv[(long long)last * ldm + last] : Sinv[(long long)last * ldm + c];
    Sinv[(long long)c * ldm + j] = val;
    Sinv[(long long)j * ldm + c] = val;
}
// a variable is freed: S gains a_v a_v' / d_v, Sinv -= w w' / (d_v + a_v'w) with w = Sinv a_v: the coefficient (one workgroup)
__global__ void __launch_bounds__(NT) k_dual_free_coef(int k, const double *__restrict__ av, const double *__restrict__ w,
                                                       const double *__restrict__ Hval, double hreg, int v, double *__restrict__ scal) {
    __shared__ double sh[4];
    double d = lane_sum4(k, [&](int j) { return av[j] * w[j]; });
    d = block_sum(d, sh);
    if (threadIdx.x == 0) scal[9] = -1.0 / (Hval[v] + hreg + d);
}
// step direction: the vector whose product with A gives the right-hand side of S dy = db - A dx_FX + A D^-1 dg_FR
__global__ void k_dual_rhs_vec(int nV, const int *__restrict__ Sb, const double *__restrict__ hinv, const double *__restrict__ gN,
                               const double *__restrict__ g, const double *__restrict__ dx, double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nV) out[i] = Sb[i] == 0 ? hinv[i] * (gN[i] - g[i]) : -dx[i];
}
__global__ void k_dual_rhs(int nAC, const int *__restrict__ AC, const int *__restrict__ Sc, const double *__restrict__ lbA,
                           const double *__restrict__ ubA, const double *__restrict__ lbAN, const double *__restrict__ ubAN,
                           const double *__restrict__ Av, double *__restrict__ rhs) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nAC) return;
    const int r = AC[j];
    rhs[j] = (Sc[r] == -1 ? delta_of(lbAN[r], lbA[r]) : delta_of(ubAN[r], ubA[r])) + Av[r];
}
// dx on the free variables from stationarity, D dx + dg = A'dy_C; H dx of every variable
__global__ void k_dual_dx(int nV, const int *__restrict__ Sb, const double *__restrict__ hinv, const double *__restrict__ Hval,
                          double hreg, const double *__restrict__ ATdy, const double *__restrict__ gN, const double *__restrict__ g,
                          double *__restrict__ dx, double *__restrict__ Hdx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nV) return;
    double d = dx[i];
    if (Sb[i] == 0) { d = hinv[i] * (ATdy[i] - (gN[i] - g[i])); dx[i] = d; }
    Hdx[i] = (Hval[i] + hreg) * d;
}
// rows of B (compressed free coordinates, one column per candidate constraint) scaled by sqrt(1 / d): S = B'B
__global__ void k_dual_scale_rows(int m, const int *__restrict__ freev, const double *__restrict__ hinv, double *__restrict__ B, long long ldb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) B[(long long)blockIdx.y * ldb + i] *= sqrt(hinv[freev[i]]);
}
__global__ void k_set_Sb(int *Sb, int v, int side) { Sb[v] = side; }
__global__ void k_clip_y(int nV, int nC, const int *Sb, const int *Sc, double *y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nV + nC) return;
    const int s = i < nV ? Sb[i] : Sc[i - nV];
    const double yi = y[i];
    if (s == 0 || (s == -1 && yi < 0.0) || (s == 1 && yi > 0.0)) y[i] = 0.0;
}

#include "qp_rs_kernels.h"

}  // namespace

// =====================================================================================
// host-side engine
// =====================================================================================
#define LCHK(call)                                             \
    do {                                                       \
        hipError_t e_ = (call);                                \
        if (e_ != hipSuccess) { err_ = e_; return RET_SETUP_FAILED; } \
    } while (0)

template <class T>
static hipError_t dalloc(T **p, size_t n) {
    hipError_t e = hipMalloc(reinterpret_cast<void **>(p), std::max<size_t>(n, 1) * sizeof(T));
    if (e != hipSuccess) return e;
    return hipMemset(*p, 0, std::max<size_t>(n, 1) * sizeof(T));
}

struct RsqpLargeEngine::Impl {
    int nV = 0, nC = 0, nAmax = 0;
    long long ld = 0, ldm = 0;
    // (sizes below 256 keep their own leading dimension: alignment buys nothing there, and the even-ld kernel variants sum in
    //  another order -- on the reference's non-convex dump hs107 that changed which side a flip takes and the run cycled)
    static long long pad16(long long n) { const bool off = getenv("RSQP_LARGE_NO_PAD") != nullptr; return (off || n < 256) ? n : ((n + 15) & ~15LL); }      // (read where a handle is created: init / bytes_needed)
    hipStream_t st = nullptr;
    hipError_t err_ = hipSuccess;
    RsqpLargeMatrices M;
    // dense state
    double *Z = nullptr, *Y = nullptr, *Minv = nullptr, *Wz = nullptr;
    // vectors (nV)
    double *x, *g, *lb, *ub, *gN, *lbN, *ubN, *dx, *w1, *w2, *w3, *w4, *w5, *w6, *wz1, *wz2, *wz3;
    double *pz_t = nullptr, *pz_v = nullptr, *pw_s = nullptr, *pw_col = nullptr;   // operands of a deferred reflection (z_reflect_and_shrink)
    double *c_wY = nullptr, *c_wY2 = nullptr, *c_xY = nullptr, *c_xi = nullptr, *c_wZ = nullptr;   // (c_wY2: the other half of c_wY's double buffer, k_remove_tail)
    double *py_t = nullptr, *py_v = nullptr, *pm_s = nullptr;                     // operands of the deferred reflections of Y / Minv (removed constraint)                    // range-space part of the step direction, carried over an added constraint
    // vectors (nC)
    double *Ax, *lbA, *ubA, *lbAN, *ubAN, *dAx, *c1, *c2, *c3, *a1, *a2, *a3, *a4;
    double *y, *dy, *part, *scal, *pt, *res_t;
    int *Sall = nullptr;        // working-set status of [variables; constraints] in ONE array (Sb = Sall, Sc = Sall + nV): the general
                                // range-space path addresses both kinds of row by one id
    int *Sb, *Sc, *AC, *posAC, *pid, *res_id;
    // host mirrors
    std::vector<int> hSb, hSc, hAC;
    int nFR = 0, nAC = 0, nZ = 0;
    int status = QPS_NOTINITIALISED, infeasible = 0, unbounded = 0, nflips = 0;
    double *h_ctl = nullptr, *d_ctl = nullptr;   // host-mapped decision block written by kernels
    unsigned long long ctl_seq = 0;              // number of the last publication launched (see publish())
    int spin_misses = 0;                         // publications that only became visible at the end of the kernel's stream
    bool spin_wait = getenv("RSQP_LARGE_NO_SPIN") == nullptr;
    double next_seq() { return (double)(++ctl_seq); }
    // wait for the last launched publication: spin on its sequence number for up to 5 ms, then (or when spinning is
    // off) block in hipStreamSynchronize. A platform on which the number never shows up early turns the spinning off.
    double wait_seconds = 0.0;                   // host time spent waiting for publications (RSQP_PROFILE report)
    long long wait_calls = 0;
    int wait_ctl() {
        struct Acc { Impl *p; double t0; ~Acc() { p->wait_seconds += now_s() - t0; p->wait_calls++; } } acc{this, now_s()};
        if (spin_wait) {
            const double want = (double)ctl_seq;
            double *flag = h_ctl + 63;
            const double t0 = now_s();
            for (int it = 0;; it++) {
                // acquire load of the sequence word: the decision block behind it is read after it
                unsigned long long bits = __atomic_load_n(reinterpret_cast<unsigned long long *>(flag), __ATOMIC_ACQUIRE);
                double seen; std::memcpy(&seen, &bits, sizeof(seen));
                if (seen == want) { spin_misses = 0; return RET_OK; }     // consecutive misses only (a long blocked set-up is not a reason to stop spinning)
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
                if ((it & 1023) == 1023 && now_s() - t0 > 5e-3) break;
            }
            if (++spin_misses >= 16) spin_wait = false;
        }
        LCHK(hipStreamSynchronize(st));
        return RET_OK;
    }
    // a publication that never arrived (device error): the deferred rank-1 updates of Z / Wz / Y / Minv were not applied
    // while nZ / nAC already count the change -- the stored factors are inconsistent, so the handle forgets them and the
    // next hot start falls back to a full set-up
    int wait_failed() {
        status = QPS_NOTINITIALISED;
        pendZ.on = pendW.on = pendY.on = pendM.on = pendS.on = pendR.on = false;
        lz_n = 0;
        return RET_SETUP_FAILED;
    }
    double *h_pinned = nullptr;  // small pinned read-back buffer
    int *h_pinned_i = nullptr;
    int nblk_ratio = 0;
    long long part_cap = 0;
    // maintained products (refreshed exactly every REFRESH working-set changes)
    double *ATy = nullptr, *Hx = nullptr, *Hdx = nullptr, *ATdy = nullptr;
    bool dirty_products = true;
    bool wz_enabled = true;   // false while setup_aux builds Z/Y/Minv; Wz is bordered afterwards
    int since_refresh = 0;
    // blocked (GEMM) set-up of a non-empty working set: dense_la.hip
    RsqpDenseWork dw;
    double *big = nullptr;    // 2 nV^2 scratch, allocated by the first blocked set-up
    int *d_fpos = nullptr, *d_cand = nullptr, *d_freev = nullptr;
    bool blocked_setup = getenv("RSQP_NO_BLOCKED_SETUP") == nullptr;
    // rows up to which y = M w runs as ONE launch (k_gemv_n1); beyond, chunks + a reduction launch. Measured
    // (tools/large_kernel_bench.py): 4096 x 4096 19.8 vs 27.8 us, 5000 x 5000 44.2 vs 35.6 us, 7670 x 7670 116.9 vs 84.5 us
    int n1_maxrows = getenv("RSQP_GEMV_N1_MAXROWS") ? atoi(getenv("RSQP_GEMV_N1_MAXROWS")) : 4096;
    bool dx_ready = false;
    bool house_done = false;      // wz2 / scal[0..3] hold the Householder data of the current wz1 (constraint_products / bound_products)
    bool live_skip = getenv("RSQP_LARGE_NO_LIVE_SKIP") == nullptr;
    bool extra_sync = getenv("RSQP_LARGE_EXTRA_SYNC") != nullptr;
    int n1_threads = getenv("RSQP_GEMV_N1_THREADS") ? atoi(getenv("RSQP_GEMV_N1_THREADS")) : 0;   // tuning: force 256 / 512 threads in k_gemv_n1
    int gemv_wgs = getenv("RSQP_GEMV_WGS") ? atoi(getenv("RSQP_GEMV_WGS")) : 4096;     // workgroups the chunked y = M w aims for
    bool reinit_from_y0 = false;
    // the last blocked set-up (rsqp_get_setup_profile): HIP-event time and algorithmic flops of its two parts
    struct SetupStat { int valid = 0, m = 0, n = 0, nZ = 0, dual = 0; float ms_tq = 0.f, ms_wz = 0.f; double flops_tq = 0.0, flops_wz = 0.0; };
    SetupStat setup_stat;
    hipEvent_t se0 = nullptr, se1 = nullptr, se2 = nullptr;
    static constexpr int BLOCKED_MIN = 32;   // fewer active constraints: the sequential construction is as fast

    ~Impl() {
        double *dv[] = {Z, Y, Minv, Wz, x, g, lb, ub, gN, lbN, ubN, dx, w1, w2, w3, w4, w5, w6, wz1, wz2, wz3, Ax, lbA, ubA,
                        lbAN, ubAN, dAx, c1, c2, c3, a1, a2, a3, a4, y, dy, part, scal, pt, res_t, ATy, Hx, Hdx, ATdy, pz_t, pz_v, pw_s, pw_col, c_wY, c_wY2, c_xY, c_xi, c_wZ, py_t, py_v, pm_s, hinv, ps_u, sym_part, wz_part,
                        rs_p, ra1, ra2, ra3, ra4, rs_dl, rs_ps_u, rs_G, band_buf, rs_WW, rs_dA, lz_vec, lz_c, lz_d};
        for (double *p : dv) if (p) (void)hipFree(p);
        if (big) (void)hipFree(big);
        rsqp_dense_work_free(&dw);
        int *iv[] = {Sall, AC, posAC, pid, res_id, d_fpos, d_cand, d_freev, dflag, R, posR};
        for (int *p : iv) if (p) (void)hipFree(p);
        if (h_ctl) (void)hipHostFree(h_ctl);
        if (h_pinned) (void)hipHostFree(h_pinned);
        if (h_pinned_i) (void)hipHostFree(h_pinned_i);
        preport();
    }

    // ---- launch helpers -----------------------------------------------------------
    bool debug = getenv("RSQP_DEBUG") != nullptr;
    // RSQP_PROFILE=1: per kernel class, HIP-event time and algorithmic bytes (serialises the stream)
    bool profile = getenv("RSQP_PROFILE") != nullptr;
    struct Prof { double ms = 0, bytes = 0; long long calls = 0; };
    Prof prof[10];
    hipEvent_t pe0 = nullptr, pe1 = nullptr;
    static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void pbegin() { if (profile) { if (!pe0) { (void)hipEventCreate(&pe0); (void)hipEventCreate(&pe1); } (void)hipEventRecord(pe0, st); } }
    void pend(int cls, double bytes) {
        if (!profile) return;
        (void)hipEventRecord(pe1, st); (void)hipEventSynchronize(pe1);
        float ms = 0; (void)hipEventElapsedTime(&ms, pe0, pe1);
        prof[cls].ms += ms; prof[cls].bytes += bytes; prof[cls].calls++;
    }
    void preport() {
        if (!profile) return;
        const char *nm[10] = {"gemv_n", "gemv_t", "ger", "wz_shrink", "wz_grow", "spmv", "ger_gemv_t", "shrink_gemv", "ger_gemv_n", "band_hinv"};
        for (int k = 0; k < 10; k++)
            if (prof[k].calls) fprintf(stderr, "[rsqp profile] %-9s calls %8lld  time %9.3f ms  bytes %10.3f GB  => %7.1f GB/s  (avg %.1f us)\n", nm[k], prof[k].calls, prof[k].ms, prof[k].bytes / 1e9, prof[k].bytes / (prof[k].ms * 1e-3) / 1e9, 1e3 * prof[k].ms / prof[k].calls);
    }
    void chk(const char *what) {
        if (!debug) return;
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { fprintf(stderr, "[rsqp large] %s: %s (nV=%d nC=%d nZ=%d nAC=%d)\n", what, hipGetErrorString(e), nV, nC, nZ, nAC); debug = false; }
    }
    static GtTask gt_task(const double *Mx, long long l, int nrows, int ncols, const double *xv, double *out) {
        GtTask t;
        t.M = Mx; t.ld = l; t.nrows = nrows; t.ncols = ncols; t.x = xv; t.xa = t.xb = t.addv = nullptr; t.omap = nullptr; t.out = out;
        return t;
    }
    void gemv_t_task(const GtTask &t) {
        pbegin();
        if (t.ncols > 0) hipLaunchKernelGGL(k_gemv_t, dim3((t.ncols + GT_COLS - 1) / GT_COLS), dim3(NT), 0, st, t);
        pend(1, 8.0 * t.nrows * (double)t.ncols + 8.0 * t.nrows + 8.0 * t.ncols);
        chk("gemv_t");
    }
    void gemv_t(const double *Mx, long long l, int nrows, int ncols, const double *xv, double *out) {
        gemv_t_task(gt_task(Mx, l, nrows, ncols, xv, out));
    }
    // two independent products, one launch
    void gemv_t_pair(const GtTask &t0, const GtTask &t1) {
        if (t0.ncols <= 0) { gemv_t_task(t1); return; }
        if (t1.ncols <= 0) { gemv_t_task(t0); return; }
        pbegin();
        const int nb0 = (t0.ncols + GT_COLS - 1) / GT_COLS, nb1 = (t1.ncols + GT_COLS - 1) / GT_COLS;
        hipLaunchKernelGGL(k_gemv_t2, dim3(nb0 + nb1), dim3(NT), 0, st, t0, nb0, t1);
        pend(1, 8.0 * t0.nrows * (double)t0.ncols + 8.0 * t1.nrows * (double)t1.ncols);
        chk("gemv_t_pair");
    }
    // out = beta*base + alpha * M w
    // mSb / mdst: optional epilogue mdst[r] = out[r] where mSb[r] == 0 (k_merge_free fused into the product)
    void gemv_n(const double *Mx, long long l, int nrows, int ncols, const double *wv, double alpha, double beta,
                const double *base, double *out, const int *mSb = nullptr, double *mdst = nullptr) {
        if (nrows <= 0) return;
        if (ncols <= 0) {
            if (base && beta != 0.0) hipLaunchKernelGGL(k_axpby, g1(nrows), dim3(NT), 0, st, nrows, beta, base, 0.0, (const double *)nullptr, out);
            else hipLaunchKernelGGL(k_fill, g1(nrows), dim3(NT), 0, st, out, nrows, 0.0);
            if (mSb) hipLaunchKernelGGL(k_merge_free, g1(nrows), dim3(NT), 0, st, nrows, mSb, out, mdst);
            return;
        }
        pbegin();
        if (nrows <= n1_maxrows || mSb) {   // launch-bound sizes: one kernel, no partials (k_gemv_n1)
            const bool vec1 = ((l & 1) == 0) && ((reinterpret_cast<unsigned long long>(Mx) & 15) == 0);
            // 16-row workgroups: up to 3072 rows they cover at most 192 of the 256 CUs, so each gets 512 threads (64 column
            // groups, twice the loads in flight: 2048 x 2048 8.7 -> 6.7 us); taller matrices fill the chip with 256 (4096 x
            // 4096: 19.9 us vs 21.4 with 512). 1024 threads measured 2x slower. (tools/large_kernel_bench.py)
            if (vec1 && (n1_threads == 512 || (n1_threads == 0 && nrows <= 3072))) hipLaunchKernelGGL((k_gemv_n1<true, 512>), dim3((nrows + 15) / 16), dim3(512), 0, st, Mx, l, nrows, ncols, wv, alpha, beta, base, out, mSb, mdst);
            else if (vec1) hipLaunchKernelGGL((k_gemv_n1<true, NT>), dim3((nrows + 15) / 16), dim3(NT), 0, st, Mx, l, nrows, ncols, wv, alpha, beta, base, out, mSb, mdst);
            else hipLaunchKernelGGL((k_gemv_n1<false, NT>), dim3((nrows + 7) / 8), dim3(NT), 0, st, Mx, l, nrows, ncols, wv, alpha, beta, base, out, mSb, mdst);
            pend(0, 8.0 * nrows * (double)ncols + 8.0 * nrows + 8.0 * ncols);
            chk("gemv_n1");
            return;
        }
        // chunks: ~4096 workgroups (at most 32 partials per row). Measured on 10 000 x 10 000 (tools/large_kernel_bench.py):
        // 768 workgroups 152 us, 2048 145 us, 4096 (capped: 2528) 136 us = 5.9 TB/s -- more loads in flight per CU
        const bool vec = ((l & 1) == 0) && ((reinterpret_cast<unsigned long long>(Mx) & 15) == 0) && nrows >= 128;
        const int rb = vec ? (nrows + 127) / 128 : (nrows + 63) / 64;
        int nch = std::max(1, std::min(std::min((gemv_wgs + rb - 1) / rb, 32), (ncols + 15) / 16));
        while ((long long)nch * nrows > part_cap) nch = (nch + 1) / 2;
        const int chunk = (ncols + nch - 1) / nch;
        nch = (ncols + chunk - 1) / chunk;
        if (vec)
            hipLaunchKernelGGL(k_gemv_n_part<true>, dim3(rb, nch), dim3(NT), 0, st, Mx, l, nrows, ncols, chunk, wv, part,
                               alpha, beta, base, out);
        else
            hipLaunchKernelGGL(k_gemv_n_part<false>, dim3(rb, nch), dim3(NT), 0, st, Mx, l, nrows, ncols, chunk, wv, part,
                               alpha, beta, base, out);
        if (nch > 1)
            hipLaunchKernelGGL(k_gemv_n_reduce, dim3((nrows + 63) / 64), dim3(NT), 0, st, part, nrows, nch, alpha, beta, base, out);
        pend(0, 8.0 * nrows * (double)ncols + 8.0 * nrows + 8.0 * ncols);
        chk("gemv_n");
    }
    // two products out = M w of the single-launch form in ONE launch (k_gemv_n1_pair); false: the pair does not qualify and the
    // caller launches them one after the other. skip0: columns with a zero weight are not read (live of them are non-zero)
    // ---- range-space (dual) path: H diagonal and positive. S = A_AC,FR D^-1 A_AC,FR' and its EXPLICIT inverse Sinv
    // (nAC x nAC, symmetric, in the buffer of Minv) replace Z, Y, Minv, Wz: per working-set change one rank-1 update of Sinv and
    // one or two products with it -- nAC^2 entries instead of nV nZ + nZ^2 + nAC^2 + nV nAC. Same homotopy, ratio tests, exchange
    // rule and drift correction; the definiteness guard of a removal never fires (Z'DZ is positive definite for every Z).
    struct { bool on = false; int n = 0; } pendS;      // Sinv[0..n) x [0..n) += scal[S_KEEP_S] ps_u ps_u' not applied yet
    static constexpr int S_KEEP_S = 45;
    double *ps_u = nullptr;
    bool dual_defer = getenv("RSQP_LARGE_NO_FUSE") == nullptr;
    // Sinv symmetric: upper triangle only (k_sym_tile); RSQP_LARGE_NO_SYM=1: full storage with the GEMV / GER kernels of the null-space path
    bool dual_sym = getenv("RSQP_LARGE_NO_SYM") == nullptr;
    double *sym_part = nullptr;      // partial sums of the tiled symmetric product: [2][tiles per side][nAmax]
    static int sym_tiles(int n) { const int nt = (n + SYT - 1) / SYT; return nt * (nt + 1) / 2; }
    // Sinv[0..n)^2 += (cs scal[slot]) v v'
    void dual_rank1(int n, const double *v, int slot, double cs) {
        if (n <= 0) return;
        if (!dual_sym) { ger(Minv, ldm, n, n, v, v, slot, cs); return; }
        pbegin();
        hipLaunchKernelGGL((k_sym_tile<true, false>), dim3(sym_tiles(n)), dim3(256), 0, st, Minv, ldm, n, v, scal, slot, cs, (const double *)nullptr,
                           (double *)nullptr, (double *)nullptr);
        pend(2, 8.0 * n * (double)n);
    }
    bool dual_lazy() const { return lz_enabled && dual_sym; }      // lazy rank-1 terms (qp_rs_kernels.h: LzP) on the triangle storage
    void dual_flush() {
        if (rsh) return;      // (the pending terms belong to the general path's matrix: rs_flush)
        if (lz_n > 0 && nAC > 0 && dual_lazy()) {
            pbegin();
            hipLaunchKernelGGL(k_sym_tile_lz, dim3(sym_tiles(nAC)), dim3(256), 0, st, Minv, ldm, nAC, lz_p());
            pend(2, 8.0 * nAC * (double)nAC);
        }
        lz_n = 0;
        if (!pendS.on) return;
        pendS.on = false;
        dual_rank1(pendS.n, ps_u, S_KEEP_S, 1.0);
    }
    // out = Sinv w, the deferred rank-1 update applied on the way (one read + one write of Sinv instead of read + write + read)
    void dual_sinv_times(const double *wv, double *out) {
        if (dual_sym) {
            if (nAC <= 0) return;
            const int nt = (nAC + SYT - 1) / SYT;
            double *P1 = sym_part, *P2 = sym_part + (size_t)nt * nAC;
            if (dual_lazy()) {      // one read of the triangle; the pending terms in the reduction
                if (lz_n > 0) hipLaunchKernelGGL(k_lz_dots, dim3(lz_n), dim3(NT), 0, st, lz_p(), nAC, wv, lz_d);
                pbegin();
                hipLaunchKernelGGL((k_sym_tile<false, true>), dim3(sym_tiles(nAC)), dim3(256), 0, st, Minv, ldm, nAC, (const double *)nullptr, scal, 0, 0.0,
                                   wv, P1, P2);
                pend(0, 4.0 * nAC * (double)nAC);
                hipLaunchKernelGGL(k_sym_reduce_lz, g1(nAC), dim3(NT), 0, st, nAC, nt, P1, P2, out, (const int *)nullptr, (double *)nullptr, lz_p(), lz_d);
                chk("dual lazy product");
                return;
            }
            pbegin();
            if (pendS.on && nAC == pendS.n + 1) {     // (ps_u is zero behind pendS.n: the update leaves the new border alone)
                pendS.on = false;
                hipLaunchKernelGGL((k_sym_tile<true, true>), dim3(sym_tiles(nAC)), dim3(256), 0, st, Minv, ldm, nAC, ps_u, scal, S_KEEP_S, 1.0, wv, P1, P2);
                pend(8, 8.0 * nAC * (double)nAC);
            } else {
                dual_flush();
                hipLaunchKernelGGL((k_sym_tile<false, true>), dim3(sym_tiles(nAC)), dim3(256), 0, st, Minv, ldm, nAC, (const double *)nullptr, scal, 0, 0.0,
                                   wv, P1, P2);
                pend(0, 4.0 * nAC * (double)nAC);
            }
            hipLaunchKernelGGL(k_sym_reduce, g1(nAC), dim3(NT), 0, st, nAC, nt, P1, P2, out);
            chk("dual sym product");
            return;
        }
        if (pendS.on && nAC == pendS.n + 1 && (ldm & 1) == 0 && ((reinterpret_cast<unsigned long long>(Minv) | reinterpret_cast<unsigned long long>(ps_u)) & 15) == 0) {
            pendS.on = false;
            pbegin();
            hipLaunchKernelGGL((k_ger_gemv_n1<NT>), dim3((nAC + 15) / 16), dim3(NT), 0, st, Minv, ldm, nAC, nAC, ps_u, ps_u, scal, S_KEEP_S, 1.0, wv,
                               1.0, 0.0, (const double *)nullptr, out, (const int *)nullptr, (double *)nullptr);
            pend(8, 16.0 * nAC * (double)nAC);
            chk("dual ger_gemv_n1");
            return;
        }
        dual_flush();
        gemv_n(Minv, ldm, nAC, nAC, wv, 1.0, 0.0, nullptr, out);
    }
    bool dual_enabled = getenv("RSQP_LARGE_NO_DUAL") == nullptr;
    bool dual = false;               // the factors of this handle are Sinv (set by setup_aux; a hot start on new vectors keeps it)
    double *hinv = nullptr;          // 1 / (diag(H) + hreg)
    int *dflag = nullptr;
    bool pair_enabled = getenv("RSQP_LARGE_NO_PAIR") == nullptr;
    bool tail_fused = getenv("RSQP_LARGE_NO_TAIL") == nullptr;      // (k_remove_tail; off: the separate kernels)
    bool gemv_n1_pair(const double *M0, long long l0, int r0, int c0, const double *w0, double *o0, const double *M1, long long l1,
                      int r1, int c1, const double *w1, double *o1, bool skip0 = false, int live = 0) {
        if (!pair_enabled || r0 <= 0 || r1 <= 0 || c0 <= 0 || c1 <= 0 || r0 > n1_maxrows || r1 > n1_maxrows) return false;
        if ((l0 & 1) || (l1 & 1) || ((reinterpret_cast<unsigned long long>(M0) | reinterpret_cast<unsigned long long>(M1)) & 15)) return false;
        const bool t512 = n1_threads == 512 || (n1_threads == 0 && std::max(r0, r1) <= 3072);
        const int nb0 = (r0 + 15) / 16, nb1 = (r1 + 15) / 16;
        pbegin();
        if (skip0) {
            if (t512) hipLaunchKernelGGL((k_gemv_n1_pair<512, true>), dim3(nb0 + nb1), dim3(512), 0, st, M0, l0, r0, c0, w0, o0, nb0, M1, l1, r1, c1, w1, o1);
            else hipLaunchKernelGGL((k_gemv_n1_pair<NT, true>), dim3(nb0 + nb1), dim3(NT), 0, st, M0, l0, r0, c0, w0, o0, nb0, M1, l1, r1, c1, w1, o1);
            pend(0, 8.0 * (r0 + (double)r1) * live + 8.0 * (r0 + r1) + 8.0 * (c0 + c1));
        } else {
            if (t512) hipLaunchKernelGGL((k_gemv_n1_pair<512, false>), dim3(nb0 + nb1), dim3(512), 0, st, M0, l0, r0, c0, w0, o0, nb0, M1, l1, r1, c1, w1, o1);
            else hipLaunchKernelGGL((k_gemv_n1_pair<NT, false>), dim3(nb0 + nb1), dim3(NT), 0, st, M0, l0, r0, c0, w0, o0, nb0, M1, l1, r1, c1, w1, o1);
            pend(0, 8.0 * r0 * (double)c0 + 8.0 * r1 * (double)c1 + 8.0 * (r0 + r1) + 8.0 * (c0 + c1));
        }
        chk("gemv_n1_pair");
        return true;
    }
    // out = M w for a vector w with `live` non-zeros (the kernel skips the other columns); falls back to gemv_n
    void gemv_n_live(const double *Mx, long long l, int nrows, int ncols, const double *wv, double *out, int live) {
        const bool vec1 = ((l & 1) == 0) && ((reinterpret_cast<unsigned long long>(Mx) & 15) == 0);
        if (!vec1 || nrows <= 0 || ncols <= 0 || nrows > 8192) { gemv_n(Mx, l, nrows, ncols, wv, 1.0, 0.0, nullptr, out); return; }
        pbegin();
        if (nrows <= 3072) hipLaunchKernelGGL((k_gemv_n1<true, 512, true>), dim3((nrows + 15) / 16), dim3(512), 0, st, Mx, l, nrows, ncols, wv, 1.0, 0.0, (const double *)nullptr, out, (const int *)nullptr, (double *)nullptr);
        else hipLaunchKernelGGL((k_gemv_n1<true, NT, true>), dim3((nrows + 15) / 16), dim3(NT), 0, st, Mx, l, nrows, ncols, wv, 1.0, 0.0, (const double *)nullptr, out, (const int *)nullptr, (double *)nullptr);
        pend(0, 8.0 * nrows * (double)live + 8.0 * nrows + 8.0 * ncols);
        chk("gemv_n_live");
    }
    void ger(double *Mx, long long l, int nrows, int ncols, const double *t, const double *v, int ci, double cs) {
        pbegin();
        if (ncols > 0 && nrows > 0) {
            const bool vec = ((l & 1) == 0) && (((reinterpret_cast<unsigned long long>(Mx) | reinterpret_cast<unsigned long long>(t)) & 15) == 0);
            if (vec)
                hipLaunchKernelGGL(k_ger_v, dim3((nrows + 2 * NT - 1) / (2 * NT), (ncols + GER_COLS - 1) / GER_COLS), dim3(NT), 0, st, Mx, l,
                                   nrows, ncols, t, v, scal, ci, cs);
            else
                hipLaunchKernelGGL(k_ger, dim3((nrows + NT - 1) / NT, ncols), dim3(NT), 0, st, Mx, l, nrows, ncols, t, v, scal, ci, cs);
        }
        pend(2, 16.0 * nrows * (double)ncols);
    }
    void dot(const double *a, const double *b, int n, int slot) {
        hipLaunchKernelGGL(k_dot, dim3(1), dim3(NT), 0, st, a, b, n, scal, slot);
    }
    void copy(const double *s, double *d, int n) { if (n > 0) hipLaunchKernelGGL(k_copy, g1(n), dim3(NT), 0, st, s, d, n); }
    void fill(double *d, int n, double v) { if (n > 0) hipLaunchKernelGGL(k_fill, g1(n), dim3(NT), 0, st, d, n, v); }
    // full products. A dense copy (column-major) is used when the matrix is dense enough that
    // 8 B/entry coalesced GEMV beats 12 B/entry gather SpMV (BASELINE's dense configuration)
    void A_times(const double *in, double *out) {   // out[nC] = A in
        if (nC <= 0) return;
        if (M.denseA) gemv_n(M.denseA, nC, nC, nV, in, 1.0, 0.0, nullptr, out);
        else { pbegin(); (void)rsqp_launch_spmv(M.blk_r, M.nblk_r, M.Arp, M.Aci, M.Arv, in, out, 1, 0, 0, 0, 0, st); pend(5, spmv_bytes(nC, nV)); }
    }
    // algorithmic bytes of one sparse product with A (12 nnz + 4 (majors + 1) + 8 majors + 8 minors; nnz from the set-up)
    double nnzA_ = 0.0;
    double spmv_bytes(int nmajor, int nminor) const { return 12.0 * nnzA_ + 4.0 * (nmajor + 1) + 8.0 * nmajor + 8.0 * nminor; }
    void AT_times(const double *in, double *out) {  // out[nV] = A' in
        if (nC <= 0) fill(out, nV, 0.0);
        // (every vector this is called with lives on the ACTIVE constraints, but reading only those rows through the row-major copy
        //  with k_gemv_n1<.., SKIP0> measured 17.5 us against 8.5 us for the full transposed product at 2048 x 4096: the skipped
        //  loads leave too few in flight; the matrices of that size sit in the 256 MB memory-side cache anyway)
        else if (M.denseA) gemv_t(M.denseA, nC, nC, nV, in, out);
        else { pbegin(); (void)rsqp_launch_spmv(M.blk_c, M.nblk_c, M.Ajc, M.Air, M.Aval, in, out, 1, 0, 0, 0, 0, st); pend(5, spmv_bytes(nV, nC)); }
    }
    // A in and (H + hreg I) in of the SAME vector in one launch when both matrices have dense copies (A through its
    // row-major copy: both are transposed products then); otherwise one after the other
    void AH_times(const double *in, double *outA, double *outH) {
        if (nC > 0 && M.denseAT && M.haveH && M.denseH && M.hreg == 0.0) {
            gemv_t_pair(gt_task(M.denseAT, nV, nV, nC, in, outA), gt_task(M.denseH, nV, nV, nV, in, outH));
            return;
        }
        A_times(in, outA);
        H_times(in, outH);
    }
    // out = (H + hreg I) in [+ add]
    void H_times(const double *in, double *out, const double *add = nullptr) {
        if (M.haveH && M.denseH && M.hreg == 0.0) {       // the sum rides in the product's epilogue
            GtTask t = gt_task(M.denseH, nV, nV, nV, in, out);
            t.addv = add;
            gemv_t_task(t);
            return;
        }
        if (!M.haveH) fill(out, nV, 0.0);
        else if (M.denseH) gemv_t(M.denseH, nV, nV, nV, in, out);
        else (void)rsqp_launch_spmv(M.blk_h, M.nblk_h, M.Hjc, M.Hir, M.Hval, in, out, 1, 0, 0, 0, 0, st);
        if (M.hreg != 0.0) hipLaunchKernelGGL(k_axpy, g1(nV), dim3(NT), 0, st, nV, M.hreg, in, out);
        if (add) hipLaunchKernelGGL(k_axpby, g1(nV), dim3(NT), 0, st, nV, 1.0, out, 1.0, add, out);
    }
    int read_scal(int s0, int n, double *out) {
        LCHK(hipMemcpyAsync(h_pinned, scal + s0, sizeof(double) * n, hipMemcpyDeviceToHost, st));
        LCHK(hipStreamSynchronize(st));
        for (int i = 0; i < n; i++) out[i] = h_pinned[i];
        return RET_OK;
    }
    void row_of_A(int r, double *a, bool all) {
        if (M.denseAT) hipLaunchKernelGGL(k_row_of_A_dense, g1(nV), dim3(NT), 0, st, M.denseAT, nV, r, Sb, all ? 1 : 0, a);
        else hipLaunchKernelGGL(k_row_of_A_fused, dim3(1), dim3(NT), 0, st, M.Arp, M.Aci, M.Arv, r, Sb, all ? 1 : 0, nV, a);
    }
    double *Zc(int c) { return Z + c * ld; }
    double *Yc(int c) { return Y + c * ld; }

    // ---- working-set operations -------------------------------------------------------
    // reflection of Z that puts the direction Z w (w in wz1, length nZ) into the last column;
    // Wz follows; the last column is then taken out of the null space. scal[0..3] = house.
    // DEFERRED form (defer = true, the incoming constraint of a homotopy step): only the last column of Z is reflected now
    // (it becomes the new Y column); the update of the other columns and the shrinking of Wz wait for the step direction,
    // whose first products with Z and Wz apply them on the way (k_ger_gemv_t, k_wz_shrink_gemv) -- flush_pending() applies
    // them on their own if anything else wants Z or Wz first.
    // INVARIANTS of the deferred updates and the carried step direction:
    //  * pendZ / pendW (added constraint) and pendY / pendM (removed constraint) are only ever set by the LAST operation of a
    //    working-set change and consumed by the step direction that follows; change_active_set() starts with flush_pending(), the
    //    homotopy ends with it, a solve starts with all four off: a matrix is never read while its update is pending, except by
    //    the fused kernel that applies it;
    //  * c_wY / c_xY / c_wZ are the wY, xY, wZ of the LAST step direction (carry_valid); a change keeps them alive only if it is one
    //    plain added (carry_pending) or removed (carry_ready) constraint -- no exchange, no flip, no bound -- and at most
    //    CARRY_REFRESH steps in a row; everything else recomputes them from the matrices;
    //  * the operands a deferred update needs later live in their own buffers (pz_*, pw_*, py_*, pm_s, c_xi, scal[40..44]): the work
    //    vectors they were computed in are reused by the next products.
    struct { bool on = false; int ncols = 0; } pendZ;
    struct { bool on = false; int nZold = 0; } pendW;
    struct { bool on = false; int n = 0; } pendY, pendM;      // reflections of Y (n columns) / Minv (n x n) behind a removed constraint
    static constexpr int S_KEEP_BETA2 = 44;
    bool fuse_passes = getenv("RSQP_LARGE_NO_FUSE") == nullptr;
    int fuse_wz_min = getenv("RSQP_LARGE_FUSE_WZ_MIN") ? atoi(getenv("RSQP_LARGE_FUSE_WZ_MIN")) : 3072;   // (tests force 0: both fused kernels on small problems)
    static constexpr int S_KEEP_BETA = 40, S_KEEP_THETA = 41;
    // Wz symmetric: upper triangle only from wz_sym_min variables on (k_symw_tile: half the bytes of every update and product; below
    // that size the matrices are cache-resident and the single-launch kernels win). Decided per set-up (setup_aux), kept by hot starts.
    bool wz_sym_enabled = getenv("RSQP_LARGE_NO_WZ_SYM") == nullptr;
    int wz_sym_min = getenv("RSQP_LARGE_WZ_SYM_MIN") ? atoi(getenv("RSQP_LARGE_WZ_SYM_MIN")) : 4096;   // (tests force 0)
    bool wz_sym = false;
    double *wz_part = nullptr;
    // out = alpha Wz[0..n)^2 w, upper storage; upd: 0 plain, 2 the deferred shrinking applied on the way (q), n = the block it leaves
    void wz_sym_times(int n, const double *wv, double alpha, double *out, int upd, const SymW &q) {
        if (n <= 0) return;
        const int nt = (n + SYT - 1) / SYT;
        double *P1 = wz_part, *P2 = wz_part + (size_t)nt * n;
        pbegin();
        if (upd == 2) hipLaunchKernelGGL((k_symw_tile<2, true>), dim3(sym_tiles(n)), dim3(256), 0, st, Wz, ld, n, q, wv, P1, P2);
        else hipLaunchKernelGGL((k_symw_tile<0, true>), dim3(sym_tiles(n)), dim3(256), 0, st, Wz, ld, n, q, wv, P1, P2);
        hipLaunchKernelGGL(k_sym_reduce_a, g1(n), dim3(NT), 0, st, n, nt, P1, P2, alpha, out);
        pend(upd == 2 ? 7 : 0, (upd == 2 ? 8.0 : 4.0) * n * (double)n);
        chk("wz_sym_times");
    }
    void wz_times(const double *wv, double alpha, double *out) {      // out = alpha Wz w (nZ x nZ), either storage
        if (wz_sym) { SymW q{}; wz_sym_times(nZ, wv, alpha, out, 0, q); }
        else gemv_n(Wz, ld, nZ, nZ, wv, alpha, 0.0, nullptr, out);
    }
    bool can_defer() const {
        return fuse_passes && wz_enabled && (ld & 1) == 0 && nZ > 1 && pz_t &&
               (((reinterpret_cast<unsigned long long>(Z) | reinterpret_cast<unsigned long long>(Wz)) & 15) == 0);
    }
    void flush_pending_ZW() {
        if (pendZ.on) { pendZ.on = false; ger(Z, ld, nV, pendZ.ncols, pz_t, pz_v, S_KEEP_BETA, -1.0); }
        if (pendW.on) { pendW.on = false; wz_shrink_now(pendW.nZold, pw_s, pz_v, pw_col, S_KEEP_BETA, S_KEEP_THETA); }
    }
    void flush_pending() {
        flush_pending_ZW();
        if (pendY.on) { pendY.on = false; ger(Y, ld, nV, pendY.n, py_t, py_v, S_KEEP_BETA2, -1.0); }
        if (pendM.on) { pendM.on = false; ger(Minv, ldm, pendM.n, pendM.n, py_v, pm_s, S_KEEP_BETA2, -1.0); }
    }
    void wz_shrink_now(int nZo, const double *s_, const double *v_, const double *col_, int sb, int stheta) {
        if (wz_sym) {
            if (nZo > 1) {
                SymW q{nullptr, s_, v_, col_, scal, sb, stheta, nZo - 1};
                pbegin();
                hipLaunchKernelGGL((k_symw_tile<2, false>), dim3(sym_tiles(nZo - 1)), dim3(256), 0, st, Wz, ld, nZo - 1, q, (const double *)nullptr,
                                   (double *)nullptr, (double *)nullptr);
                pend(3, 8.0 * (double)nZo * nZo);
            }
            return;
        }
        pbegin();
        if (nZo > 1) {
            if ((ld & 1) == 0)
                hipLaunchKernelGGL(k_wz_shrink_v, dim3((nZo - 1 + 2 * NT - 1) / (2 * NT), (nZo - 1 + WZ_COLS - 1) / WZ_COLS), dim3(NT), 0, st, Wz,
                                   ld, nZo, s_, v_, col_, scal, sb, stheta);
            else
                hipLaunchKernelGGL(k_wz_shrink, dim3((nZo - 1 + NT - 1) / NT, nZo - 1), dim3(NT), 0, st, Wz, ld, nZo, s_, v_, col_,
                                   scal, sb, stheta);
        }
        pend(3, 16.0 * (double)nZo * nZo);
    }
    // ycol != nullptr: the reflected last column of Z is also copied there (the new column of Y) -- returns true if that was done
    // (the deferred form does it in the launch that keeps the reflector), false: the caller copies
    bool z_reflect_and_shrink(bool defer = false, double *ycol = nullptr) {
        flush_pending();
        if (!house_done) hipLaunchKernelGGL(k_house, dim3(1), dim3(NT), 0, st, wz1, nZ, wz2, scal, 0);   // v -> wz2 (else: done by the products' kernel)
        house_done = false;
        const bool dz = defer && can_defer();
        double *s_ = dz ? pw_s : wz3, *col_ = dz ? pw_col : w6;
        // t = Z v and s = Wz v: one launch at launch-bound sizes
        const bool paired = wz_enabled && !wz_sym && gemv_n1_pair(Z, ld, nV, nZ, wz2, dz ? pz_t : w5, Wz, ld, nZ, nZ, wz2, s_);
        if (!paired) gemv_n(Z, ld, nV, nZ, wz2, 1.0, 0.0, nullptr, dz ? pz_t : w5);     // t = Z v
        bool copied = false;
        if (dz) {
            if (ycol) {
                hipLaunchKernelGGL(k_keep_reflect_lastcol, g1(std::max(nZ, nV)), dim3(NT), 0, st, wz2, nZ, pz_v, scal, 1, S_KEEP_BETA, nV, pz_t,
                                   Zc(nZ - 1), ycol);
                copied = true;
            } else {
                hipLaunchKernelGGL(k_keep_reflector, g1(nZ), dim3(NT), 0, st, wz2, nZ, pz_v, scal, 1, S_KEEP_BETA);
                ger(Zc(nZ - 1), ld, nV, 1, pz_t, pz_v + (nZ - 1), S_KEEP_BETA, -1.0);   // the column that moves to Y: now
            }
            pendZ.on = true; pendZ.ncols = nZ - 1;
        } else ger(Z, ld, nV, nZ, w5, wz2, 1, -1.0);                                     // Z -= beta t v'
        if (!wz_enabled) return copied;
        if (!paired) wz_times(wz2, 1.0, s_);                                             // s = Wz v
        hipLaunchKernelGGL(k_wz_lastcol, g1(nZ), dim3(NT), 0, st, Wz, ld, nZ, s_, wz2, scal, 1, dz ? S_KEEP_THETA : 4, col_);   // theta = v's
        if (dz) { pendW.on = true; pendW.nZold = nZ; }
        else wz_shrink_now(nZ, wz3, wz2, w6, 1, 4);
        return copied;
    }
    // the step direction's  out = Z'x + addv  and  out = alpha Wz w  with the deferred updates applied on the way
    void gemv_t_Z_pending(const double *xv, const double *addv, double *out) {
        const bool ok = pendZ.on && pendZ.ncols == nZ && nZ > 0 &&
                        (((reinterpret_cast<unsigned long long>(xv) | reinterpret_cast<unsigned long long>(pz_t)) & 15) == 0);
        if (!ok) {
            flush_pending_ZW();
            GtTask t = gt_task(Z, ld, nV, nZ, xv, out);
            t.addv = addv;
            gemv_t_task(t);
            return;
        }
        pendZ.on = false;
        pbegin();
        hipLaunchKernelGGL(k_ger_gemv_t, dim3((nZ + GT_COLS - 1) / GT_COLS), dim3(NT), 0, st, Z, ld, nV, nZ, pz_t, pz_v, scal, S_KEEP_BETA, -1.0, xv,
                           addv, out);
        pend(6, 16.0 * nV * (double)nZ);
        chk("ger_gemv_t");
    }
    void gemv_n_Wz_pending(const double *wv, double alpha, double *out) {
        // (below ~3000 columns the fused pass is launch-bound -- partial sums + reduction, 35 us at nZ = 2300 -- and the separate
        //  kernels, 19 + 9 us, are faster)
        if (wz_sym && pendW.on && pendW.nZold - 1 == nZ && nZ > 0) {      // (upper storage: the fused pass at every size)
            pendW.on = false;
            SymW q{nullptr, pw_s, pz_v, pw_col, scal, S_KEEP_BETA, S_KEEP_THETA, nZ};
            wz_sym_times(nZ, wv, alpha, out, 2, q);
            return;
        }
        if (!(pendW.on && pendW.nZold - 1 == nZ && nZ >= fuse_wz_min && nZ > 0) || wz_sym) {
            if (pendW.on) { pendW.on = false; wz_shrink_now(pendW.nZold, pw_s, pz_v, pw_col, S_KEEP_BETA, S_KEEP_THETA); }
            wz_times(wv, alpha, out);
            return;
        }
        pendW.on = false;
        pbegin();
        const int rb = (nZ + 127) / 128;
        int nch = std::max(1, std::min(std::min((gemv_wgs + rb - 1) / rb, 32), (nZ + 15) / 16));
        while ((long long)nch * nZ > part_cap) nch = (nch + 1) / 2;
        const int chunk = (nZ + nch - 1) / nch;
        nch = (nZ + chunk - 1) / chunk;
        hipLaunchKernelGGL(k_wz_shrink_gemv, dim3(rb, nch), dim3(NT), 0, st, Wz, ld, nZ + 1, pw_s, pz_v, pw_col, scal, S_KEEP_BETA, S_KEEP_THETA, chunk,
                           wv, alpha, part, out);
        if (nch > 1) hipLaunchKernelGGL(k_gemv_n_reduce, dim3((nZ + 63) / 64), dim3(NT), 0, st, part, nZ, nch, alpha, 0.0, (const double *)nullptr, out);
        pend(7, 16.0 * nZ * (double)nZ);
        chk("wz_shrink_gemv");
    }

    // append the Y column `ycol` (already stored at Y[:, nAC]) for constraint r whose products
    // with the old Y are in a1 (wY) and with the new column in scal[eta_slot]
    // The step direction's wY = Minv bA and xY = Y wY are CARRIED over an incoming constraint (k_carry_add: one
    // O(n) kernel instead of one pass over Minv and one over Y) and recomputed exactly after any other change, at the first
    // step, and every CARRY_REFRESH carried steps (RSQP_LARGE_NO_CARRY=1: always exactly).
    static constexpr int S_KEEP_ETA = 43, S_KEEP_WLAST = 42, CARRY_REFRESH = 8;
    bool carry_enabled = getenv("RSQP_LARGE_NO_CARRY") == nullptr;
    bool carry_null_enabled = getenv("RSQP_LARGE_NO_CARRY_NULL") == nullptr;   // (the null-space part as well)
    bool carry_valid = false;        // c_wY / c_xY are those of the last step direction, nothing but a homotopy step since
    bool carry_pending = false;      // ... and the change behind it was a plain added constraint (border kept in c_xi, S_KEEP_ETA)
    bool carry_ready = false;        // ... or a plain removed constraint: c_wY / c_xY already transformed (remove_constraint_tq)
    bool plain_removal = false;      // the change is ONE removed constraint (not the partner of an exchange)
    int carried = 0;
    long long stat_carried = 0, stat_carried_null = 0;
    bool plain_add = false;          // the change is ONE added constraint (no exchange partner removed first)
    double last_tau = 0.0;
    // (yidx >= 0: y[yidx] = yval set by the same launch -- the multiplier of the incoming constraint)
    void minv_append(int eta_slot, bool eta_from_house, int r, int side, bool keep_for_carry = false, int yidx = -1, double yval = 0.0) {
        // new row nAC: -(wY' Minv)/eta ; new column nAC: 0 ; corner 1/eta ; working set: constraint r at position nAC
        gemv_t(Minv, ldm, nAC, nAC, a1, a2);  // a2[j] = sum_i wY[i] Minv[i][j]
        hipLaunchKernelGGL(k_minv_border_keep, g1(nAC + 1), dim3(NT), 0, st, Minv, ldm, nAC, a2, scal, eta_slot, eta_from_house ? 1 : 0,
                           AC, posAC, Sc, r, side, keep_for_carry ? c_xi : (double *)nullptr, S_KEEP_ETA, y, yidx, yval);
    }

    // (yidx / yval: see minv_append)
    int add_constraint(int r, int side, bool skipZ, bool defer = false, int yidx = -1, double yval = 0.0) {
        // a (free part) in w1; wZ in wz1; wY in a1 were computed by the caller (li_test_constraint)
        if (!skipZ) {
            // new Y column = last column of the reflected Z; eta = a'y_new = image sign * alpha
            if (!z_reflect_and_shrink(defer, Yc(nAC))) copy(Zc(nZ - 1), Yc(nAC), nV);
        } else {
            // exchange / flip: the row is orthogonal to all null-space columns but the last
            flush_pending();
            house_done = false;
            copy(Zc(nZ - 1), Yc(nAC), nV);
            dot(w1, Zc(nZ - 1), nV, 5);
        }
        nZ--;
        const bool keep = defer && !skipZ && plain_add && carry_enabled && carry_valid && carried < CARRY_REFRESH;
        minv_append(5, !skipZ, r, side, keep, yidx, yval);
        carry_pending = keep;
        hAC[nAC] = r; hSc[r] = side;
        nAC++;
        return RET_OK;
    }

    // products of constraint row r with the bases: w1 = a_FR, wz1 = Z'a, a1 = Y'a; scal[6]=|a|^2, scal[7]=|wZ|^2
    void constraint_products(int r) {
        row_of_A(r, w1, false);
        if (M.sparse_rows) {
            const int nb = (nZ + 255) / 256 + (nAC + 255) / 256;
            if (nb > 0)
                hipLaunchKernelGGL(k_row_times_bases, dim3(nb), dim3(256), 0, st, M.Arp, M.Aci, M.Arv, r, Z, nZ, wz1, Y, nAC, a1, ld);
        } else {
            gemv_t_pair(gt_task(Z, ld, nV, nZ, w1, wz1), gt_task(Y, ld, nV, nAC, w1, a1));
        }
        hipLaunchKernelGGL(k_norms_publish, dim3(1), dim3(NT), 0, st, w1, nV, wz1, nZ, scal, 6, 7, d_ctl, next_seq(), wz2);
        house_done = nZ > 0;
    }
    void bound_products(int v) {
        hipLaunchKernelGGL(k_bound_products, dim3(1), dim3(NT), 0, st, Z, Y, ld, v, nZ, nAC, wz1, a1, scal, 6, 7, d_ctl, next_seq(), wz2);
        house_done = nZ > 0;
    }

    // second stage shared by add_bound and (mirrored) remove_bound: reflection on [Y, extra]
    // add_bound(v): qY (row v of Y) in a1, q* = zs[v] where zs = Z's last column after stage 1
    int add_bound(int v, int side, bool skipZ) {
        if (!skipZ) z_reflect_and_shrink();          // wz1 = row v of Z (from bound_products)
        else house_done = false;
        double *zs = Zc(nZ - 1);
        nZ--;
        // q~ = [qY ; q*]; |q~| = 1. vt = q~ with last += sgn(q*); beta~ = 1/(1+|q*|)
        hipLaunchKernelGGL(k_house_unit, dim3(1), dim3(1), 0, st, a1, nAC, scal, zs, v);  // a1[nAC] = q* + sgn; scal[8]=beta~, scal[9]=gamma, scal[10]=vlast
        // t = Y vY + zs * vlast
        gemv_n(Y, ld, nV, nAC, a1, 1.0, 0.0, nullptr, w5);
        hipLaunchKernelGGL(k_axpy_s, g1(nV), dim3(NT), 0, st, nV, scal, 10, zs, w5);
        ger(Y, ld, nV, nAC, w5, a1, 8, -1.0);        // Y -= beta~ t vY'
        // Minv += gamma vY (vY' Minv)
        gemv_t(Minv, ldm, nAC, nAC, a1, a2);
        ger(Minv, ldm, nAC, nAC, a1, a2, 9, 1.0);
        // clean row v
        hipLaunchKernelGGL(k_clean_bound, g1(std::max(std::max(nAC, nZ), 1)), dim3(NT), 0, st, Y, Z, ld, v, nAC, nZ, Sb, side);
        hSb[v] = side;
        nFR--;
        return RET_OK;
    }

    // grow Wz by the new null-space column Z[:, nZ]; returns 1 if positive definite (then nZ++)
    int wz_grow(int *pd) {
        double *z = Zc(nZ);
        H_times(z, w2);
        gemv_t(Z, ld, nV, nZ, w2, wz1);           // k = Z'Hz
        wz_times(wz1, 1.0, wz2);                              // u = Wz k
        // kappa = z'Hz, ku = k'u, scal[13] = rho2, scal[14] = threshold
        hipLaunchKernelGGL(k_rho2, dim3(1), dim3(NT), 0, st, scal, d_ctl, z, w2, nV, wz1, wz2, nZ, next_seq());
        // the growth is launched before the host has the verdict: the kernel tests rho2 > threshold itself (scal[13], [14])
        pbegin();
        if (wz_sym) {
            SymW q{wz2, nullptr, nullptr, nullptr, scal, 13, 0, 0};
            if (nZ > 0) hipLaunchKernelGGL((k_symw_tile<3, false>), dim3(sym_tiles(nZ)), dim3(256), 0, st, Wz, ld, nZ, q, (const double *)nullptr,
                                           (double *)nullptr, (double *)nullptr);
            hipLaunchKernelGGL(k_wz_grow_col_sym, g1(nZ + 1), dim3(NT), 0, st, Wz, ld, nZ, wz2, scal, 13);
        } else if ((ld & 1) == 0)
            hipLaunchKernelGGL(k_wz_grow_v, dim3((nZ + 1 + 2 * NT - 1) / (2 * NT), (nZ + 1 + WZ_COLS - 1) / WZ_COLS), dim3(NT), 0, st, Wz, ld,
                               nZ, wz2, scal, 13);
        else
            hipLaunchKernelGGL(k_wz_grow, dim3((nZ + 1 + NT - 1) / NT, nZ + 1), dim3(NT), 0, st, Wz, ld, nZ, wz2, scal, 13);
        pend(4, 16.0 * (double)nZ * nZ);
        if (wait_ctl() != RET_OK) return wait_failed();
        *pd = h_ctl[4] > h_ctl[5];
        if (*pd) nZ++;
        return RET_OK;
    }

    // TQ part of removing the constraint at position k: Y loses a column, it lands in Z[:, nZ]
    void remove_constraint_tq(int k) {
        const int r = hAC[k];
        hipLaunchKernelGGL(k_house, dim3(1), dim3(NT), 0, st, Minv + k * ldm, nAC, a2, scal, 0);  // u = Minv[:, k]; v -> a2
        // a plain removal whose step direction will be carried (below) does not need Y and Minv before that step direction's LAST
        // two products (Y'r, Minv'(Y'r)): their reflections wait and ride on those (k_ger_gemv_t) -- only the last column of each is
        // reflected now (it moves to Z / into the slot of the removed constraint)
        const bool will_carry = plain_removal && carry_enabled && carry_valid && carried < CARRY_REFRESH && nAC > 1;
        const bool dfr = will_carry && fuse_passes && (ld & 1) == 0 && (ldm & 1) == 0 &&
                         (((reinterpret_cast<unsigned long long>(Y) | reinterpret_cast<unsigned long long>(Minv)) & 15) == 0);
        gemv_n(Y, ld, nV, nAC, a2, 1.0, 0.0, nullptr, dfr ? py_t : w5);  // t = Y v
        gemv_t(Minv, ldm, nAC, nAC, a2, a3);                             // s' = v' Minv
        if (dfr && tail_fused) {
            // (dfr implies will_carry) everything that follows in one launch
            hipLaunchKernelGGL(k_remove_tail, g1(std::max(nV, nAC)), dim3(NT), 0, st, nAC, nV, k, 1.0 - last_tau, a2, a3, py_t, py_v, pm_s, scal, 1,
                               S_KEEP_BETA2, S_KEEP_WLAST, Yc(nAC - 1), Zc(nZ), Minv + (long long)(nAC - 1) * ldm,
                               k != nAC - 1 ? Minv + k * ldm : (double *)nullptr, c_wY, c_wY2, c_xY, AC, posAC, Sc, r, y, nV + r);
            std::swap(c_wY, c_wY2);
            pendY.on = true; pendY.n = nAC - 1;
            pendM.on = true; pendM.n = nAC - 1;
            carry_ready = true;
            if (k != nAC - 1) hAC[k] = hAC[nAC - 1];
            hSc[r] = 0;
            nAC--;
            return;
        }
        if (dfr) {
            hipLaunchKernelGGL(k_keep_removal, g1(nAC), dim3(NT), 0, st, nAC, a2, a3, k, py_v, pm_s, scal, 1, S_KEEP_BETA2);
            ger(Yc(nAC - 1), ld, nV, 1, py_t, a2 + (nAC - 1), 1, -1.0);
            ger(Minv + (long long)(nAC - 1) * ldm, ldm, nAC, 1, a2, a3 + (nAC - 1), 1, -1.0);
            pendY.on = true; pendY.n = nAC - 1;
            pendM.on = true; pendM.n = nAC - 1;
        } else {
            ger(Y, ld, nV, nAC, w5, a2, 1, -1.0);                        // Y -= beta t v'
            ger(Minv, ldm, nAC, nAC, a2, a3, 1, -1.0);                   // Minv -= beta v s'
        }
        copy(Yc(nAC - 1), Zc(nZ), nV);                                   // new null-space column
        if (will_carry) {
            const double om = 1.0 - last_tau;
            hipLaunchKernelGGL(k_carry_remove_wY, dim3(1), dim3(NT), 0, st, nAC - 1, om, a2, c_wY, scal, 1, S_KEEP_WLAST);
            hipLaunchKernelGGL(k_carry_remove_xY, g1(nV), dim3(NT), 0, st, nV, om, Zc(nZ), scal, S_KEEP_WLAST, c_xY);
            carry_ready = true;
        }
        // delete row (nAC-1) [implicit] and column k: move the last column into k
        if (k != nAC - 1) {
            copy(Minv + (long long)(nAC - 1) * ldm, Minv + k * ldm, nAC);
            hAC[k] = hAC[nAC - 1];
        }
        hipLaunchKernelGGL(k_ws_remove_c, dim3(1), dim3(1), 0, st, AC, posAC, Sc, k, nAC - 1, r, y, nV + r);   // also y[nV + r] = 0
        hSc[r] = 0;
        nAC--;
    }
    int position_of(int r) const {
        for (int j = 0; j < nAC; j++) if (hAC[j] == r) return j;
        return -1;
    }

    // TQ part of freeing variable v: null space gains the column Z[:, nZ]
    void remove_bound_tq(int v) {
        hipLaunchKernelGGL(k_free_bound_ws, dim3(1), dim3(1), 0, st, Sb, v, y);   // Sb[v] = 0, y[v] = 0
        hSb[v] = 0;
        nFR++;
        double *znew = Zc(nZ);
        fill(znew, nV, 0.0);
        if (nAC == 0) {
            hipLaunchKernelGGL(k_set1, dim3(1), dim3(1), 0, st, znew, v, 1.0);
            return;
        }
        fill(a4, nAC, 0.0);
        hipLaunchKernelGGL(k_col_of_A_active, dim3(4), dim3(NT), 0, st, M.Ajc, M.Air, M.Aval, v, posAC, a4);  // a_v
        gemv_n(Minv, ldm, nAC, nAC, a4, 1.0, 0.0, nullptr, a1);          // c = Minv a_v
        // c~ = [-c ; 1], nu = sqrt(1+|c|^2); vt = c~ with last += nu; beta~ = 1/(nu(nu+1))
        hipLaunchKernelGGL(k_house_free, dim3(1), dim3(NT), 0, st, a1, nAC, scal);   // a1 := vY = -c ; scal[8]=beta~, scal[10]=vlast
        // t = Y vY + e_v vlast
        gemv_n(Y, ld, nV, nAC, a1, 1.0, 0.0, nullptr, w5);
        hipLaunchKernelGGL(k_add_scal_at, dim3(1), dim3(1), 0, st, w5, v, scal, 10);
        // p = A_AC t (before Y changes it does not matter: t is already formed)
        A_times(w5, c3);
        hipLaunchKernelGGL(k_gather_active, g1(nAC), dim3(NT), 0, st, c3, AC, nAC, a2);   // p
        // new column: e_v - beta~ t vlast
        hipLaunchKernelGGL(k_newcol_free, g1(nV), dim3(NT), 0, st, nV, v, w5, scal, znew);
        ger(Y, ld, nV, nAC, w5, a1, 8, -1.0);                            // Y -= beta~ t vY'
        // Sherman-Morrison: Minv += beta~/(1 - beta~ vY'q1) q1 q2',  q1 = Minv p, q2' = vY' Minv
        gemv_n(Minv, ldm, nAC, nAC, a2, 1.0, 0.0, nullptr, a3);          // q1
        gemv_t(Minv, ldm, nAC, nAC, a1, a4);                             // q2
        dot(a1, a3, nAC, 15);
        hipLaunchKernelGGL(k_sm_coef, dim3(1), dim3(1), 0, st, scal);    // scal[16] = beta~/(1 - beta~*scal[15])
        ger(Minv, ldm, nAC, nAC, a3, a4, 16, 1.0);
    }

    // removal with definiteness guard; returns RET_OK / RET_UNBOUNDED
    int remove_with_guard(bool is_bound, int idx, const double *lbN_h, const double *ubN_h) {
        (void)lbN_h; (void)ubN_h;
        int pd = 0;
        if (is_bound) {
            const int old = hSb[idx];
            remove_bound_tq(idx);   // (zeroes y[idx] as well)
            if (wz_grow(&pd) != RET_OK) return RET_SETUP_FAILED;
            if (pd) return RET_OK;
            flush_pending();          // (the flip below reads Y / Minv: deferred reflections first)
            // flip: put the variable back on the opposite side (or the same if that one is infinite)
            double b[2];
            LCHK(hipMemcpyAsync(h_pinned, lbN + idx, 8, hipMemcpyDeviceToHost, st));
            LCHK(hipMemcpyAsync(h_pinned + 1, ubN + idx, 8, hipMemcpyDeviceToHost, st));
            LCHK(hipStreamSynchronize(st));
            b[0] = h_pinned[0]; b[1] = h_pinned[1];
            const bool cant = (old == -1 && b[1] >= RSQP_INFTY) || (old == 1 && b[0] <= -RSQP_INFTY);
            nZ++;  // the candidate column is still Z[:, nZ]: treat it as the last null-space column
            bound_products(idx);
            add_bound(idx, cant ? old : -old, true);
            if (cant) return RET_UNBOUNDED;
            hipLaunchKernelGGL(k_copy1, dim3(1), dim3(1), 0, st, old == -1 ? ub : lb, idx, x, idx);
            nflips++;
            return RET_OK;
        } else {
            const int old = hSc[idx], k = position_of(idx);
            remove_constraint_tq(k);   // (zeroes y[nV + idx] as well)
            if (wz_grow(&pd) != RET_OK) return RET_SETUP_FAILED;
            if (pd) return RET_OK;
            flush_pending();          // (the flip below reads Y / Minv: deferred reflections first)
            LCHK(hipMemcpyAsync(h_pinned, lbAN + idx, 8, hipMemcpyDeviceToHost, st));
            LCHK(hipMemcpyAsync(h_pinned + 1, ubAN + idx, 8, hipMemcpyDeviceToHost, st));
            LCHK(hipStreamSynchronize(st));
            const bool cant = (old == -1 && h_pinned[1] >= RSQP_INFTY) || (old == 1 && h_pinned[0] <= -RSQP_INFTY);
            nZ++;
            constraint_products(idx);
            add_constraint(idx, cant ? old : -old, true);
            if (cant) return RET_UNBOUNDED;
            hipLaunchKernelGGL(k_copy1, dim3(1), dim3(1), 0, st, old == -1 ? ubA : lbA, idx, Ax, idx);
            nflips++;
            return RET_OK;
        }
    }

    // exchange: incoming row in w4 (all variables), its Y-products in a1. Finds the partner,
    // shifts the duals. ret: RET_OK / RET_INFEASIBLE; partner in (pkind, pidx), y_new
    int ensure_LI(int side, double *y_new, int *pkind, int *pidx) {
        if (!dual && !rsh) {      // (the range-space paths have xi = Sinv c scattered in c1 and A_AC'xi in w2 already: dual_residual / rs_residual)
            fill(c1, nC, 0.0);
            gemv_t(Minv, ldm, nAC, nAC, a1, a2);   // xiC = Minv' wY  -> xi[j] = sum_i Minv[i][j] wY[i]
            if (nAC > 0) hipLaunchKernelGGL(k_scatter_active, g1(nAC), dim3(NT), 0, st, a2, AC, nAC, c1);
            AT_times(c1, w2);
        }
        hipLaunchKernelGGL(k_xiB, g1(nV), dim3(NT), 0, st, nV, Sb, w4, w2, w3);
        const double sgn = side == 1 ? -1.0 : 1.0;
        hipLaunchKernelGGL(k_partner1, dim3(nblk_ratio), dim3(NT), 0, st, nV, nC, Sb, Sc, y, c1, w3, sgn, pt, pid);
        hipLaunchKernelGGL(k_argmin2, dim3(1), dim3(NT), 0, st, nblk_ratio, pt, pid, d_ctl, next_seq(), (double *)nullptr);
        if (wait_ctl() != RET_OK) return wait_failed();
        const double t = h_ctl[0];
        const int id = (int)h_ctl[1];
        if (id == 0x7fffffff) return RET_INFEASIBLE;
        hipLaunchKernelGGL(k_shift_duals, g1(nV + nC), dim3(NT), 0, st, nV, nC, Sb, Sc, t, sgn, c1, w3, y);
        dirty_products = true;
        *y_new = sgn * t;
        *pkind = id < nC ? 1 : 2;
        *pidx = id < nC ? id : id - nC;
        return RET_OK;
    }

    int li_decision(bool *li) {
        if (wait_ctl() != RET_OK) return wait_failed();
        const double a2 = h_ctl[2], w2n = h_ctl[3];
        *li = nZ > 0 && a2 > 0.0 && std::sqrt(w2n) > RSQP_EPS_LI * std::sqrt(a2);
        return RET_OK;
    }

    bool dual_carry_enabled = getenv("RSQP_LARGE_NO_CARRY") == nullptr;
    int dual_carry_row = -1;         // the constraint a carried plain addition brought in (position nAC - 1)
    int dual_change_active_set(int kind, int idx, int side) {
        carry_pending = carry_ready = false;
        if (kind == 1) {
            const int k = position_of(idx);
            // a plain removal: the multiplier step is carried (c_wY transformed here, with column k of Sinv as it is before the update)
            const bool will_carry = dual_carry_enabled && carry_valid && carried < CARRY_REFRESH && nAC > 1;
            dual_remove_constraint(k, will_carry);
            carry_ready = will_carry;
            return RET_OK;
        }
        if (kind == 2) { dual_remove_bound(idx); return RET_OK; }
        double ynew = 0.0;
        bool li = false, dual_exchanged = false;
        if (kind == 3) dual_constraint_products(idx); else dual_bound_products(idx);
        if (dual_li_decision(&li) != RET_OK) return RET_SETUP_FAILED;
        if (!li) {
            int pkind = 0, pidx = -1;
            dual_exchanged = true;
            if (kind == 3) row_of_A(idx, w4, true);
            else { fill(w4, nV, 0.0); hipLaunchKernelGGL(k_set1, dim3(1), dim3(1), 0, st, w4, idx, 1.0); }
            int rc = ensure_LI(side, &ynew, &pkind, &pidx);
            if (rc != RET_OK) return rc;
            if (pkind == 1) dual_remove_constraint(position_of(pidx), false); else dual_remove_bound(pidx);
            if (kind == 3) dual_constraint_products(idx); else dual_bound_products(idx);
            if (dual_li_decision(&li) != RET_OK) return RET_SETUP_FAILED;
            // (the partner's removal has made room for the incoming row by construction of the exchange: a residual that stays
            //  in the rounding band does not veto it as long as the pivot of the bordering is positive)
            if (!li && !(h_ctl[4] > 1e-14 * h_ctl[5])) return RET_SETUP_FAILED;
        }
        if (kind == 3) {
            dual_add_constraint(idx, side, nV + idx, ynew);
            // a plain addition (no exchange before it: c in a1, u in a2, 1 / s in scal[8] are those of the bordering)
            carry_pending = dual_carry_enabled && ynew == 0.0 && li && carry_valid && carried < CARRY_REFRESH && !dual_exchanged;
            dual_carry_row = idx;
        } else {
            dual_add_bound(idx, side);
            hipLaunchKernelGGL(k_set1, dim3(1), dim3(1), 0, st, y, idx, ynew);
        }
        return RET_OK;
    }

    int change_active_set(int kind, int idx, int side) {
        if (rsh) return rs_change_active_set(kind, idx, side);
        if (dual) return dual_change_active_set(kind, idx, side);
        flush_pending();
        carry_pending = carry_ready = false;       // (set again by a plain added / removed constraint below)
        if (kind == 1) {
            plain_removal = true;
            const int flips_before_change = nflips;
            const int rc = remove_with_guard(false, idx, nullptr, nullptr);
            plain_removal = false;
            if (nflips != flips_before_change) carry_ready = false;     // the constraint came back on its other side: not a plain removal
            return rc;
        }
        if (kind == 2) return remove_with_guard(true, idx, nullptr, nullptr);
        double ynew = 0.0;
        bool li = false, full = true;
        if (kind == 3) constraint_products(idx); else bound_products(idx);
        if (li_decision(&li) != RET_OK) return RET_SETUP_FAILED;
        if (!li) {
            int pkind = 0, pidx = -1;
            if (kind == 3) row_of_A(idx, w4, true);
            else { fill(w4, nV, 0.0); hipLaunchKernelGGL(k_set1, dim3(1), dim3(1), 0, st, w4, idx, 1.0); }
            int rc = ensure_LI(side, &ynew, &pkind, &pidx);
            if (rc != RET_OK) return rc;
            int pd = 0;
            if (pkind == 1) {
                remove_constraint_tq(position_of(pidx));   // (zeroes y[nV + pidx] as well)
            } else {
                remove_bound_tq(pidx);   // (zeroes y[pidx] as well)
            }
            if (wz_grow(&pd) != RET_OK) return RET_SETUP_FAILED;
            full = pd != 0;
            if (!full) nZ++;   // keep the candidate column as the last null-space column (R-less exchange)
            if (kind == 3) constraint_products(idx); else bound_products(idx);
        }
        if (kind == 3) {
            plain_add = li;
            add_constraint(idx, side, !full, true, nV + idx, ynew);      // (the last operation of the change: the step direction follows;
            plain_add = false;                                           //  sets y[nV + idx] = ynew as well)
        } else {
            add_bound(idx, side, !full);
            hipLaunchKernelGGL(k_set1, dim3(1), dim3(1), 0, st, y, idx, ynew);
        }
        return RET_OK;
    }

    // ---- range-space (dual) path ---------------------------------------------------------
    // may this solve take it? H stored as exactly one entry per column on the diagonal (host pattern check by the caller:
    // M.diagH), every d + hreg > 0 (device check here; one host round trip per set-up)
    int dual_prepare(bool *ok) {
        *ok = false;
        if (!dual_enabled || !M.diagH || !M.haveH || nC <= 0) return RET_OK;
        LCHK(hipMemsetAsync(dflag, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_dual_hinv, g1(nV), dim3(NT), 0, st, nV, M.Hval, M.hreg, hinv, dflag);
        LCHK(hipMemcpyAsync(h_pinned_i, dflag, sizeof(int), hipMemcpyDeviceToHost, st));
        LCHK(hipStreamSynchronize(st));
        *ok = h_pinned_i[0] == 0;
        return RET_OK;
    }
    // products of an incoming row with the working set: a_FR in w1, D^-1 a_FR in w5, c in a1, u = Sinv c in a2 (and scattered by
    // constraint in c1), A_AC'u in w2; |a_FR|^2, |r|^2, the pivot s published (scal[5..8])
    // Two stages: the pivot s first -- s > 1e-6 a_FR'D^-1 a_FR is independence beyond doubt (|P a| > 1e-3 |a| in the D^-1 metric);
    // only below that the residual r of the representation by the active rows is formed (A_AC'u: one more product with A) and
    // published in a second round (dual_residual: also what an exchange needs, c1 = u by constraint and w2 = A_AC'u)
    void dual_products_tail() {
        A_times(w5, c3);
        if (nAC > 0) hipLaunchKernelGGL(k_gather_active, g1(nAC), dim3(NT), 0, st, c3, AC, nAC, a1);
        dual_sinv_times(a1, a2);
        hipLaunchKernelGGL(k_dual_li_publish, dim3(1), dim3(1024), 0, st, nV, (const int *)nullptr, w1, w5, (const double *)nullptr, nAC, a1, a2,
                           scal, d_ctl, next_seq());
        dual_have_residual = false;
    }
    bool dual_have_residual = false;
    void dual_residual() {
        fill(c1, nC, 0.0);
        if (nAC > 0) hipLaunchKernelGGL(k_scatter_active, g1(nAC), dim3(NT), 0, st, a2, AC, nAC, c1);
        AT_times(c1, w2);
        hipLaunchKernelGGL(k_dual_li_publish, dim3(1), dim3(1024), 0, st, nV, Sb, w1, w5, w2, nAC, a1, a2, scal, d_ctl, next_seq());
        dual_have_residual = true;
    }
    void dual_constraint_products(int r) {
        if (!M.denseAT) hipLaunchKernelGGL(k_row_of_A_fused, dim3(1), dim3(NT), 0, st, M.Arp, M.Aci, M.Arv, r, Sb, 0, nV, w1, hinv, w5);
        else {
            row_of_A(r, w1, false);
            hipLaunchKernelGGL(k_dual_scale, g1(nV), dim3(NT), 0, st, nV, hinv, w1, w5);
        }
        dual_products_tail();
    }
    void dual_bound_products(int v) {
        hipLaunchKernelGGL(k_dual_unit, g1(nV), dim3(NT), 0, st, nV, v, Sb, hinv, w1, w5);
        dual_products_tail();
    }
    int dual_li_decision(bool *li) {
        if (wait_ctl() != RET_OK) return wait_failed();
        double a2n = h_ctl[2], sp = h_ctl[4];
        const double ad = h_ctl[5];
        if (nFR - nAC > 0 && a2n > 0.0 && sp > 1e-6 * ad) { *li = true; return RET_OK; }
        dual_residual();
        if (wait_ctl() != RET_OK) return wait_failed();
        a2n = h_ctl[2]; sp = h_ctl[4];
        const double r2 = h_ctl[3];
        *li = nFR - nAC > 0 && a2n > 0.0 && std::sqrt(r2) > RSQP_EPS_LI * std::sqrt(a2n) && sp > 0.0;
        return RET_OK;
    }
    // Sinv <- [[Sinv + u u'/s, -u/s], [-u'/s, 1/s]]  (u in a2, 1/s in scal[8])
    void lz_push(int n, const double *v, int src) {      // c v v' with c = scal[src] joins the pending terms
        if (lz_n >= LZK) dual_flush();
        hipLaunchKernelGGL(k_lz_push, g1(std::max(n, 1)), dim3(NT), 0, st, n, v, scal, src, lz_vec, lz_stride, lz_n, lz_c);
        lz_n++;
    }
    void dual_add_constraint(int r, int side, int yidx = -1, double yval = 0.0) {
        if (dual_lazy()) {
            if (lz_n >= LZK) dual_flush();
            hipLaunchKernelGGL(k_lz_border, g1(nAC + 1), dim3(NT), 0, st, Minv, ldm, nAC, a2, scal, AC, posAC, Sc, r, side, y, yidx, yval, lz_vec, lz_stride, lz_n, lz_c);
            lz_n++;
            hAC[nAC] = r; hSc[r] = side;
            nAC++;
            nZ = nFR - nAC;
            return;
        }
        dual_flush();       // (none is pending here: the products of this row have applied it)
        const bool defer = dual_defer && nAC > 0;
        if (defer) {
            if (!dual_sym) hipLaunchKernelGGL(k_dual_keep_u, g1(nAC + 1), dim3(NT), 0, st, nAC, a2, ps_u, scal, S_KEEP_S);
            pendS.on = true; pendS.n = nAC;
        } else dual_rank1(nAC, a2, 8, 1.0);
        if (dual_sym) hipLaunchKernelGGL(k_dual_border_sym, g1(nAC + 1), dim3(NT), 0, st, Minv, ldm, nAC, a2, scal, AC, posAC, Sc, r, side, y, yidx, yval,
                                         defer ? ps_u : (double *)nullptr, S_KEEP_S);
        else hipLaunchKernelGGL(k_dual_border, g1(nAC + 1), dim3(NT), 0, st, Minv, ldm, nAC, a2, scal, AC, posAC, Sc, r, side, y, yidx, yval);
        hAC[nAC] = r; hSc[r] = side;
        nAC++;
        nZ = nFR - nAC;
    }
    // a variable joins the fixed set: S loses a_v a_v'/d_v, Sinv += u u'/s with the same u and s as a bordering would use
    void dual_add_bound(int v, int side) {
        if (dual_lazy()) { if (nAC > 0) lz_push(nAC, a2, 8); }
        else { dual_flush(); dual_rank1(nAC, a2, 8, 1.0); }
        hipLaunchKernelGGL(k_set_Sb, dim3(1), dim3(1), 0, st, Sb, v, side);
        hSb[v] = side;
        nFR--;
        nZ = nFR - nAC;
    }
    void dual_remove_constraint(int k, bool carry) {
        const int r = hAC[k];
        if (dual_lazy()) {
            if (lz_n >= LZK) dual_flush();
            hipLaunchKernelGGL(k_lz_colcoef, g1(nAC), dim3(NT), 0, st, Minv, ldm, nAC, k, lz_p(), a3, scal);
            if (carry) hipLaunchKernelGGL(k_dual_carry_remove, dim3(1), dim3(NT), 0, st, nAC, k, 1.0 - last_tau, a3, c_wY);
            hipLaunchKernelGGL(k_lz_remove_fix, g1(std::max(nAC, 1)), dim3(NT), 0, st, nAC, k, a3, scal, lz_vec, lz_stride, lz_n, lz_c);
            lz_n++;
        } else {
        dual_flush();
        if (dual_sym) hipLaunchKernelGGL(k_dual_colcoef_sym, g1(nAC), dim3(NT), 0, st, Minv, ldm, nAC, k, a3, scal);
        else hipLaunchKernelGGL(k_dual_colcoef, g1(nAC), dim3(NT), 0, st, Minv, ldm, nAC, k, a3, scal);
        if (carry) hipLaunchKernelGGL(k_dual_carry_remove, dim3(1), dim3(NT), 0, st, nAC, k, 1.0 - last_tau, a3, c_wY);
        dual_rank1(nAC, a3, 9, 1.0);
        }
        if (dual_sym) hipLaunchKernelGGL(k_dual_move_last_sym, g1(std::max(nAC - 1, 1)), dim3(NT), 0, st, Minv, ldm, nAC, k, AC, posAC, Sc, r, y, nV + r);
        else hipLaunchKernelGGL(k_dual_move_last, g1(std::max(nAC - 1, 1)), dim3(NT), 0, st, Minv, ldm, nAC, k, AC, posAC, Sc, r, y, nV + r);
        if (k != nAC - 1) hAC[k] = hAC[nAC - 1];
        hSc[r] = 0;
        nAC--;
        nZ = nFR - nAC;
    }
    void dual_remove_bound(int v) {
        hipLaunchKernelGGL(k_free_bound_ws, dim3(1), dim3(1), 0, st, Sb, v, y);   // Sb[v] = 0, y[v] = 0
        hSb[v] = 0;
        nFR++;
        nZ = nFR - nAC;
        if (nAC == 0) return;
        fill(a4, nAC, 0.0);
        hipLaunchKernelGGL(k_col_of_A_active, dim3(4), dim3(NT), 0, st, M.Ajc, M.Air, M.Aval, v, posAC, a4);  // a_v
        dual_sinv_times(a4, a3);                                         // w = Sinv a_v
        hipLaunchKernelGGL(k_dual_free_coef, dim3(1), dim3(NT), 0, st, nAC, a4, a3, M.Hval, M.hreg, v, scal);
        if (dual_lazy()) lz_push(nAC, a3, 9); else dual_rank1(nAC, a3, 9, 1.0);
    }
    // factors for a guessed working set: S = B'B with B = D^-1/2 A_cand,FR' (GEMM), Cholesky, inverse. RET_FALLBACK: dependent rows
    // in the guess (or too few candidates): the caller adds them one by one
    int dual_setup_blocked(const std::vector<int> &gc, const std::vector<int> &freev, const std::vector<int> &cand) {
        const int m = nFR, n = (int)cand.size();
        if (ensure_big() != RET_OK) return RET_SETUP_FAILED;
        std::vector<int> fpos(nV, -1);
        for (int i = 0; i < m; i++) fpos[freev[i]] = i;
        LCHK(hipMemcpyAsync(d_fpos, fpos.data(), sizeof(int) * nV, hipMemcpyHostToDevice, st));
        LCHK(hipMemcpyAsync(d_cand, cand.data(), sizeof(int) * n, hipMemcpyHostToDevice, st));
        LCHK(hipMemcpyAsync(d_freev, freev.data(), sizeof(int) * m, hipMemcpyHostToDevice, st));
        const long long lb_ = pad16(m), lg = pad16(n);
        double *B = Y, *G = big, *Ui = big + (size_t)ld * ld;
        LCHK(hipMemsetAsync(B, 0, sizeof(double) * (size_t)lb_ * n, st));
        hipLaunchKernelGGL(k_build_B, dim3(n), dim3(NT), 0, st, M.Arp, M.Aci, M.Arv, d_cand, d_fpos, B, lb_);
        hipLaunchKernelGGL(k_dual_scale_rows, dim3((m + NT - 1) / NT, n), dim3(NT), 0, st, m, d_freev, hinv, B, lb_);
        if (!se0) { (void)hipEventCreate(&se0); (void)hipEventCreate(&se1); (void)hipEventCreate(&se2); }
        setup_stat = SetupStat();
        (void)hipEventRecord(se0, st);
        LCHK(rsqp_dgemm_upper(true, false, n, m, 1.0, B, lb_, B, lb_, 0.0, G, lg, st));      // (the Cholesky factorisation reads the upper triangle)
        (void)hipEventRecord(se1, st);
        LCHK(hipMemsetAsync(dw.flag, 0, sizeof(int) * 4, st));
        LCHK(rsqp_dpotrf_upper(n, G, lg, 1e-10, RSQP_EPS_PD_ABS, &dw, st));      // (a pivot below 1e-10 of its diagonal: the Gram matrix cannot decide independence -- one by one then)
        LCHK(hipMemcpyAsync(h_pinned_i, dw.flag, sizeof(int) * 4, hipMemcpyDeviceToHost, st));
        LCHK(hipStreamSynchronize(st));      // (the host vectors above go out of scope as well)
        if (h_pinned_i[1] != 0) return RET_FALLBACK;
        LCHK(rsqp_dtrtri_upper(n, G, lg, Ui, lg, &dw, st));
        LCHK(rsqp_dtrmmt_upper(n, 1.0, Ui, lg, Minv, ldm, st));
        if (!dual_sym) LCHK(rsqp_mirror_upper(n, Minv, ldm, st));
        (void)hipEventRecord(se2, st);
        {   // what the matrix cores were asked for (algorithmic: the symmetric results counted once): Gram matrix n^2 m, Cholesky
            // n^3/3, triangular inverse n^3/3, U^-1 U^-T n^3/3
            (void)hipEventSynchronize(se2);
            float ms1 = 0.f, ms2 = 0.f;
            (void)hipEventElapsedTime(&ms1, se0, se1); (void)hipEventElapsedTime(&ms2, se1, se2);
            const double dn = n, dm = m;
            setup_stat.valid = 1; setup_stat.dual = 1; setup_stat.m = m; setup_stat.n = n; setup_stat.nZ = m - n;
            setup_stat.ms_tq = ms1; setup_stat.ms_wz = ms2; setup_stat.flops_tq = dn * dn * dm; setup_stat.flops_wz = dn * dn * dn;
        }
        std::vector<int> hpos(nC, -1);
        for (int k = 0; k < n; k++) { hAC[k] = cand[k]; hpos[cand[k]] = k; hSc[cand[k]] = gc[cand[k]]; }
        LCHK(hipMemcpyAsync(AC, hAC.data(), sizeof(int) * n, hipMemcpyHostToDevice, st));
        LCHK(hipMemcpyAsync(posAC, hpos.data(), sizeof(int) * nC, hipMemcpyHostToDevice, st));
        LCHK(hipMemcpyAsync(Sc, hSc.data(), sizeof(int) * nC, hipMemcpyHostToDevice, st));
        LCHK(hipStreamSynchronize(st));
        nAC = n;
        nZ = nFR - nAC;
        chk("dual_setup_blocked");
        return RET_OK;
    }
    // step direction: S dy = db - A dx_FX + A D^-1 dg_FR (one product with A for the right-hand side), D dx_FR = A'dy - dg
    void dual_step_direction() {
        if (!dx_ready) {
            hipLaunchKernelGGL(k_dx_fixed_zero_dy, g1(nV + nC), dim3(NT), 0, st, nV, nC, Sb, lb, ub, lbN, ubN, dx, dy);
            hipLaunchKernelGGL(k_dual_rhs_vec, g1(nV), dim3(NT), 0, st, nV, Sb, hinv, gN, g, dx, w5);
        }       // (else: k_drift_all has formed dx on the fixed variables and the right-hand-side vector w5 already)
        dx_ready = false;
        const bool carried_step = nAC > 0 && carry_valid && (carry_ready || carry_pending);
        if (!carried_step) A_times(w5, c3);      // (the right-hand side of the exact multiplier step; a carried one needs one entry of it at most)
        bool scattered = false;
        if (nAC > 0) {
            if (carry_ready && carry_valid) {
                carried++; stat_carried++;                                    // (transformed by dual_remove_constraint already)
            } else if (carry_pending && carry_valid) {
                hipLaunchKernelGGL(k_dual_carry_add_w, dim3(1), dim3(1024), 0, st, nAC - 1, 1.0 - last_tau, a1, a2, c_wY, scal, dual_carry_row, Sc,
                                   lbA, ubA, lbAN, ubAN, M.Arp, M.Aci, M.Arv, w5, AC, dy + nV);
                carried++; stat_carried++;
                scattered = true;
            } else {
                hipLaunchKernelGGL(k_dual_rhs, g1(nAC), dim3(NT), 0, st, nAC, AC, Sc, lbA, ubA, lbAN, ubAN, c3, a1);
                dual_sinv_times(a1, c_wY);
                carried = 0;
            }
            if (!scattered) hipLaunchKernelGGL(k_scatter_active, g1(nAC), dim3(NT), 0, st, c_wY, AC, nAC, dy + nV);
        }
        carry_pending = carry_ready = false;
        carry_valid = true;
        if (fuse_passes && nC > 0 && !M.denseA) {
            // (sparse A: the dx kernel rides behind the product, in the thread that has just formed the entry -- same bits, a launch less)
            RsqpSpmvDualDx e;
            e.Sb = Sb; e.hinv = hinv; e.Hval = M.Hval; e.hreg = M.hreg; e.gN = gN; e.g = g; e.dx = dx; e.Hdx = Hdx;
            pbegin();
            (void)rsqp_launch_spmv_dualdx(M.blk_c, M.nblk_c, M.Ajc, M.Air, M.Aval, dy + nV, ATdy, e, st);
            pend(5, spmv_bytes(nV, nC));
        } else {
            AT_times(dy + nV, ATdy);
            hipLaunchKernelGGL(k_dual_dx, g1(nV), dim3(NT), 0, st, nV, Sb, hinv, M.Hval, M.hreg, ATdy, gN, g, dx, Hdx);
        }
        A_times(dx, dAx);
        chk("dual_step_direction");
    }

    // ---- step direction -----------------------------------------------------------------
    void step_direction() {
        if (rsh) { rs_step_direction(); return; }
        if (dual) { dual_step_direction(); return; }
        if (!dx_ready) hipLaunchKernelGGL(k_dx_fixed_zero_dy, g1(nV + nC), dim3(NT), 0, st, nV, nC, Sb, lb, ub, lbN, ubN, dx, dy);
        dx_ready = false;           // (set by drift_correction, whose kernel then has done this already)
        // what of the last step direction is carried over the change (range space: wY = Minv bA, xY = Y wY; null space: wZ)
        bool carry_null = false;      // the null-space part is carried too (needs the deferred reflection's v / col / beta)
        const bool was_ready = carry_ready && carry_valid;
        const bool carry_add = !was_ready && carry_pending && carry_valid && nAC > 0;
        if (carry_add)
            carry_null = carry_null_enabled && pendZ.on && pendZ.ncols == nZ && pendW.on && pendW.nZold - 1 == nZ && nZ > 0 && nV <= 16384 &&
                         (((reinterpret_cast<unsigned long long>(pz_t)) & 15) == 0);
        const bool carry_null_grow = was_ready && carry_null_enabled && nZ > 0 && 1.0 - last_tau > 1e-6;
        // A dx_FX, H dx_FX: dx is zero on the free variables here (mean over the dense 2048 x 4096 cold start: 196 of
        // 2048 entries live), so the column-major products read the live columns only. A removed constraint with both parts carried
        // needs neither product (no right-hand side is formed), an added one with both parts carried only A dx_FX
        if (carry_null_grow) {
        } else if (live_skip && M.denseA && M.haveH && M.denseH && M.hreg == 0.0 && 4 * (nV - nFR) < nV) {
            if (!gemv_n1_pair(M.denseA, nC, nC, nV, dx, c1, M.denseH, nV, nV, nV, dx, w2, true, nV - nFR)) {
                gemv_n_live(M.denseA, nC, nC, nV, dx, c1, nV - nFR);
                gemv_n_live(M.denseH, nV, nV, nV, dx, w2, nV - nFR);       // H symmetric
            }
        } else if (carry_add && carry_null) A_times(dx, c1);    // (tmpg is not read on this path: k_sd_prep forms it from a stale w2)
        else AH_times(dx, c1, w2);
        // bA -> a1, tmpg -> w1 (a carried removed constraint whose null-space part is carried as well needs neither)
        if (!carry_null_grow)
            hipLaunchKernelGGL(k_sd_prep, g1(std::max(nAC, nV)), dim3(NT), 0, st, nAC, AC, Sc, lbA, ubA, lbAN, ubAN, c1, a1, nV, w2, gN, g, w1);
        if (was_ready) {
            carried++; stat_carried++;                                     // (transformed by remove_constraint_tq already)
        } else if (carry_add) {
            const double om = 1.0 - last_tau;
            hipLaunchKernelGGL(k_carry_add, dim3(g1(std::max(nV, 1)).x + 1), dim3(NT), 0, st, nAC - 1, om, a1, c_xi, c_wY, scal, S_KEEP_ETA,
                               S_KEEP_WLAST, nV, Yc(nAC - 1), c_xY, carry_null ? nZ : -1, pz_v, pw_col, c_wZ, S_KEEP_BETA);
            carried++; stat_carried++;
            if (carry_null) {
                stat_carried_null++;
                pendW.on = false;
                wz_shrink_now(pendW.nZold, pw_s, pz_v, pw_col, S_KEEP_BETA, S_KEEP_THETA);        // (no product with Wz to ride on)
            }
        } else {
            if (pendY.on) { pendY.on = false; ger(Y, ld, nV, pendY.n, py_t, py_v, S_KEEP_BETA2, -1.0); }      // (a removal whose step
            if (pendM.on) { pendM.on = false; ger(Minv, ldm, pendM.n, pendM.n, py_v, pm_s, S_KEEP_BETA2, -1.0); }   //  direction is exact after all)
            gemv_n(Minv, ldm, nAC, nAC, a1, 1.0, 0.0, nullptr, c_wY);
            gemv_n(Y, ld, nV, nAC, c_wY, 1.0, 0.0, nullptr, c_xY);         // xY
            carried = 0;
        }
        carry_pending = carry_ready = false;
        carry_valid = true;
        double *const w3 = c_xY;                                           // (xY lives in its own buffer: the next step may scale it)
        // null space: wZ = -Wz Z'(tmpg + H xY) ; dx_FR = xY + Z wZ
        if (carry_null_grow) {
            // a removed constraint: wZ from the bordered system (the new null-space column is Z[:, nZ - 1], u / rho^2 are wz_grow's)
            stat_carried_null++;
            hipLaunchKernelGGL(k_carry_wZ_grow, dim3(1), dim3(NT), 0, st, nZ - 1, nV, 1.0 - last_tau, Zc(nZ - 1), Hdx, gN, g, wz2, c_wZ, scal, 13,
                               S_KEEP_WLAST);
            gemv_n(Z, ld, nV, nZ, c_wZ, 1.0, 1.0, w3, w4, Sb, dx);
        } else if (carry_null) {
            // wZ carried (k_carry_add above): the deferred reflection of Z rides on the product Z wZ instead of on Z'w
            pendZ.on = false;
            pbegin();
            hipLaunchKernelGGL((k_ger_gemv_n1<NT>), dim3((nV + 15) / 16), dim3(NT), 0, st, Z, ld, nV, nZ, pz_t, pz_v, scal, S_KEEP_BETA, -1.0,
                               c_wZ, 1.0, 1.0, w3, w4, Sb, dx);
            pend(8, 16.0 * nV * (double)nZ);
            chk("ger_gemv_n1");
        } else {
            H_times(w3, w2, w1);                                           // w2 = H xY + tmpg
            gemv_t_Z_pending(w2, nullptr, wz1);                           // (+ the deferred reflection of Z)
            gemv_n_Wz_pending(wz1, -1.0, c_wZ);                            // (+ the deferred shrinking of Wz)
            gemv_n(Z, ld, nV, nZ, c_wZ, 1.0, 1.0, w3, w4, Sb, dx);         // xY + Z wZ, merged into dx on the free variables
        }
        // multipliers: dyAC = Minv' Y'(H dx + dg); A dx (for the ratio tests) rides along with H dx
        AH_times(dx, dAx, Hdx);
        {
            const bool al = (((reinterpret_cast<unsigned long long>(Hdx) | reinterpret_cast<unsigned long long>(gN) |
                               reinterpret_cast<unsigned long long>(g) | reinterpret_cast<unsigned long long>(a1)) & 15) == 0);
            if (pendY.on && pendY.n == nAC && nAC > 0 && al) {            // (+ the deferred reflection of Y behind a removed constraint)
                pendY.on = false;
                pbegin();
                hipLaunchKernelGGL(k_ger_gemv_t, dim3((nAC + GT_COLS - 1) / GT_COLS), dim3(NT), 0, st, Y, ld, nV, nAC, py_t, py_v, scal, S_KEEP_BETA2,
                                   -1.0, Hdx, (const double *)nullptr, a1, gN, g, (const int *)nullptr);
                pend(6, 16.0 * nV * (double)nAC);
            } else {
                if (pendY.on) { pendY.on = false; ger(Y, ld, nV, pendY.n, py_t, py_v, S_KEEP_BETA2, -1.0); }
                GtTask t = gt_task(Y, ld, nV, nAC, Hdx, a1);               // a1 = Y'(H dx + (gN - g))
                t.xa = gN; t.xb = g;
                gemv_t_task(t);
            }
            if (pendM.on && pendM.n == nAC && nAC > 0 && al) {            // (+ the deferred reflection of Minv)
                pendM.on = false;
                pbegin();
                hipLaunchKernelGGL(k_ger_gemv_t, dim3((nAC + GT_COLS - 1) / GT_COLS), dim3(NT), 0, st, Minv, ldm, nAC, nAC, py_v, pm_s, scal,
                                   S_KEEP_BETA2, -1.0, a1, (const double *)nullptr, dy + nV, (const double *)nullptr, (const double *)nullptr, AC);
                pend(6, 16.0 * nAC * (double)nAC);
            } else {
                if (pendM.on) { pendM.on = false; ger(Minv, ldm, pendM.n, pendM.n, py_v, pm_s, S_KEEP_BETA2, -1.0); }
                GtTask u = gt_task(Minv, ldm, nAC, nAC, a1, dy + nV);      // dy[nV + AC[j]] = sum_i Minv[i][j] a1[i]
                u.omap = AC;
                gemv_t_task(u);
            }
        }
        AT_times(dy + nV, ATdy);
        // (dy of the fixed variables: formed by k_ratio1, which is the next kernel)
        chk("step_direction");
    }

    // A x, A'y_C and H x follow the iterate by axpy (their increments are by-products of the
    // step direction); every REFRESH changes, and whenever an exchange shifted the duals, they
    // are recomputed exactly so that rounding cannot accumulate
    static constexpr int REFRESH = 8;
    void refresh_products() {
        A_times(x, Ax);
        AT_times(y + nV, ATy);
        H_times(x, Hx);
        dirty_products = false;
        since_refresh = 0;
    }
    void drift_correction() {
        bool refreshed = false;
        if (dirty_products || ++since_refresh >= REFRESH) {
            hipLaunchKernelGGL(k_fix_x, g1(nV), dim3(NT), 0, st, nV, Sb, lb, ub, x);
            refresh_products();
            refreshed = true;
        }
        hipLaunchKernelGGL(k_drift_all, g1(nV + nC), dim3(NT), 0, st, nV, nC, Sb, Sc, lb, ub, x, Ax, lbA, ubA, ATy,
                           y, Hx, g, lbN, ubN, dx, dy, dual ? hinv : (const double *)nullptr, gN, dual ? w5 : (double *)nullptr);
        if (rsh && refreshed) rs_refresh_p();      // p = H^-1 (gN - g), A p: exact again (in between they shrink with the step, k_step_all)
        dx_ready = true;
        chk("drift");
    }

    int homotopy(int maxit, int *nWSR) {
        int iter = 0, rcode = RET_OK;
        long long kind_count[5] = {0, 0, 0, 0, 0};   // (RSQP_PROFILE: changes by kind)
        stat_carried = stat_carried_null = 0;
        double sum_nFR = 0.0, sum_nAC = 0.0, sum_nZ = 0.0;      // reported by RSQP_PROFILE: the sizes the products run on
        status = QPS_PERFORMINGHOMOTOPY;
        dx_ready = false;
        pendZ.on = pendW.on = pendY.on = pendM.on = pendS.on = false;      // (nothing is deferred across solves; a solve that failed half-way leaves nothing behind)
        lz_n = 0;
        carry_valid = carry_pending = carry_ready = false;
        carried = 0;
        refresh_products();
        hipLaunchKernelGGL(k_rerelax, g1(nV), dim3(NT), 0, st, nV, Sb, x, lbN, ubN, lb, ub);
        if (nC > 0) hipLaunchKernelGGL(k_rerelax, g1(nC), dim3(NT), 0, st, nC, Sc, Ax, lbAN, ubAN, lbA, ubA);
        if (rsh) { pendR.on = false; lz_n = 0; rs_refresh_p(); }
        for (;;) {
            step_direction();
            hipLaunchKernelGGL(k_ratio1, dim3(nblk_ratio), dim3(NT), 0, st, nV, nC, Sb, Sc, x, y, dx, dy, Ax, dAx, lb, ub,
                               lbA, ubA, lbN, ubN, lbAN, ubAN, pt, pid, d_ctl, next_seq(), scal + 30, res_id, Hdx, gN, g, ATdy, dy);
            // the homotopy step decodes the winner on the device and runs while the host waits for its own copy
            hipLaunchKernelGGL(k_step_all, g1(nV + nC), dim3(NT), 0, st, nV, nC, scal + 30, iter < maxit ? 1 : 0, Sb, x, g, lb, ub, gN, lbN,
                               ubN, dx, ATdy, ATy, Hdx, Hx, lbA, ubA, lbAN, ubAN, dAx, Ax, dy, y, rsh ? rs_p : (double *)nullptr,
                               rsh ? rs_Ap : (double *)nullptr);
            if (wait_ctl() != RET_OK) return wait_failed();
            double tau = h_ctl[0];
            const int bid = (int)h_ctl[1];
            int kind = 0, idx = -1, side = 0;
            if (bid != 0x7fffffff) {
                if (bid < nC) { kind = 1; idx = bid; }
                else if (bid < nC + nV) { kind = 2; idx = bid - nC; }
                else if (bid < 2 * nC + nV) { kind = 3; idx = bid - nC - nV; side = -1; }
                else if (bid < 3 * nC + nV) { kind = 3; idx = bid - 2 * nC - nV; side = 1; }
                else if (bid < 3 * nC + 2 * nV) { kind = 4; idx = bid - 3 * nC - nV; side = -1; }
                else { kind = 4; idx = bid - 3 * nC - 2 * nV; side = 1; }
            } else tau = 1.0;
            const int done = kind == 0;
            (void)tau;
            if (extra_sync) (void)hipStreamSynchronize(st);      // tuning: what one more host round trip per change costs
            if (done) { A_times(x, Ax); status = QPS_SOLVED; break; }
            if (iter >= maxit) { rcode = RET_MAX_NWSR; break; }
            last_tau = tau;
            kind_count[kind]++;
            rcode = change_active_set(kind, idx, side);
            if (!carry_pending && !carry_ready) carry_valid = false;      // anything but a plain added / removed constraint: exact products next
            if (rcode == RET_INFEASIBLE) { infeasible = 1; break; }
            if (rcode == RET_UNBOUNDED) { unbounded = 1; break; }
            if (rcode != RET_OK) break;
            iter++;
            sum_nFR += nFR; sum_nAC += nAC; sum_nZ += nZ;
            drift_correction();
        }
        if ((profile || getenv("RSQP_LARGE_WAITSTAT")) && iter > 0)
            fprintf(stderr, "[rsqp profile] homotopy: %d changes, mean nFR %.1f nAC %.1f nZ %.1f (nV %d nC %d); host waited %.3f s in %lld round trips; return code %d, final nFR %d nAC %d\n", iter,
                    sum_nFR / iter, sum_nAC / iter, sum_nZ / iter, nV, nC, wait_seconds, wait_calls, rcode, nFR, nAC);
        if (profile && iter > 0)
            fprintf(stderr, "[rsqp profile] changes by kind: constraint out %lld, bound out %lld, constraint in %lld, bound in %lld; step directions with the range-space part carried %lld, null-space part too %lld\n",
                    kind_count[1], kind_count[2], kind_count[3], kind_count[4], stat_carried, stat_carried_null);
        flush_pending();
        dual_flush();
        rs_flush();
        if (rsh && rs_refine_enabled && status == QPS_SOLVED && rcode == RET_OK && iter > 0) rs_refine();      // (no change, nothing accumulated: a hot start on unchanged data returns the same bits)
        *nWSR = iter;
        return rcode;
    }


    // ---- GENERAL range-space path (rsh; DESIGN 4.5, kernels in qp_rs_kernels.h, bodies in qp_rs_path.h) ----------------------------
    // any symmetric positive definite H: active bounds AND constraints are rows of C, Sinv = (C H^-1 C')^-1 (upper triangle, in the
    // buffer of Wz, leading dimension ld), H^-1 an operator built once per Hessian (banded LDL' / explicit dense inverse in Z)
    bool rsh_enabled = getenv("RSQP_LARGE_NO_RSH") == nullptr;
    bool force_null_space = false;   // this solve is the null-space retry of a range-space attempt that failed numerically (solve())
    bool rs_force_dense = getenv("RSQP_LARGE_RSH_DENSE") != nullptr;      // (tests: the dense operator on a banded Hessian)
    bool rsh = false;
    int rs_kind = 0;                 // 1: banded factor (k_band_apply), 2: explicit dense inverse in Z, 3: 2 + the static tableau
                                     // WW = [I; A] H^-1 [I A'] of small dense problems (nV + nC <= WW_MAX)
    static constexpr int WW_MAX = 8192, S_WW_AD = 47;
    bool rs_no_ww = getenv("RSQP_LARGE_NO_TABLEAU") != nullptr;
    double *rs_WW = nullptr, *rs_dA = nullptr;
    long long ldw = 0;
    int rs_cur_id = -1;
    int rs_build_ww();
    int nR = 0;                      // active rows: (nV - nFR) + nAC
    int *R = nullptr, *posR = nullptr;
    std::vector<int> hR, hposR;
    double *rs_p = nullptr, *rs_Ap = nullptr, *ra1 = nullptr, *ra2 = nullptr, *ra3 = nullptr, *ra4 = nullptr, *rs_dl = nullptr, *rs_ps_u = nullptr;
    double *rs_G = nullptr;          // ld x ld scratch of the blocked set-up (allocated on first use)
    double *band_buf = nullptr;      // the nine arrays of the banded operator
    BandOp band{};
    bool band_attr_set = false;
    bool band_hinv = false;          // Z holds the explicit inverse of the banded H as well (products of incoming rows)
    bool rs_band_columns = getenv("RSQP_LARGE_BAND_NO_COLUMNS") == nullptr;
    double band_seq = 0.0;           // tag of the last multi-workgroup product (k_band_apply_mw)
    bool band_mw = getenv("RSQP_LARGE_BAND_1WG") == nullptr;      // (tests / tuning: the one-workgroup kernel for single vectors as well)
    struct { bool on = false; int n = 0; } pendR;
    // lazy rank-1 terms of Sinv (qp_rs_kernels.h): vectors lz_vec[k] (stride lz_stride), coefficients lz_c[k], dots lz_d[k]
    bool lz_enabled = getenv("RSQP_LARGE_NO_LAZY") == nullptr;
    double *lz_vec = nullptr, *lz_c = nullptr, *lz_d = nullptr;
    long long lz_stride = 0;
    int lz_n = 0;
    LzP lz_p() const { LzP q; q.vec = lz_vec; q.stride = lz_stride; q.c = lz_c; q.np = lz_n; return q; }
    int rs_prepare(bool *ok);
    int rs_build_band(const std::vector<double> &hv, bool *ok);
    int rs_build_dense(bool *ok);
    void rs_hinv_apply(const double *in, const double *sub, double *out, bool fix_dx);
    void band_launch(int ncols, const double *in, const double *sub, double *out, long long ldc, bool qmode);
    void rs_rank1(int n, const double *v, int slot, double cs);
    void rs_flush();
    void rs_sinv_times(const double *wv, double *out, bool scatter_dy = false);
    bool rs_carry_enabled = getenv("RSQP_LARGE_NO_CARRY") == nullptr;
    int rs_carry_id = -1;
    void rs_products(int id);
    void rs_residual();
    int rs_li_decision(bool *li);
    void rs_add_row(int id, int side, int yidx, double yval);
    void rs_remove_row(int k, bool carry = false);
    int rs_change_active_set(int kind, int idx, int side);
    void rs_step_direction();
    void rs_refresh_p();
    void rs_refine();
    bool rs_refine_enabled = getenv("RSQP_LARGE_NO_REFINE") == nullptr;
    int rs_setup_rows(const std::vector<int> &rows, const std::vector<int> &gb, const std::vector<int> &gc);
    int rs_setup(const std::vector<int> &gb, const std::vector<int> &gc);
    void rs_count(int id, int delta);

    // ---- blocked set-up (dense_la.hip) --------------------------------------------------------------
    static constexpr int RET_FALLBACK = -77;
    int ensure_big() {
        if (big) return RET_OK;
        LCHK(hipMalloc(reinterpret_cast<void **>(&big), sizeof(double) * 2 * (size_t)ld * ld));
        return RET_OK;
    }
    // Y, Z, Minv for the candidate constraints `cand` (all taken: a linearly dependent one makes
    // the caller fall back to the sequential construction, which skips it). Works in the compressed
    // coordinates of the nFR free variables: B = A_cand,FR' = Q R, Y = Q[:, :n], Z = Q[:, n:],
    // A_AC Y = R'  =>  Minv = R^-T.
    int setup_tq_blocked(const std::vector<int> &gc, const std::vector<int> &freev, const std::vector<int> &cand) {
        const int m = nFR, n = (int)cand.size();
        if (ensure_big() != RET_OK) return RET_SETUP_FAILED;
        std::vector<int> fpos(nV, -1);
        for (int i = 0; i < m; i++) fpos[freev[i]] = i;
        LCHK(hipMemcpyAsync(d_fpos, fpos.data(), sizeof(int) * nV, hipMemcpyHostToDevice, st));
        LCHK(hipMemcpyAsync(d_cand, cand.data(), sizeof(int) * n, hipMemcpyHostToDevice, st));
        LCHK(hipMemcpyAsync(d_freev, freev.data(), sizeof(int) * m, hipMemcpyHostToDevice, st));
        // leading dimensions of the factorisation's operands padded to 16 doubles like the engine state (m = nFR is
        // whatever the working set leaves: 9983 on the sparse 10k x 20k sequence)
        const long long lb_ = pad16(m), lx = pad16(n);
        double *B = Y;            // m x n, ld lb_ <= ld (Y is rewritten at the end)
        double *X = big;          // n x n  R^-1
        double *Q = big + (size_t)ld * ld;   // m x m
        for (int attempt = 0; attempt < 2; attempt++) {
            LCHK(hipMemsetAsync(B, 0, sizeof(double) * (size_t)lb_ * n, st));
            LCHK(hipMemsetAsync(dw.flag, 0, sizeof(int) * 4, st));
            hipLaunchKernelGGL(k_build_B, dim3(n), dim3(NT), 0, st, M.Arp, M.Aci, M.Arv, d_cand, d_fpos, B, lb_);
            LCHK(rsqp_dgeqrf(m, n, B, lb_, RSQP_EPS_LI, &dw, st));
            LCHK(hipMemcpyAsync(h_pinned_i, dw.flag, sizeof(int) * 4, hipMemcpyDeviceToHost, st));
            LCHK(hipStreamSynchronize(st));
            // a panel too ill-conditioned for the Cholesky-QR panel factorisation: once more with the column kernel, which decides
            // the independence of every column on its own
            if (h_pinned_i[2] == 0 || !dw.panel_cholqr) break;
            dw.panel_cholqr = false;
        }
        dw.panel_cholqr = true;
        if (h_pinned_i[0] != 0) {   // dependent rows in the guess: start over, one constraint at a time
            if (nZ > 0) {
                LCHK(hipMemsetAsync(Z, 0, sizeof(double) * (size_t)ld * nZ, st));
                hipLaunchKernelGGL(k_unit_cols, g1(nZ), dim3(NT), 0, st, Z, ld, d_freev, nZ);
            }
            return RET_FALLBACK;
        }
        LCHK(rsqp_dtrtri_upper(n, B, lb_, X, lx, &dw, st));
        hipLaunchKernelGGL(k_transpose, dim3((n + 31) / 32, (n + 31) / 32), dim3(32, 8), 0, st, n, X, lx, Minv, ldm);
        LCHK(rsqp_dorgqr(m, n, B, lb_, Q, lb_, &dw, st));
        LCHK(hipMemsetAsync(Y, 0, sizeof(double) * (size_t)ld * n, st));
        if (m > n) LCHK(hipMemsetAsync(Z, 0, sizeof(double) * (size_t)ld * (m - n), st));
        hipLaunchKernelGGL(k_scatter_Q, dim3((m + NT - 1) / NT, m), dim3(NT), 0, st, m, n, Q, lb_, d_freev, Y, Z, ld);
        // working-set bookkeeping
        std::vector<int> hpos(nC, -1);
        for (int k = 0; k < n; k++) { hAC[k] = cand[k]; hpos[cand[k]] = k; hSc[cand[k]] = gc[cand[k]]; }
        LCHK(hipMemcpyAsync(AC, hAC.data(), sizeof(int) * n, hipMemcpyHostToDevice, st));
        LCHK(hipMemcpyAsync(posAC, hpos.data(), sizeof(int) * nC, hipMemcpyHostToDevice, st));
        LCHK(hipMemcpyAsync(Sc, hSc.data(), sizeof(int) * nC, hipMemcpyHostToDevice, st));
        LCHK(hipStreamSynchronize(st));   // the host vectors above go out of scope
        nAC = n;
        nZ = m - n;
        chk("setup_tq_blocked");
        return RET_OK;
    }
    // Wz = (Z'(H + hreg I)Z)^-1 = U^-1 U^-T with Z'HZ = U'U; the pivots pass the same definiteness
    // test as the bordering (wz_grow)
    int setup_wz_blocked() {
        if (ensure_big() != RET_OK) return RET_SETUP_FAILED;
        double *HZ = big, *G = big + (size_t)ld * ld;
        const long long lg = pad16(nZ);
        // H Z, column by column of Z in one batched launch
        if (!M.haveH) LCHK(hipMemsetAsync(HZ, 0, sizeof(double) * (size_t)ld * nZ, st));
        else if (M.denseH) LCHK(rsqp_dgemm(false, false, nV, nZ, nV, 1.0, M.denseH, nV, Z, ld, 0.0, HZ, ld, st));
        else LCHK(rsqp_launch_spmv(M.blk_h, M.nblk_h, M.Hjc, M.Hir, M.Hval, Z, HZ, nZ, 0, 0, ld, ld, st));
        if (M.hreg != 0.0) {
            const long long tot = (long long)ld * nZ;
            hipLaunchKernelGGL(k_add_scaled, dim3((unsigned)((tot + NT - 1) / NT)), dim3(NT), 0, st, tot, M.hreg, Z, HZ);
        }
        LCHK(rsqp_dgemm(true, false, nZ, nZ, nV, 1.0, Z, ld, HZ, ld, 0.0, G, lg, st));
        LCHK(hipMemsetAsync(dw.flag, 0, sizeof(int) * 4, st));
        LCHK(rsqp_dpotrf_upper(nZ, G, lg, RSQP_EPS_PD_REL, RSQP_EPS_PD_ABS, &dw, st));
        LCHK(hipMemcpyAsync(h_pinned_i, dw.flag, sizeof(int) * 4, hipMemcpyDeviceToHost, st));
        LCHK(hipStreamSynchronize(st));
        if (h_pinned_i[1] != 0) return RET_SETUP_FAILED;   // not positive definite on the null space
        double *Ui = HZ;   // nZ x nZ
        LCHK(rsqp_dtrtri_upper(nZ, G, lg, Ui, lg, &dw, st));
        LCHK(rsqp_dgemm_tri(false, true, nZ, nZ, nZ, 1.0, Ui, lg, Ui, lg, 0.0, Wz, ld, 2, st));      // (Ui upper triangular: row i starts at column i; both triangles of the result wanted)
        chk("setup_wz_blocked");
        return RET_OK;
    }

    // ---- auxiliary QP -------------------------------------------------------------------------
    // working-set guess on the host (hSb / hSc targets), x / y on the device already
    int setup_aux(const std::vector<int> &gb, const std::vector<int> &gc) {
        status = QPS_PREPARINGAUXILIARYQP;
        infeasible = unbounded = 0;
        nFR = nAC = nZ = 0;
        const double t_setup0 = now_s();
        // 1. bounds: Z = unit columns of the free variables (no constraint active yet)
        LCHK(hipMemcpyAsync(Sb, gb.data(), sizeof(int) * nV, hipMemcpyHostToDevice, st));
        hSb = gb;
        std::fill(hSc.begin(), hSc.end(), 0);
        LCHK(hipMemsetAsync(Sc, 0, sizeof(int) * std::max(nC, 1), st));
        LCHK(hipMemsetAsync(posAC, 0xff, sizeof(int) * std::max(nC, 1), st));
        std::vector<int> freev;
        for (int v = 0; v < nV; v++) if (gb[v] == 0) freev.push_back(v);
        nFR = nZ = (int)freev.size();
        if (rsh) {
            const int rcs = rs_setup(gb, gc);
            if (rcs != RET_OK) return rcs;
            if (profile) { (void)hipStreamSynchronize(st); fprintf(stderr, "[rsqp profile] setup_aux (general range-space path, %s H^-1): nFR %d nAC %d, t=%.3f s\n", rs_kind == 1 ? "banded" : "dense", nFR, nAC, now_s() - t_setup0); }
        } else if (dual) {
            // range-space path: Sinv for the guessed constraints (blocked: GEMM + Cholesky + inverse; else one bordering each)
            A_times(x, Ax);
            std::vector<int> cand;
            for (int r = 0; r < nC; r++) if (gc[r] != 0) cand.push_back(r);
            bool done = cand.empty();
            if (!done && blocked_setup && (int)cand.size() >= BLOCKED_MIN && (int)cand.size() <= nFR) {
                const int rcb = dual_setup_blocked(gc, freev, cand);
                if (rcb == RET_OK) done = true;
                else if (rcb != RET_FALLBACK) return rcb;
            }
            if (!done) {
                for (int r : cand) {
                    dual_constraint_products(r);
                    bool li = false;
                    if (dual_li_decision(&li) != RET_OK) return RET_SETUP_FAILED;
                    if (li) dual_add_constraint(r, gc[r]);
                }
                dual_flush();
            }
            nZ = nFR - nAC;
            if (profile) { (void)hipStreamSynchronize(st); fprintf(stderr, "[rsqp profile] setup_aux (range-space path): nFR %d nAC %d, t=%.3f s\n", nFR, nAC, now_s() - t_setup0); }
        } else {
        wz_sym = wz_sym_enabled && nV >= wz_sym_min;
        if (nZ > 0) {
            LCHK(hipMemsetAsync(Z, 0, sizeof(double) * (size_t)ld * nZ, st));
            // free-variable index list staged in dy (read as ints; dy is rewritten before its next use)
            LCHK(hipMemcpyAsync(reinterpret_cast<int *>(dy), freev.data(), sizeof(int) * freev.size(), hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(k_unit_cols, g1(nZ), dim3(NT), 0, st, Z, ld, reinterpret_cast<const int *>(dy), nZ);
        }
        // 2. constraints. Many of them (a hot start with new matrices, a warm start): blocked Householder
        //    QR of A_AC,FR' on the matrix cores; otherwise one reflection of Z per constraint.
        //    The inverse reduced Hessian is built afterwards in both cases.
        A_times(x, Ax);
        std::vector<int> cand;
        for (int r = 0; r < nC; r++) if (gc[r] != 0) cand.push_back(r);
        bool tq_done = false;
        if (blocked_setup && (int)cand.size() >= BLOCKED_MIN && (int)cand.size() <= nFR) {
            setup_stat = SetupStat();
            if (!se0) { (void)hipEventCreate(&se0); (void)hipEventCreate(&se1); (void)hipEventCreate(&se2); }
            (void)hipEventRecord(se0, st);
            const int rcb = setup_tq_blocked(gc, freev, cand);
            (void)hipEventRecord(se1, st);
            if (rcb == RET_OK) {
                const double m = nFR, n = nAC;
                (void)hipEventSynchronize(se1);
                float ms = 0.f; (void)hipEventElapsedTime(&ms, se0, se1);
                setup_stat.valid = 1; setup_stat.m = nFR; setup_stat.n = nAC; setup_stat.ms_tq = ms;
                // Householder QR of m x n: 2 n^2 (m - n/3); explicit Q (m x m) from n reflectors: 4 (m^2 n - m n^2 + n^3/3);
                // inverse of the n x n triangle: n^3 / 3
                setup_stat.flops_tq = 2.0 * n * n * (m - n / 3.0) + 4.0 * (m * m * n - m * n * n + n * n * n / 3.0) + n * n * n / 3.0;
            }
            if (rcb == RET_OK) tq_done = true;
            else if (rcb != RET_FALLBACK) return rcb;
        }
        if (!tq_done) {
            wz_enabled = false;
            for (int r = 0; r < nC; r++) {
                if (gc[r] == 0) continue;
                constraint_products(r);
                bool li = false;
                if (li_decision(&li) != RET_OK) { wz_enabled = true; return RET_SETUP_FAILED; }
                if (li) add_constraint(r, gc[r], false);
            }
            wz_enabled = true;
        }
        if (profile) { (void)hipStreamSynchronize(st); fprintf(stderr, "[rsqp profile] setup_aux: TQ part done (%s), nFR %d nAC %d nZ %d, t=%.3f s\n", tq_done ? "blocked QR" : "sequential", nFR, nAC, nZ, now_s() - t_setup0); }
        // 3. Wz = (Z'HZ)^-1: blocked Cholesky + inverse, or bordering over the null-space columns
        bool wz_done = false;
        if (blocked_setup && nZ >= BLOCKED_MIN) {
            if (!se0) { (void)hipEventCreate(&se0); (void)hipEventCreate(&se1); (void)hipEventCreate(&se2); }
            (void)hipEventRecord(se1, st);
            const int rcw = setup_wz_blocked();
            (void)hipEventRecord(se2, st);
            if (rcw == RET_OK && setup_stat.valid) {
                const double z = nZ, v = nV;
                (void)hipEventSynchronize(se2);
                float ms = 0.f; (void)hipEventElapsedTime(&ms, se1, se2);
                setup_stat.nZ = nZ; setup_stat.ms_wz = ms;
                // Z'(HZ) (symmetric: v z^2 multiply-adds = 2 v z^2 / 2 ... counted as the half a SYRK needs: v z^2 flops x 1),
                // Cholesky z^3/3, triangular inverse z^3/3, U^-1 U^-T z^3/3 (symmetric product)
                setup_stat.flops_wz = v * z * z + z * z * z;
                if (M.denseH) setup_stat.flops_wz += 2.0 * v * v * z;
            }
            if (rcw == RET_OK) wz_done = true;
            else return rcw;
        }
        if (!wz_done) {
            const int nZf = nZ;
            nZ = 0;
            for (int k = 0; k < nZf; k++) {
                int pd = 0;
                if (wz_grow(&pd) != RET_OK) return RET_SETUP_FAILED;
                if (!pd) return RET_SETUP_FAILED;
            }
        }
        }   // (null-space path)
        // multipliers: zero when inactive, clipped to the admissible sign
        hipLaunchKernelGGL(k_clip_y, g1(nV + nC), dim3(NT), 0, st, nV, nC, Sb, Sc, y);
        AT_times(y + nV, w1);
        H_times(x, w2);
        hipLaunchKernelGGL(k_aux_v, g1(nV), dim3(NT), 0, st, nV, Sb, x, w1, y, w2, lbN, ubN, g, lb, ub);
        if (nC > 0) hipLaunchKernelGGL(k_aux_c, g1(nC), dim3(NT), 0, st, nC, Sc, Ax, lbAN, ubAN, lbA, ubA);
        status = QPS_AUXILIARYQPSOLVED;
        if (profile) { (void)hipStreamSynchronize(st); fprintf(stderr, "[rsqp profile] setup_aux: done, t=%.3f s\n", now_s() - t_setup0); }
        return RET_OK;
    }
};

#include "qp_rs_path.h"

// =====================================================================================
// public wrapper
// =====================================================================================
RsqpLargeEngine::RsqpLargeEngine() : p_(new Impl()) {}
RsqpLargeEngine::~RsqpLargeEngine() { delete p_; }

long long RsqpLargeEngine::bytes_needed(int nV, int nC) {
    const long long nA = Impl::pad16(std::min(nV, nC)), l = Impl::pad16(nV);
    // Z, Wz, Y, Minv, vectors + the scratch of the blocked set-up (2 ld^2, allocated on first use) and its panels
    return 8LL * (4LL * l * l + l * nA + nA * nA + 40LL * (nV + nC) + 9LL * 64 * nV);
}

hipError_t RsqpLargeEngine::init(int nV, int nC, hipStream_t stream) {
    Impl &P = *p_;
    P.nV = nV; P.nC = nC; P.nAmax = std::min(nV, nC); P.st = stream;
    // leading dimensions padded to 16 doubles (128 B): every column of Z / Y / Wz / Minv starts on a cache-line boundary,
    // so the 16-byte loads of the column-major products are aligned whatever nV is
    P.ld = Impl::pad16(nV); P.ldm = Impl::pad16(std::max(P.nAmax, 1));
    hipError_t e;
#define DA(ptr, n) if ((e = dalloc(&P.ptr, (size_t)(n))) != hipSuccess) return e
    DA(Z, (size_t)P.ld * nV); DA(Wz, (size_t)P.ld * nV); DA(Y, (size_t)P.ld * std::max(P.nAmax, 1)); DA(Minv, (size_t)P.ldm * P.ldm);
    DA(x, nV); DA(g, nV); DA(lb, nV); DA(ub, nV); DA(gN, nV); DA(lbN, nV); DA(ubN, nV); DA(dx, nV);
    DA(w1, nV); DA(w2, nV); DA(w3, nV); DA(w4, nV); DA(w5, nV); DA(w6, nV); DA(wz1, nV); DA(wz2, nV); DA(wz3, nV);
    DA(pz_t, nV); DA(pz_v, nV); DA(pw_s, nV); DA(pw_col, nV);
    DA(c_wY, P.nAmax + 2); DA(c_wY2, P.nAmax + 2); DA(c_xY, nV); DA(c_xi, P.nAmax + 2); DA(c_wZ, nV);
    DA(py_t, nV); DA(py_v, P.nAmax + 2); DA(pm_s, P.nAmax + 2);
    DA(Ax, nC); DA(lbA, nC); DA(ubA, nC); DA(lbAN, nC); DA(ubAN, nC); DA(dAx, nC); DA(c1, nC); DA(c2, nC); DA(c3, nC);
    DA(a1, P.nAmax + 2); DA(a2, P.nAmax + 2); DA(a3, P.nAmax + 2); DA(a4, P.nAmax + 2);
    DA(y, nV + nC); DA(dy, nV + nC);
    P.part_cap = std::max<long long>((long long)std::max(nV, nC) * 64, (long long)std::max(nV, nC) * (std::max(nV, nC) / 64 + 1));
    DA(part, (size_t)P.part_cap);
    DA(ATy, nV); DA(Hx, nV); DA(Hdx, nV); DA(ATdy, nV);
    DA(scal, 64);
    P.nblk_ratio = std::min(1024, std::max(1, (nV + nC + NT - 1) / NT));
    DA(pt, P.nblk_ratio); DA(res_t, 2);
    DA(Sall, nV + std::max(nC, 1)); P.Sb = P.Sall; P.Sc = P.Sall + nV;
    DA(AC, nC); DA(posAC, nC); DA(pid, P.nblk_ratio); DA(res_id, 2);
    DA(R, nV + 1); DA(posR, nV + nC + 1);
    if ((e = hipMemsetAsync(P.res_id, 0, 2 * sizeof(int), stream)) != hipSuccess) return e;      // res_id[0]: ticket counter of k_ratio1
    DA(d_fpos, nV); DA(d_cand, nC); DA(d_freev, nV);
    DA(hinv, nV); DA(dflag, 4); DA(ps_u, P.nAmax + 4);
    DA(sym_part, 2 * (size_t)((P.nAmax + SYT - 1) / SYT + 1) * (size_t)std::max(P.nAmax, 1));
    P.lz_stride = Impl::pad16(nV + 4); DA(lz_vec, (size_t)LZK * P.lz_stride); DA(lz_c, LZK); DA(lz_d, LZK);
    DA(rs_p, nV + std::max(nC, 1)); P.rs_Ap = P.rs_p + nV;      // ([p; A p] contiguous: one product with the tableau refreshes both)
    DA(ra1, nV + 4); DA(ra2, nV + 4); DA(ra3, nV + 4); DA(ra4, nV + 4); DA(rs_dl, nV + 4); DA(rs_ps_u, nV + 4);
    DA(wz_part, 2 * (size_t)((nV + SYT - 1) / SYT + 1) * (size_t)nV);
#undef DA
    if ((e = rsqp_dense_work_alloc(&P.dw, nV)) != hipSuccess) return e;
    if ((e = hipHostMalloc(reinterpret_cast<void **>(&P.h_ctl), 64 * sizeof(double), hipHostMallocMapped)) != hipSuccess) return e;
    if ((e = hipHostGetDevicePointer(reinterpret_cast<void **>(&P.d_ctl), P.h_ctl, 0)) != hipSuccess) return e;
    if ((e = hipHostMalloc(reinterpret_cast<void **>(&P.h_pinned), 64 * sizeof(double))) != hipSuccess) return e;
    if ((e = hipHostMalloc(reinterpret_cast<void **>(&P.h_pinned_i), 64 * sizeof(int))) != hipSuccess) return e;
    P.hSb.assign(nV, 0); P.hSc.assign(nC, 0); P.hAC.assign(std::max(nC, 1), 0);
    P.hR.assign(nV + 1, 0); P.hposR.assign(nV + nC + 1, -1);
    return hipSuccess;
}

void RsqpLargeEngine::set_matrices(const RsqpLargeMatrices &m) { p_->M = m; p_->nnzA_ = (double)m.Annz; }

int RsqpLargeEngine::solve(int mode, const double *d_g, const double *d_lb, const double *d_ub, const double *d_lbA,
                           const double *d_ubA, int *nWSR, const double *h_x0, const double *h_y0, const int *h_guess_b) {
    Impl &P = *p_;
    const int nV = P.nV, nC = P.nC;
    hipStream_t st = P.st;
    P.err_ = hipSuccess;
    const int maxit = *nWSR;
    *nWSR = 0;
    const bool tstat = P.profile || getenv("RSQP_LARGE_WAITSTAT") != nullptr;
    const double t_solve0 = Impl::now_s();
    if (mode != RSQP_LMODE_COLD && P.status == QPS_NOTINITIALISED) mode = RSQP_LMODE_COLD;
    // targets, clamped to +-INFTY
    hipLaunchKernelGGL(k_copy, g1(nV), dim3(NT), 0, st, d_g, P.gN, nV);
    hipLaunchKernelGGL(k_clamp_copy, g1(nV), dim3(NT), 0, st, nV, d_lb, P.lbN);
    hipLaunchKernelGGL(k_clamp_copy, g1(nV), dim3(NT), 0, st, nV, d_ub, P.ubN);
    if (nC > 0) {
        hipLaunchKernelGGL(k_clamp_copy, g1(nC), dim3(NT), 0, st, nC, d_lbA, P.lbAN);
        hipLaunchKernelGGL(k_clamp_copy, g1(nC), dim3(NT), 0, st, nC, d_ubA, P.ubAN);
    }
    // consistency of the data + working-set guess need the targets on the host
    std::vector<double> hl(nV), hu(nV), hlA(nC), huA(nC);
    if (hipMemcpyAsync(hl.data(), P.lbN, 8 * nV, hipMemcpyDeviceToHost, st) != hipSuccess) return RET_SETUP_FAILED;
    if (hipMemcpyAsync(hu.data(), P.ubN, 8 * nV, hipMemcpyDeviceToHost, st) != hipSuccess) return RET_SETUP_FAILED;
    if (nC > 0) {
        if (hipMemcpyAsync(hlA.data(), P.lbAN, 8 * nC, hipMemcpyDeviceToHost, st) != hipSuccess) return RET_SETUP_FAILED;
        if (hipMemcpyAsync(huA.data(), P.ubAN, 8 * nC, hipMemcpyDeviceToHost, st) != hipSuccess) return RET_SETUP_FAILED;
    }
    if (hipStreamSynchronize(st) != hipSuccess) return RET_SETUP_FAILED;
    bool bad = false;
    for (int v = 0; v < nV; v++) bad |= hl[v] > hu[v] + RSQP_EPS;
    for (int i = 0; i < nC; i++) bad |= hlA[i] > huA[i] + RSQP_EPS;
    if (bad) {
        P.infeasible = 1; P.unbounded = 0;
        if (mode == RSQP_LMODE_COLD) {
            (void)hipMemsetAsync(P.x, 0, 8 * nV, st); (void)hipMemsetAsync(P.y, 0, 8 * (nV + nC), st);
            std::fill(P.hSb.begin(), P.hSb.end(), 0); std::fill(P.hSc.begin(), P.hSc.end(), 0);
            (void)hipMemsetAsync(P.Sb, 0, 4 * nV, st); (void)hipMemsetAsync(P.Sc, 0, 4 * std::max(nC, 1), st);
        }
        return RET_INFEASIBLE;
    }
    int rc = RET_OK;
    if (mode != RSQP_LMODE_HOT_VECTORS) {
        std::vector<int> gb(nV, 0), gc(nC, 0);
        std::vector<double> hx(nV, 0.0), hy(nV + nC, 0.0), hAx(nC, 0.0);
        const bool have_x0 = mode == RSQP_LMODE_HOT_MATRICES || (mode == RSQP_LMODE_WARM && h_x0);
        const bool have_y0 = mode == RSQP_LMODE_HOT_MATRICES || (mode == RSQP_LMODE_WARM && h_y0);
        if (mode == RSQP_LMODE_HOT_MATRICES) {
            if (hipMemcpy(hx.data(), P.x, 8 * nV, hipMemcpyDeviceToHost) != hipSuccess) return RET_SETUP_FAILED;
            if (hipMemcpy(hy.data(), P.y, 8 * (nV + nC), hipMemcpyDeviceToHost) != hipSuccess) return RET_SETUP_FAILED;
            gb = P.hSb; gc = P.hSc;
        } else if (mode == RSQP_LMODE_WARM) {
            if (h_x0) std::copy(h_x0, h_x0 + nV, hx.begin());
            if (h_y0) std::copy(h_y0, h_y0 + nV + nC, hy.begin());
        }
        if (hipMemcpyAsync(P.x, hx.data(), 8 * nV, hipMemcpyHostToDevice, st) != hipSuccess) return RET_SETUP_FAILED;
        if (hipMemcpyAsync(P.y, hy.data(), 8 * (nV + nC), hipMemcpyHostToDevice, st) != hipSuccess) return RET_SETUP_FAILED;
        if (mode != RSQP_LMODE_HOT_MATRICES) {
            // obtainAuxiliaryWorkingSet (see the CPU restatement): bounds
            for (int v = 0; v < nV; v++) {
                int s;
                if (mode == RSQP_LMODE_WARM && h_guess_b) s = h_guess_b[v];
                else if (have_x0) s = hx[v] <= hl[v] + RSQP_BOUND_TOLERANCE ? -1 : (hx[v] >= hu[v] - RSQP_BOUND_TOLERANCE ? 1 : 0);
                else if (have_y0) s = hy[v] > RSQP_EPS ? -1 : (hy[v] < -RSQP_EPS ? 1 : 0);
                else s = -1;
                if (s == -1 && hl[v] <= -RSQP_INFTY) s = (hu[v] < RSQP_INFTY && !have_x0 && !(mode == RSQP_LMODE_WARM && h_guess_b)) ? 1 : 0;
                if (s == 1 && hu[v] >= RSQP_INFTY) s = 0;
                gb[v] = s;
            }
            if (have_x0 && nC > 0) {   // constraints from A x0
                P.A_times(P.x, P.Ax);
                if (hipMemcpyAsync(hAx.data(), P.Ax, 8 * nC, hipMemcpyDeviceToHost, st) != hipSuccess) return RET_SETUP_FAILED;
                if (hipStreamSynchronize(st) != hipSuccess) return RET_SETUP_FAILED;
            }
            for (int i = 0; i < nC; i++) {
                int s = 0;
                // no guessed constraints in this call shape: sides from the signs of y0 (reinit_from_y0, opt-in)
                // or -- the default, as qpOASES does when x0 is given too -- only from where A x0 sits
                if (have_y0 && (!have_x0 || P.reinit_from_y0)) s = hy[nV + i] > RSQP_EPS ? -1 : (hy[nV + i] < -RSQP_EPS ? 1 : 0);
                else if (have_x0) s = hAx[i] <= hlA[i] + RSQP_BOUND_TOLERANCE ? -1 : (hAx[i] >= huA[i] - RSQP_BOUND_TOLERANCE ? 1 : 0);
                if (s == -1 && hlA[i] <= -RSQP_INFTY) s = 0;
                if (s == 1 && huA[i] >= RSQP_INFTY) s = 0;
                gc[i] = s;
            }
        }
        P.nflips = 0;
        const double t_prep0 = Impl::now_s();
        if (P.force_null_space) { P.dual = false; P.rsh = false; }      // (the retry behind a range-space path that failed numerically)
        else {
            bool ok = false;
            if (P.dual_prepare(&ok) != RET_OK) return RET_SETUP_FAILED;
            P.dual = ok;
            P.rsh = false;
            if (!ok) {
                bool ok2 = false;
                if (P.rs_prepare(&ok2) != RET_OK) return RET_SETUP_FAILED;
                P.rsh = ok2;
            }
        }
        const double t_prep1 = Impl::now_s();
        rc = P.setup_aux(gb, gc);
        if (tstat) { (void)hipStreamSynchronize(st); fprintf(stderr, "[rsqp profile] solve: data checks + guess %.4f s, H^-1 operator %.4f s, factors of the guess %.4f s\n", t_prep0 - t_solve0, t_prep1 - t_prep0, Impl::now_s() - t_prep1); }
        if (rc != RET_OK && mode != RSQP_LMODE_COLD) {   // fall back to a cold start
            (void)hipMemsetAsync(P.x, 0, 8 * nV, st); (void)hipMemsetAsync(P.y, 0, 8 * (nV + nC), st);
            for (int v = 0; v < nV; v++) gb[v] = hl[v] > -RSQP_INFTY ? -1 : (hu[v] < RSQP_INFTY ? 1 : 0);
            std::fill(gc.begin(), gc.end(), 0);
            rc = P.setup_aux(gb, gc);
        }
        if (rc != RET_OK) return rc;
    } else {
        P.infeasible = P.unbounded = 0;
    }
    int n = maxit;
    rc = P.homotopy(maxit, &n);
    *nWSR = n;
    (void)hipStreamSynchronize(st);
    if (P.rsh && P.rs_kind == 1) {       // a spin of the multi-workgroup banded product that ran out (never seen): nothing of this solve can be trusted
        int e = 0;
        if (hipMemcpy(&e, P.dflag + 1, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess || e != 0) { P.status = QPS_NOTINITIALISED; rc = RET_SETUP_FAILED; }
    }
    if (rc == RET_SETUP_FAILED && (P.rsh || P.dual) && !P.force_null_space && P.err_ == hipSuccess) {
        // A range-space path gave up on a pivot: the explicit inverse of C H^-1 C' squares the conditioning of the active rows, and on
        // the way to a vertex of an INFEASIBLE QP (multipliers growing without bound) that is where it ends -- seen on a 142 x 208
        // member of the randomised large-engine check, whose null-space run reports "infeasible", as the CPU restatement of qpOASES does. The null-space path
        // takes the solve over from a cold start; the changes of both attempts are counted.
        P.force_null_space = true;
        int n2 = maxit;
        rc = solve(RSQP_LMODE_COLD, d_g, d_lb, d_ub, d_lbA, d_ubA, &n2, nullptr, nullptr, nullptr);
        P.force_null_space = false;
        *nWSR = n + n2;
    }
    return rc;
}

const double *RsqpLargeEngine::d_x() const { return p_->x; }
const double *RsqpLargeEngine::d_y() const { return p_->y; }
const int *RsqpLargeEngine::d_Sb() const { return p_->Sb; }
const int *RsqpLargeEngine::d_Sc() const { return p_->Sc; }
int RsqpLargeEngine::status_word() const {
    return p_->infeasible ? 100 + p_->status : (p_->unbounded ? 200 + p_->status : p_->status);
}
int RsqpLargeEngine::nflips() const { return p_->nflips; }
int RsqpLargeEngine::path() const { return p_->rsh ? 1 + p_->rs_kind : (p_->dual ? 1 : 0); }      // (4: dense H^-1 + tableau)
hipError_t RsqpLargeEngine::last_error() const { return p_->err_; }
const char *RsqpLargeEngine::profile_name(int k) {
    static const char *nm[PROFILE_CLASSES] = {"gemv_n", "gemv_t", "ger", "wz_shrink", "wz_grow", "spmv", "ger_gemv_t", "shrink_gemv", "ger_gemv_n", "band_hinv"};
    return k >= 0 && k < PROFILE_CLASSES ? nm[k] : "";
}
void RsqpLargeEngine::profile_enable(bool on) { p_->profile = on; }
int RsqpLargeEngine::setup_profile(double *out8) const {
    const Impl::SetupStat &t = p_->setup_stat;
    if (!t.valid) return 0;
    out8[0] = t.m; out8[1] = t.n; out8[2] = t.nZ; out8[3] = t.ms_tq; out8[4] = t.ms_wz; out8[5] = t.flops_tq; out8[6] = t.flops_wz; out8[7] = t.dual;
    return 1;
}
// tuning aid: device time per call of one product / update kernel class on an nrows x ncols matrix (the engine's Z
// buffer, leading dimension nV; contents are whatever the buffer holds -- the result is not looked at)
int RsqpLargeEngine::time_kernel(int kind, int nrows, int ncols, int reps, float *ms) {
    Impl &P = *p_;
    if (nrows <= 0 || ncols <= 0 || nrows > P.nV || ncols > P.nV || reps <= 0 || kind < 0 || kind > 2) return -1;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1;
    (void)hipMemsetAsync(P.Z, 0, sizeof(double) * (size_t)P.ld * P.nV, P.st);
    P.fill(P.w1, P.nV, 1.0); P.fill(P.w2, P.nV, 0.5);
    for (int pass = 0; pass < 2; pass++) {            // pass 0 warms up
        (void)hipEventRecord(e0, P.st);
        for (int r = 0; r < (pass ? reps : 2); r++) {
            if (kind == 0) P.gemv_n(P.Z, P.ld, nrows, ncols, P.w1, 1.0, 0.0, nullptr, P.w3);
            else if (kind == 1) P.gemv_t(P.Z, P.ld, nrows, ncols, P.w1, P.w3);
            else P.ger(P.Z, P.ld, nrows, ncols, P.w1, P.w2, 0, 1e-3);
        }
        (void)hipEventRecord(e1, P.st);
        if (hipEventSynchronize(e1) != hipSuccess) return -1;
    }
    float t = 0.f;
    (void)hipEventElapsedTime(&t, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms = t / reps;
    return P.err_ == hipSuccess ? 0 : -1;
}
void RsqpLargeEngine::set_reinit_from_y0(bool on) { p_->reinit_from_y0 = on; }
void RsqpLargeEngine::profile_get(double *out) const {
    for (int k = 0; k < PROFILE_CLASSES; k++) {
        out[4 * k] = (double)p_->prof[k].calls; out[4 * k + 1] = p_->prof[k].ms; out[4 * k + 2] = p_->prof[k].bytes; out[4 * k + 3] = 0.0;
    }
}
double RsqpLargeEngine::objective() {
    Impl &P = *p_;
    P.H_times(P.x, P.w2);
    P.dot(P.x, P.w2, P.nV, 20);
    P.dot(P.gN, P.x, P.nV, 21);
    P.dot(P.x, P.x, P.nV, 22);
    double r[3] = {0, 0, 0};
    P.read_scal(20, 3, r);
    return 0.5 * (r[0] - P.M.hreg * r[2]) + r[1];
}
