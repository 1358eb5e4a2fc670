// sparse.hip -- HBM-bound sparse kernels of the QP path for gfx950.
//
//   * csx_stream_spmv : out[major] = sum_k val[k] * in[idx[k]]  over a compressed-major
//     matrix. On the CSC arrays of A it is SpHbMat::transposed_times (A'y, reference
//     src/SpHbMat.cpp:659-696); on the CSR copy of A, or on the symmetric H, it is
//     SpHbMat::times (A x / H x, :698-737). LDS-staged "stream" form: a workgroup streams a
//     contiguous chunk of (val, idx) with fully coalesced loads, multiplies with the
//     gathered input (the input vector is L2-resident), parks the products in LDS and
//     then reduces each short segment from LDS -- no atomics, fixed summation order
//     (entry order inside a segment, exactly the order of the reference loops).
//     Batched over blockIdx.y for independent matrices (one QP each).
//   * scatter_values : SpHbMat::setMatVal (src/SpHbMat.cpp:368-393) -- value refresh of
//     the device CSC through the permutation `order`, plus the gather that refreshes
//     the CSR copy.
//   * kkt_* : fused qpOASESInterface::get_working_set (+ the A x product) and
//     ::test_optimality (src/qpOASESInterface.cpp:498-684, 835-895).
#include "rsqp_sparse.h"

namespace {

constexpr int SPMV_NT = 256;
constexpr int SPMV_CHUNK = 2048;  // entries staged per workgroup: 16 KiB of LDS

__global__ void __launch_bounds__(SPMV_NT)
csx_stream_spmv(const int *__restrict__ blk, const int *__restrict__ ptr, const int *__restrict__ idx,
                const double *__restrict__ val, const double *__restrict__ in, double *__restrict__ out,
                long long ptr_stride, long long nnz_stride, long long in_stride, long long out_stride) {
    __shared__ double prod[SPMV_CHUNK];
    const int m = blockIdx.y;
    ptr += m * ptr_stride; idx += m * nnz_stride; val += m * nnz_stride;
    in += m * in_stride; out += m * out_stride;
    const int r0 = blk[blockIdx.x], r1 = blk[blockIdx.x + 1];
    const int k0 = ptr[r0], k1 = ptr[r1];
    if (k1 - k0 <= SPMV_CHUNK) {
        for (int k = k0 + threadIdx.x; k < k1; k += SPMV_NT) prod[k - k0] = val[k] * in[idx[k]];
        __syncthreads();
        for (int r = r0 + threadIdx.x; r < r1; r += SPMV_NT) {
            double s = 0.0;
            const int a = ptr[r] - k0, b = ptr[r + 1] - k0;
            for (int k = a; k < b; k++) s += prod[k];
            out[r] = s;
        }
    } else {
        // a single long segment (the block builder never mixes it with others):
        // every lane accumulates a strided slice, fixed-order tree afterwards
        double s = 0.0;
        for (int k = k0 + threadIdx.x; k < k1; k += SPMV_NT) s += val[k] * in[idx[k]];
        prod[threadIdx.x] = s;
        __syncthreads();
        for (int o = SPMV_NT / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) prod[threadIdx.x] += prod[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[r0] = prod[0];
    }
}

__global__ void scatter_values(int n, const int *__restrict__ order, const int *__restrict__ tmap,
                               const double *__restrict__ tv, double *__restrict__ val) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) val[order[i]] = tv[tmap ? tmap[i] : i];
}

__global__ void gather_values(int n, const int *__restrict__ perm, const double *__restrict__ src,
                              double *__restrict__ dst) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}

// ---------------------------------------------------------------------------------
// KKT certificate. One workgroup of 256 threads per QP; products come from Ax / ATy / Hx
// computed beforehand (large QPs: csx_stream_spmv) or inside (small QPs, batched).
// Sums use a fixed tree => run-to-run deterministic.
// ---------------------------------------------------------------------------------
constexpr int KKT_NT = 256;

__device__ inline double block_sum_256(double v, double *sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__device__ inline int map_bound(int ws, double x, double lb, double ub) {
    // src/qpOASESInterface.cpp:846-868
    if (ws == 1) return fabs(x - lb) < 1.0e-8 ? RSQP_K_BOTH : RSQP_K_ABOVE;
    if (ws == -1) return fabs(x - ub) < 1.0e-8 ? RSQP_K_BOTH : RSQP_K_BELOW;
    return ws == 0 ? RSQP_K_INACTIVE : RSQP_K_INVALID;
}
__device__ inline int map_constr(int ws, double Ax, double lbA, double ubA) {
    // src/qpOASESInterface.cpp:871-892 -- fabs() wraps the comparison there, so the
    // test is the SIGNED one
    if (ws == 1) return (Ax - lbA < 1.0e-8) ? RSQP_K_BOTH : RSQP_K_ABOVE;
    if (ws == -1) return (Ax - ubA < 1.0e-8) ? RSQP_K_BOTH : RSQP_K_BELOW;
    return ws == 0 ? RSQP_K_INACTIVE : RSQP_K_INVALID;
}

__device__ inline void kkt_terms(int W, double yv, double val, double lo, double hi, double &dual,
                                 double &compl_, int &bad) {
    // dual feasibility :533-578 and complementarity :611-658
    switch (W) {
    case RSQP_K_INACTIVE: dual += fabs(yv); compl_ += fabs(yv); break;
    case RSQP_K_BELOW: dual += -fmin(0.0, yv); compl_ += fabs(yv * (val - lo)); break;
    case RSQP_K_ABOVE: dual += fmax(0.0, yv); compl_ += fabs(yv * (hi - val)); break;
    case RSQP_K_BOTH: break;
    default: bad = 1;
    }
}

__global__ void __launch_bounds__(KKT_NT)
kkt_kernel(RsqpKktArgs a) {
    __shared__ double sh[4];
    const int q = blockIdx.x;
    const int nV = a.nV ? a.nV[q] : a.nV1, nC = a.nC ? a.nC[q] : a.nC1;
    const long long oV = a.offV ? a.offV[q] : 0, oC = a.offC ? a.offC[q] : 0;
    const double *x = a.x + oV, *y = a.y + oV + oC, *g = a.g + oV, *lb = a.lb + oV, *ub = a.ub + oV;
    const double *lbA = a.lbA + oC, *ubA = a.ubA + oC, *Ax = a.Ax + oC, *ATy = a.ATy + oV, *Hx = a.Hx + oV;
    const int *wsb = a.ws_b + oV, *wsc = a.ws_c + oC;
    int *Wb = a.W_b + oV, *Wc = a.W_c + oC;
    double primal = 0.0, dual = 0.0, compl_ = 0.0, stat = 0.0;
    int bad = 0;
    for (int v = threadIdx.x; v < nV; v += KKT_NT) {
        double xv = x[v], l = fmax(lb[v], -RSQP_K_INFTY), u = fmin(ub[v], RSQP_K_INFTY), yv = y[v];
        int W = map_bound(wsb[v], xv, l, u);
        Wb[v] = W;
        primal += fmax(0.0, l - xv) + -fmin(0.0, u - xv);            // :518-521
        kkt_terms(W, yv, xv, l, u, dual, compl_, bad);
        stat += fabs(ATy[v] + yv - g[v] - Hx[v]);                       // :595-604
    }
    for (int i = threadIdx.x; i < nC; i += KKT_NT) {
        double ax = Ax[i], l = fmax(lbA[i], -RSQP_K_INFTY), u = fmin(ubA[i], RSQP_K_INFTY), yv = y[nV + i];
        int W = map_constr(wsc[i], ax, l, u);
        Wc[i] = W;
        primal += fmax(0.0, l - ax) + -fmin(0.0, u - ax);            // :524-527
        kkt_terms(W, yv, ax, l, u, dual, compl_, bad);
    }
    primal = block_sum_256(primal, sh);
    dual = block_sum_256(dual, sh);
    compl_ = block_sum_256(compl_, sh);
    stat = block_sum_256(stat, sh);
    double fb = block_sum_256((double)bad, sh);
    if (threadIdx.x == 0) {
        double *o = a.out + 6LL * q;
        o[0] = primal; o[1] = dual; o[2] = compl_; o[3] = stat;
        o[4] = compl_ + stat + dual + primal;   // :664-665
        o[5] = fb;
    }
}

// products for a batch of SMALL problems: one workgroup per problem, lane per row/column
__global__ void __launch_bounds__(KKT_NT)
small_products_kernel(const QPDesc *desc, const int *Ajc, const int *Air, const double *Aval,
                      const int *Arp, const int *Aci, const double *Arv, const int *Hjc, const int *Hir,
                      const double *Hval, const double *x, const double *y, double *Ax, double *ATy,
                      double *Hx) {
    const QPDesc d = desc[blockIdx.x];
    const double *xq = x + d.offV, *yc = y + d.offV + d.offC + d.nV;
    for (int r = threadIdx.x; r < d.nC; r += KKT_NT) {
        const int *rp = Arp + d.offArp;
        double s = 0.0;
        for (int k = rp[r]; k < rp[r + 1]; k++) s += Arv[d.offAnz + k] * xq[Aci[d.offAnz + k]];
        Ax[d.offC + r] = s;
    }
    for (int c = threadIdx.x; c < d.nV; c += KKT_NT) {
        const int *jc = Ajc + d.offAjc;
        double s = 0.0;
        for (int k = jc[c]; k < jc[c + 1]; k++) s += Aval[d.offAnz + k] * yc[Air[d.offAnz + k]];
        ATy[d.offV + c] = s;
        double h = 0.0;
        if (d.haveH) {
            const int *hj = Hjc + d.offHjc;
            for (int k = hj[c]; k < hj[c + 1]; k++) h += Hval[d.offHnz + k] * xq[Hir[d.offHnz + k]];
        }
        Hx[d.offV + c] = h;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------
hipError_t rsqp_launch_spmv(const int *blk, int nblk, const int *ptr, const int *idx, const double *val,
                            const double *in, double *out, int nbatch, long long ptr_stride,
                            long long nnz_stride, long long in_stride, long long out_stride,
                            hipStream_t stream) {
    if (nblk <= 0 || nbatch <= 0) return hipSuccess;
    hipLaunchKernelGGL(csx_stream_spmv, dim3(nblk, nbatch), dim3(SPMV_NT), 0, stream, blk, ptr, idx, val, in,
                       out, ptr_stride, nnz_stride, in_stride, out_stride);
    return hipGetLastError();
}

int rsqp_spmv_chunk(void) { return SPMV_CHUNK; }

hipError_t rsqp_launch_scatter(int n, const int *order, const int *tmap, const double *tv, double *val,
                               hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_values, dim3((n + 255) / 256), dim3(256), 0, stream, n, order, tmap, tv, val);
    return hipGetLastError();
}

hipError_t rsqp_launch_gather(int n, const int *perm, const double *src, double *dst, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_values, dim3((n + 255) / 256), dim3(256), 0, stream, n, perm, src, dst);
    return hipGetLastError();
}

hipError_t rsqp_launch_kkt(const RsqpKktArgs &a, int nq, hipStream_t stream) {
    hipLaunchKernelGGL(kkt_kernel, dim3(nq), dim3(KKT_NT), 0, stream, a);
    return hipGetLastError();
}

hipError_t rsqp_launch_small_products(const QPPools &p, int nq, double *Ax, double *ATy, double *Hx,
                                      hipStream_t stream) {
    hipLaunchKernelGGL(small_products_kernel, dim3(nq), dim3(KKT_NT), 0, stream, p.desc, p.Ajc, p.Air, p.Aval,
                       p.Arp, p.Aci, p.Arv, p.Hjc, p.Hir, p.Hval, p.x, p.y, Ax, ATy, Hx);
    return hipGetLastError();
}
