// qp_rs_kernels.h -- kernels of the GENERAL range-space path of the HBM-resident engine (qp_large.hip, Impl::rsh):
// any symmetric positive definite Hessian. Included inside qp_large.hip's anonymous namespace (it uses lane_sum4, block_sum,
// publish, delta_of, NT, SYT of that file).
//
// Formulation. The active BOUNDS and the active CONSTRAINTS are both rows of one matrix C (row id v < nV: e_v', the bound of
// variable v; id nV + i: row i of A), and the engine keeps the explicit inverse of the Schur complement
//
//     S = C H^-1 C'   (nR x nR, nR = fixed variables + active constraints),        Sinv = S^-1, upper triangle (k_sym_tile)
//
// H never changes during a solve, so H^-1 is an OPERATOR built once per Hessian: the LDL' factor of a banded H (half bandwidth
// <= 2) applied by ONE workgroup in chunked form (k_band_apply), or the explicit dense inverse (GEMV). Every working-set change is
// one of TWO operations -- a row joins (bordering of Sinv) or leaves (rank-1 elimination) -- whatever it is a row of.
//
//     step direction:  S dl = db + C p  (p = H^-1 (gN - g)),   q = C'dl,   dx = H^-1 (q - (gN - g)),   H dx = q - (gN - g)
//     row c joins:     w = H^-1 c',  cv = C w,  u = Sinv cv,  s = c w - cv'u;  Sinv <- [[Sinv + u u'/s, -u/s], [-u'/s, 1/s]]
//     row k leaves:    Sinv -= v v'/v_k (v = column k), the last row / column moves into the slot
//
// Reference: the arithmetic stands in for qpOASES' SQProblem::hotstart / init behind src/qpOASESInterface.cpp:155-206.
#pragma once

// ---- banded H^-1 (half bandwidth <= 2): x = L^-T D^-1 L^-1 b, L unit lower with sub-diagonals l1 (i, i - 1) and l2 (i, i - 2) ------
// The two triangular solves are linear recurrences of depth 2. Rows are cut into T chunks of c rows; thread t < T solves chunk t
// with zero incoming state (sequential over c rows, the vector in LDS, the factor read in a chunk-interleaved layout so that the
// T threads load consecutive addresses), one thread then chains the T outgoing states through the precomputed homogeneous
// solutions, and all lanes add `state x homogeneous solution` to their rows. ~2 (c + T) dependent steps instead of 2 n.
struct BandOp {
    int n, T, c;                  // rows, chunks, rows per chunk (T * c >= n)
    const double *l1i, *l2i;      // forward factor entries, interleaved: l1i[k * T + t] = l1[t * c + k] (0 beyond n)
    const double *u1i, *u2i;      // backward: u1[i] = l1[i + 1], u2[i] = l2[i + 2], interleaved the same way
    const double *ga, *gb;        // forward homogeneous solutions per row (state (1, 0) / (0, 1) ahead of the row's chunk)
    const double *ha, *hb;        // backward homogeneous solutions per row (state behind the row's chunk)
    const double *dinv;           // 1 / d
};
constexpr int BAND_MAX_N = 16384;         // the vector lives in LDS (128 KB)
constexpr int BAND_T = 128;
__device__ __forceinline__ void band_apply_body(const BandOp &op, const double *__restrict__ in, const double *__restrict__ sub,
                                                double *__restrict__ out, double *v, double (*st)[2], double (*PE)[6]) {
    const int n = op.n, T = op.T, c = op.c, tid = threadIdx.x, nth = blockDim.x;
    for (int i = tid; i < n; i += nth) v[i] = sub ? in[i] - sub[i] : in[i];
    __syncthreads();
    // forward, chunk by chunk with zero incoming state; then the chunk's affine map "incoming state -> outgoing state"
    // (z[e-1], z[e-2]) = E + P (z[s-1], z[s-2]) goes to LDS so that the chaining thread reads no global memory
    if (tid < T) {
        const int s0 = tid * c, e = min(s0 + c, n);
        double z1 = 0.0, z2 = 0.0;
        for (int i = s0; i < e; i++) {
            const int k = i - s0;
            const double z = v[i] - op.l1i[k * T + tid] * z1 - op.l2i[k * T + tid] * z2;
            v[i] = z; z2 = z1; z1 = z;
        }
        double *q = PE[tid];
        if (e <= s0) { q[0] = 0.0; q[1] = 1.0; q[2] = 0.0; q[3] = 0.0; q[4] = 0.0; q[5] = 1.0; }
        else {
            q[0] = z1; q[1] = op.ga[e - 1]; q[2] = op.gb[e - 1];
            if (e - 2 >= s0) { q[3] = z2; q[4] = op.ga[e - 2]; q[5] = op.gb[e - 2]; }
            else { q[3] = 0.0; q[4] = 1.0; q[5] = 0.0; }
        }
    }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0;          // state entering chunk t: (z[s - 1], z[s - 2])
        for (int t = 0; t < T; t++) {
            st[t][0] = a; st[t][1] = b;
            const double *q = PE[t];
            const double na = q[0] + q[1] * a + q[2] * b, nb = q[3] + q[4] * a + q[5] * b;
            a = na; b = nb;
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += nth) { const int t = i / c; v[i] = (v[i] + op.ga[i] * st[t][0] + op.gb[i] * st[t][1]) * op.dinv[i]; }
    __syncthreads();
    // backward: x[i] = y[i] - u1[i] x[i+1] - u2[i] x[i+2]; a chunk's map (x[s], x[s+1]) = E + P (x[e], x[e+1])
    if (tid < T) {
        const int s0 = tid * c, e = min(s0 + c, n);
        double x1 = 0.0, x2 = 0.0;
        for (int i = e - 1; i >= s0; i--) {
            const int k = i - s0;
            const double x = v[i] - op.u1i[k * T + tid] * x1 - op.u2i[k * T + tid] * x2;
            v[i] = x; x2 = x1; x1 = x;
        }
        double *q = PE[tid];
        if (e <= s0) { q[0] = 0.0; q[1] = 1.0; q[2] = 0.0; q[3] = 0.0; q[4] = 0.0; q[5] = 1.0; }
        else {
            q[0] = x1; q[1] = op.ha[s0]; q[2] = op.hb[s0];
            if (s0 + 1 < e) { q[3] = x2; q[4] = op.ha[s0 + 1]; q[5] = op.hb[s0 + 1]; }
            else { q[3] = 0.0; q[4] = 1.0; q[5] = 0.0; }
        }
    }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0;          // state entering chunk t from behind: (x[e], x[e + 1])
        for (int t = T - 1; t >= 0; t--) {
            st[t][0] = a; st[t][1] = b;
            const double *q = PE[t];
            const double na = q[0] + q[1] * a + q[2] * b, nb = q[3] + q[4] * a + q[5] * b;
            a = na; b = nb;
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += nth) { const int t = i / c; out[i] = v[i] + op.ha[i] * st[t][0] + op.hb[i] * st[t][1]; }
}
// out = H^-1 (in - sub)   (sub may be null; in == out allowed). One workgroup; column blockIdx.x of a matrix when ld != 0.
// mSb / mfix: optional epilogue of the step direction -- out[v] = mfix[v] where mSb[v] != 0 (dx exactly on its bound's move)
__global__ void __launch_bounds__(1024) k_band_apply(BandOp op, const double *__restrict__ in, const double *__restrict__ sub,
                                                     double *__restrict__ out, long long ld, const int *__restrict__ mSb,
                                                     const double *__restrict__ mfix) {
    extern __shared__ double band_lds[];
    double *v = band_lds;
    __shared__ double st[BAND_T][2], PE[BAND_T][6];
    const long long off = (long long)blockIdx.x * ld;
    band_apply_body(op, in + off, sub, out + off, v, st, PE);
    if (mSb) {
        __syncthreads();
        for (int i = threadIdx.x; i < op.n; i += blockDim.x) if (mSb[i] != 0) out[i] = mfix[i];
    }
}

// ---- rows of C ----------------------------------------------------------------------------------------------------------------
// the incoming row as a dense vector: id < nV: e_id, else row id - nV of A (all variables). One workgroup.
__global__ void __launch_bounds__(NT) k_rs_row(int nV, int id, const int *__restrict__ rp, const int *__restrict__ ci,
                                               const double *__restrict__ rv, const double *__restrict__ denseAT, double *__restrict__ a) {
    if (id >= nV && denseAT) {
        const double *row = denseAT + (long long)(id - nV) * nV;
        for (int v = threadIdx.x; v < nV; v += NT) a[v] = row[v];
        return;
    }
    for (int v = threadIdx.x; v < nV; v += NT) a[v] = 0.0;
    __syncthreads();
    if (id < nV) { if (threadIdx.x == 0) a[id] = 1.0; return; }
    const int r = id - nV;
    for (int k = rp[r] + threadIdx.x; k < rp[r + 1]; k += NT) a[ci[k]] = rv[k];
}
// w = Hinv c' for the explicit dense inverse: a bound picks column v, a sparse row combines the columns of its entries
__global__ void __launch_bounds__(NT) k_rs_hinv_row(int nV, int id, const int *__restrict__ rp, const int *__restrict__ ci,
                                                    const double *__restrict__ rv, const double *__restrict__ Hinv, long long ld,
                                                    double *__restrict__ w) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= nV) return;
    if (id < nV) { w[i] = Hinv[(long long)id * ld + i]; return; }
    const int r = id - nV;
    double s = 0.0;
    for (int k = rp[r]; k < rp[r + 1]; k++) s += rv[k] * Hinv[(long long)ci[k] * ld + i];
    w[i] = s;
}
// cv[j] = (C w)[j]: w itself on a bound row, (A w) on a constraint row
__global__ void k_rs_gather(int nR, const int *__restrict__ R, int nV, const double *__restrict__ w, const double *__restrict__ Aw,
                            double *__restrict__ cv) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nR) return;
    const int id = R[j];
    cv[j] = id < nV ? w[id] : Aw[id - nV];
}
// the reverse: a vector over the active rows back to [variables; constraints] (entries of inactive rows are left alone)
__global__ void k_rs_scatter(int nR, const int *__restrict__ R, const double *__restrict__ act, double *__restrict__ full) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nR) full[R[j]] = act[j];
}
// ... and only the constraint rows, by constraint index (the exchange rule's xi_C)
__global__ void k_rs_scatter_c(int nR, const int *__restrict__ R, int nV, const double *__restrict__ act, double *__restrict__ byc) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nR && R[j] >= nV) byc[R[j] - nV] = act[j];
}
// independence of an incoming row c from the active ones. Stage 1 (atu == null): a2 = |c_FR|^2, ad = c w (w = H^-1 c'),
// s = ad - cv'u the pivot of the bordering. Stage 2 (atu = A_AC' u_C): r2 = |(c - atu)_FR|^2 as well -- the first-order residual
// of the row's representation by the active rows (Z'r = Z'c: the test of the other engines). One workgroup of 1024; published.
__global__ void __launch_bounds__(1024) k_rs_li_publish(int nV, const int *__restrict__ Sb, const double *__restrict__ a,
                                                        const double *__restrict__ w, const double *__restrict__ atu, int k,
                                                        const double *__restrict__ cv, const double *__restrict__ u,
                                                        double *__restrict__ scal, double *__restrict__ ctl, double seqv) {
    __shared__ double sh[4][16];
    double a2 = 0.0, r2 = 0.0, ad = 0.0, cu = 0.0;
    for (int i = threadIdx.x; i < nV; i += 1024) {
        const double x0 = a[i];
        ad += x0 * w[i];
        if (Sb[i] == 0) { a2 += x0 * x0; if (atu) { const double q0 = x0 - atu[i]; r2 += q0 * q0; } }
    }
    for (int j = threadIdx.x; j < k; j += 1024) cu += cv[j] * u[j];
    double v4[4] = {a2, r2, ad, cu};
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
    for (int q = 0; q < 4; q++) { double v = 0.0; for (int x = 0; x < 16; x++) v += sh[q][x]; t[q] = v; }
    const double sp = t[2] - t[3];
    scal[6] = t[0]; scal[7] = t[1]; scal[5] = sp; scal[8] = sp != 0.0 ? 1.0 / sp : 0.0;
    ctl[2] = t[0]; ctl[3] = t[1]; ctl[4] = sp; ctl[5] = t[2];
    publish(ctl, seqv);
}

// ---- step direction -------------------------------------------------------------------------------------------------------------
// right-hand side of S dl = db + C p over the active rows
__global__ void k_rs_rhs(int nR, const int *__restrict__ R, int nV, const int *__restrict__ Sall, const double *__restrict__ lb,
                         const double *__restrict__ ub, const double *__restrict__ lbN, const double *__restrict__ ubN,
                         const double *__restrict__ lbA, const double *__restrict__ ubA, const double *__restrict__ lbAN,
                         const double *__restrict__ ubAN, const double *__restrict__ p, const double *__restrict__ Ap,
                         double *__restrict__ rhs) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nR) return;
    const int id = R[j];
    if (id < nV) rhs[j] = (Sall[id] == -1 ? delta_of(lbN[id], lb[id]) : delta_of(ubN[id], ub[id])) + p[id];
    else { const int r = id - nV; rhs[j] = (Sall[id] == -1 ? delta_of(lbAN[r], lbA[r]) : delta_of(ubAN[r], ubA[r])) + Ap[r]; }
}
// q = A'dl_C + dl_B;  H dx = q - (gN - g)   (dx = H^-1 of it follows)
// (dfix: the moves of the fixed variables, which dx holds on entry, are kept for the epilogue of the H^-1 product)
__global__ void k_rs_q(int nV, const int *__restrict__ Sb, const double *__restrict__ ATdy, const double *__restrict__ dy,
                       const double *__restrict__ gN, const double *__restrict__ g, double *__restrict__ Hdx,
                       const double *__restrict__ dx, double *__restrict__ dfix) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nV) return;
    Hdx[i] = (ATdy[i] + (Sb[i] != 0 ? dy[i] : 0.0)) - (gN[i] - g[i]);
    dfix[i] = dx[i];
}
// dx on the fixed variables exactly on their bound's move (the H^-1 product gives it up to rounding)
__global__ void k_rs_fix_dx(int nV, const int *__restrict__ Sb, const double *__restrict__ dfix, double *__restrict__ dx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nV && Sb[i] != 0) dx[i] = dfix[i];
}
// p = H^-1 (gN - g) and A p shrink with the homotopy step like gN - g itself (tau from the device copy of the ratio test)
__global__ void k_rs_scale_p(int nV, int nC, const double *__restrict__ res, double *__restrict__ p, double *__restrict__ Ap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int bid = (int)res[1];
    if (bid == 0x7fffffff) return;
    const double om = 1.0 - res[0];
    if (i < nV) p[i] *= om;
    if (i < nC) Ap[i] *= om;
}
__global__ void k_rs_diff(int n, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] - b[i];
}

// ---- set-up ---------------------------------------------------------------------------------------------------------------------
// Cd[:, j] = row R[j] of C as a dense column (columns zero-filled beforehand)
__global__ void k_rs_build_C(int nV, const int *__restrict__ R, const int *__restrict__ rp, const int *__restrict__ ci,
                             const double *__restrict__ rv, double *__restrict__ Cd, long long ld) {
    const int id = R[blockIdx.x];
    double *col = Cd + (long long)blockIdx.x * ld;
    if (id < nV) { if (threadIdx.x == 0) col[id] = 1.0; return; }
    const int r = id - nV;
    for (int k = rp[r] + threadIdx.x; k < rp[r + 1]; k += blockDim.x) col[ci[k]] = rv[k];
}
// Sinv of the working set "every variable fixed, no constraint": S = H^-1, so Sinv = H itself (upper triangle, dense; zero-filled
// beforehand). One thread per stored entry of the CSC.
__global__ void k_rs_sinv_from_H(int nV, const int *__restrict__ jc, const int *__restrict__ ir, const double *__restrict__ val,
                                 double hreg, double *__restrict__ Sinv, long long ld) {
    const int c = blockIdx.x;
    for (int k = jc[c] + threadIdx.x; k < jc[c + 1]; k += blockDim.x) {
        const int r = ir[k];
        if (r <= c) Sinv[(long long)c * ld + r] = val[k] + (r == c ? hreg : 0.0);
    }
}
__global__ void k_rs_iota(int n, int *__restrict__ R, int *__restrict__ pos) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { R[i] = i; pos[i] = i; }
}
