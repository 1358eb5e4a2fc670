// qp_small_x.h -- included by qp_small.hip inside its anonymous namespace.
//
// EngineX: the LDS-resident engine in the EXPLICIT-INVERSE formulation of qp_large.hip
// (orthonormal Z / Y, Minv = (A_AC,FR Y)^-1, Wz = (Z'HZ)^-1, one Householder reflection aimed at
// the last column per working-set change, Sherman-Morrison / bordering updates of the inverses,
// every solve a GEMV). Same homotopy, ratio tests, tie breaks and guards as Engine (the Givens /
// TQ formulation) and the CPU restatement -- step directions do not depend on the bases -- but nothing in
// an iteration is a sequential chain of length nZ: a Givens sweep or a triangular solve costs
// ~300 cycles of LDS round trips per step, a GEMV of the same size a handful of pipelined loads
// per lane. Used for problems with more than a few variables, where those chains dominate.
// Formulas: see the kernels of qp_large.hip (k_house, k_wz_*, k_minv_border, k_house_unit,
// k_house_free, k_sm_coef), which this file restates for one wave working in LDS.

// ---- products of the several-waves-per-problem build, as real (non-inlined) functions: the explicit-inverse engine
// issues ~30 matrix-vector products per working-set change; inlined, the four-wave kernel was 260 KB of code -- four
// times the 64 KB instruction cache two CUs share -- and ran at the speed of its instruction fetches.
#ifndef RSQP_XINLINE
#define RSQP_XINLINE __forceinline__
#endif
#define WSYNC()                                                  \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)
// a / b for 0 <= a < 4096, 1 <= b <= 256 without the ~40-instruction integer-division sequence (there is no
// hardware divide; the lane -> (output, slice) maps below are evaluated on every product): (a + 0.5) / b is at
// least 0.5 / 256 away from an integer, far beyond the error of a float reciprocal
__device__ __forceinline__ int fdiv_small(int a, int b) { return (int)(((float)a + 0.5f) * __frcp_rn((float)b)); }
// dot over a contiguous slice: sum_{j in [j0, j1)} a[j * sa] * b[j], four independent partial sums. UNIT: sa == 1 at
// compile time -- both operands are then contiguous, the eight loads of a trip sit at immediate offsets of one address
// and pair up into ds_read2_b64 (4 LDS instructions + 2 FMA per two terms instead of 4 loads + address arithmetic)
template <bool UNIT>
__device__ __forceinline__ double xdots(const ldouble *a, int sa, const ldouble *b, int j0, int j1) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const int da = UNIT ? 1 : sa;
    const ldouble *pa = a + j0 * da, *pb = b + j0;
    int left = j1 - j0;
    for (; left >= 8; left -= 8) {
        double m[8], x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { m[u] = pa[u * da]; x[u] = pb[u]; }
        s0 += m[0] * x[0]; s1 += m[1] * x[1]; s2 += m[2] * x[2]; s3 += m[3] * x[3];
        s0 += m[4] * x[4]; s1 += m[5] * x[5]; s2 += m[6] * x[6]; s3 += m[7] * x[7];
        pa += 8 * da; pb += 8;
    }
    if (left >= 4) {
        double m[4], x[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { m[u] = pa[u * da]; x[u] = pb[u]; }
        s0 += m[0] * x[0]; s1 += m[1] * x[1]; s2 += m[2] * x[2]; s3 += m[3] * x[3];
        pa += 4 * da; pb += 4; left -= 4;
    }
    for (; left > 0; left--) { s0 += pa[0] * pb[0]; pa += da; pb += 1; }
    return (s0 + s1) + (s2 + s3);
}
// The outputs of a product are dealt to the waves [w0, w0 + nw) in runs of opw = ceil(nout / nw); inside a wave
// P = 64 / opw lanes share an output, each summing a contiguous slice of the inner dimension; the slices meet in
// `part` behind a wave-scope fence (LDS operations of one wave execute in order) and are added in slice order
// by the first lane of the output. Waves outside [w0, w0 + nw) skip the call, so independent products given
// disjoint wave ranges run side by side; the caller closes the group with ONE workgroup barrier.
// TR: out[c] = sum_r M[c*l + r] xv[r] (nout = ncols, inner = nrows); else out[r] = beta base[r] + alpha sum_c M[c*l + r] xv[c]
// EPI: element-wise work that rides on the product -- called as epi(index, value) by the lane that stores out[index]
// (what used to be a separate loop + barrier after the product: a barrier-separated stage costs ~1 k cycles here
// whatever it computes)
struct XNoEpi { __device__ __forceinline__ void operator()(int, double) const {} };
template <bool TR, class EPI = XNoEpi>
__device__ RSQP_XINLINE void xgemv_w(const ldouble *M, int l, int nrows, int ncols, const ldouble *xv, double alpha,
                                     double beta, const ldouble *base, ldouble *out, int w0, int nw, ldouble *part, int lane,
                                     EPI epi = EPI()) {
    const int w = (lane >> 6) - w0, li = lane & 63;
    if (w < 0 || w >= nw) return;
    const int nout = TR ? ncols : nrows, ninner = TR ? nrows : ncols;
    if (nout <= 0) return;
    const int opw = nw == 1 ? nout : fdiv_small(nout + nw - 1, nw);
    ldouble *pw = part + (lane & ~63);
    if (opw > 64 || ninner < 8) {            // a lane owns whole outputs (several when the run exceeds the wave)
        for (int o = li; o < opw; o += 64) {
            const int og = w * opw + o;
            if (og < nout) {
                const double t = TR ? xdots<true>(M + og * l, 1, xv, 0, ninner) : xdots<false>(M + og, l, xv, 0, ninner);
                const double val = TR ? t : (base ? beta * base[og] : 0.0) + alpha * t;
                out[og] = val;
                epi(og, val);
            }
        }
        return;
    }
    int P = fdiv_small(64, opw);
#ifndef RSQP_XPMAX
#define RSQP_XPMAX 8
#endif
    if (P > RSQP_XPMAX) P = RSQP_XPMAX;
    const int p = fdiv_small(li, opw), o = li - p * opw, og = w * opw + o;
    const bool on = p < P && og < nout;
    const int ch = P == 1 ? ninner : fdiv_small(ninner + P - 1, P), j0 = p * ch, j1 = j0 + ch < ninner ? j0 + ch : ninner;
    if (on) pw[li] = j0 < j1 ? (TR ? xdots<true>(M + og * l, 1, xv, j0, j1) : xdots<false>(M + og, l, xv, j0, j1)) : 0.0;
    WSYNC();
    if (on && p == 0) {
        double t = pw[o];
        for (int k = 1; k < P; k++) t += pw[k * opw + o];
        const double val = TR ? t : (base ? beta * base[og] : 0.0) + alpha * t;
        out[og] = val;
        epi(og, val);
    }
}
// M[c*l + r] += coef * t[r] * v[c], the nrows x ncols elements dealt over nl lanes (rows fastest)
__device__ RSQP_XINLINE void xger_w(ldouble *M, int l, int nrows, int ncols, const ldouble *t, const ldouble *v, double coef,
                                    int nl, int lane) {
    if (nrows <= 0 || ncols <= 0) return;
    int np = nrows <= nl ? fdiv_small(nl, nrows) : 0;
    if (np < 1) {            // more rows than lanes: lane per row, all columns
        for (int r = lane; r < nrows; r += nl) {
            const double tr = coef * t[r];
            for (int c = 0; c < ncols; c++) M[c * l + r] += tr * v[c];
        }
        return;
    }
    if (np > ncols) np = ncols;
    const int pp = fdiv_small(lane, nrows), r = lane - pp * nrows;
    if (pp >= np) return;
    const double tr = coef * t[r];
    for (int c = pp; c < ncols; c += np) M[c * l + r] += tr * v[c];
}

template <int L, bool MAT_LDS>
struct EngineX {
    typedef typename MatPtr<MAT_LDS>::I MI;
    typedef typename MatPtr<MAT_LDS>::D MD;
    int nV, nC, ld, ldy, sizeT, ldm, haveH;
    double hreg;
    MI Ajc, Air; MD Aval;
    MI Arp, Aci; MD Arv;
    MI Hjc, Hir; MD Hval;
    static constexpr bool DENSE_MATS = MAT_LDS;   // A and H as dense column-major copies in LDS
    ldouble *Ad, *Hd;                             // nC x nV (ld nC), nV x nV (ld nV)
    ldouble *Z, *Wz, *Y, *Minv;
    ldouble *x, *g, *lb, *ub, *gN, *lbN, *ubN, *dx, *w1, *w2, *w3, *w4, *w5, *w6, *wz1, *wz2, *wz3;
    ldouble *Hx, *ATy;         // (H + hreg I) x and A'y_C of the iterate, carried by their increments (see drift_correction)
    ldouble *Ax, *lbA, *ubA, *lbAN, *ubAN, *dAx, *c1, *c2, *c3;
    ldouble *a1, *a2, *a3, *a4;
    ldouble *y, *dy, *scal;
    ldouble *part;             // L > 64: one double per lane for the split matrix phases (set by the kernel)
    ldouble *wq, *wv4, *wc1;   // staging aliases used by the kernel wrapper (guess / x0 / guessed constraints)
    lint *Sb, *Sc, *AC, *posAC, *iscal;
    int lane;
    int nFR, nAC, nZ, status, infeasible, unbounded, nflips;
    int since_refresh, dirty_products;
    int dx_ready;              // dx_FX and dy = 0 of the next step direction were written by drift_correction (FUSED builds)
    int hdim;                  // the stored entries of H lie in its leading hdim x hdim block (see stage_dense)
    long long tlast;

    // doubles of this formulation's image (<= rsqp_image_doubles, the size of the persistent copy)
    __host__ __device__ static long long image_doubles(int nV, int nC) {
        const long long ld = rsqp_ld(nV), sT = nV < nC ? nV : nC;
        return 2 * ld * nV + sT * ld + 19LL * nV + 9LL * nC + 2LL * (nV + nC) + 16 + 4 * (sT + 2);
    }
    __host__ __device__ static long long factor_doubles(int nV, int nC) {   // Z (+Y), Wz: (re)initialised by setup_aux
        return 2LL * rsqp_ld(nV) * nV;
    }
    // leading part of the image that survives a solve (bases, inverses, iterate, auxiliary data, multipliers)
    __host__ __device__ static long long persist_doubles(int nV, int nC) {
        const long long ld = rsqp_ld(nV), sT = nV < nC ? nV : nC;
        return 2 * ld * nV + sT * ld + 4LL * nV + 3LL * nC + (nV + nC);
    }
    __host__ __device__ static long long image_ints(int nV, int nC) { return nV + 3LL * nC + 8; }
    __device__ __forceinline__ void carve(lchar *base, int nV_, int nC_) {
        nV = nV_; nC = nC_; ld = rsqp_ld(nV); sizeT = nV < nC ? nV : nC; ldm = sizeT | 1;
        ldouble *p = (ldouble *)base;
        Z = p; p += ld * nV;
        Wz = p; p += ld * nV;
        // Y shares Z's array: column j of Y is column nV-1-j of that array (nZ + nAC <= nV, so the two
        // never meet); the T slot of the image (sizeT * ld doubles) holds Minv and the four a-vectors
        Y = Z + (nV - 1) * ld; ldy = -ld;
        ldouble *tslot = p; p += sizeT * ld;
#define CARVE_V(name) name = p; p += nV
#define CARVE_C(name) name = p; p += nC
        // what a hot start needs (persist_doubles, written back to HBM) ...
        CARVE_V(x); CARVE_V(g); CARVE_V(lb); CARVE_V(ub);
        CARVE_C(Ax); CARVE_C(lbA); CARVE_C(ubA);
        y = p; p += nV + nC;
        // ... and the per-solve scratch
        CARVE_V(gN); CARVE_V(lbN); CARVE_V(ubN);
        CARVE_V(dx); CARVE_V(w1); CARVE_V(w2); CARVE_V(w3); CARVE_V(w4); CARVE_V(w5); CARVE_V(w6);
        CARVE_V(wz1); CARVE_V(wz2); CARVE_V(wz3); CARVE_V(Hx); CARVE_V(ATy);
        CARVE_C(lbAN); CARVE_C(ubAN); CARVE_C(dAx); CARVE_C(c1); CARVE_C(c2); CARVE_C(c3);
#undef CARVE_V
#undef CARVE_C
        dy = p; p += nV + nC;
        scal = p; p += 16;
        Minv = tslot;                      // sizeT * ldm <= sizeT * ld
        a1 = p; p += sizeT + 2; a2 = p; p += sizeT + 2; a3 = p; p += sizeT + 2; a4 = p; p += sizeT + 2;
        lint *ip = (lint *)p;
        Sb = ip; ip += nV;
        Sc = ip; ip += nC;
        AC = ip; ip += nC;
        posAC = ip; ip += nC;
        iscal = ip; ip += 8;
        wq = wz1; wv4 = w4; wc1 = c1;
        nZ = 0;
    }

    // dense copies of the matrices behind the image (hs0xx-scale problems: they are small, and a dense
    // product is a run of independent loads where the compressed formats chain index -> value -> gather)
    __device__ __forceinline__ void stage_dense(lchar *mem, const int *gAjc, const int *gAir, const double *gAval,
                                                const int *gHjc, const int *gHir, const double *gHval) {
        Ad = (ldouble *)mem;
        Hd = Ad + nC * nV;
        for (int k = lane; k < nC * nV + nV * nV; k += L) Ad[k] = 0.0;
        SYNC();
        int hm = 0;
        PFOR(c, nV) {
            for (int k = gAjc[c]; k < gAjc[c + 1]; k++) Ad[gAir[k] + c * nC] = gAval[k];
            if (haveH)
                for (int k = gHjc[c]; k < gHjc[c + 1]; k++) {
                    const int r = gHir[k];
                    Hd[r + c * nV] = gHval[k];
                    hm = max(hm, max(r, c) + 1);
                }
        }
        SYNC();
        // The QPhandler formulation [x u v] gives H entries to the original variables only (the slacks have a linear
        // penalty): for a 69 x 28 member of the hs0xx batch 13 x 13 of the 69 x 69 dense copy. The Hessian products run
        // on that leading block; the rest of their output is zero. (Skipped terms are exact zeros: same sums.)
        double t = -(double)hm;
        int id = 0;
        block_argmin(t, id);
        hdim = (int)(-t);
    }
    // out = H v on the leading hdim x hdim block, zero beyond; the caller closes with the barrier
    template <class EPI = XNoEpi>
    __device__ __forceinline__ void h_tail_zero(ldouble *out, EPI epi = EPI()) {
        for (int c = hdim + lane; c < nV; c += L) { out[c] = 0.0; epi(c, 0.0); }
    }

    // ------------------------------------------------------------------ reductions
    // butterflies over the lanes of the problem; with several waves (L > 64) the wave results meet
    // in `scal` and are combined in wave order, so every lane ends with the same value
    __device__ __forceinline__ double block_sum(double v) {
        v = allreduce_sum<ilog2c(L > 64 ? 64 : L)>(v);
        if constexpr (L > 64) {
            __syncthreads();
            if ((lane & 63) == 0) scal[lane >> 6] = v;
            __syncthreads();
            v = 0.0;
#pragma unroll
            for (int w = 0; w < L / 64; w++) v += scal[w];
        }
        return v;
    }
    __device__ __forceinline__ void block_argmin(double &t, int &id) {
        allreduce_argmin<ilog2c(L > 64 ? 64 : L)>(t, id);
        if constexpr (L > 64) {
            __syncthreads();
            if ((lane & 63) == 0) { scal[lane >> 6] = t; scal[8 + (lane >> 6)] = (double)id; }   // ids exceed the 16-bit LDS integers
            __syncthreads();
            t = scal[0]; id = (int)scal[8];
#pragma unroll
            for (int w = 1; w < L / 64; w++) {
                const double t2 = scal[w]; const int id2 = (int)scal[8 + w];
                if (t2 < t || (t2 == t && id2 < id)) { t = t2; id = id2; }
            }
        }
    }
    __device__ __forceinline__ double dot(const ldouble *a, const ldouble *b, int n) {
        if constexpr (L > 64) {
            // several waves per problem: every wave forms the whole sum by itself from the (already published) LDS
            // operands -- same arithmetic in each of them, so all lanes agree and no workgroup barrier is needed
            double s = 0.0;
            for (int i = lane & 63; i < n; i += 64) s += a[i] * b[i];
            return allreduce_sum<6>(s);
        }
        double s = 0.0;
        PFOR(i, n) s += a[i] * b[i];
        return block_sum(s);
    }

    // ------------------------------------------------------------------ dense building blocks (column-major)
    // Four waves per problem (L > 64): a matrix phase is split over (index, part) = (lane % n1,
    // lane / n1): every part takes a contiguous slice of the inner dimension, partial sums meet in
    // `part` (one double per lane) and are added in part order. false = this lane has no slice.
    __device__ __forceinline__ bool split2d(int n1, int n2, int &i, int &j0, int &j1, int &nparts) {
        nparts = fdiv_small(L, n1);
        if (nparts > n2) nparts = n2;
        if (nparts < 1) nparts = 1;
        const int p = fdiv_small(lane, n1);
        i = lane - p * n1;
        const int chunk = fdiv_small(n2 + nparts - 1, nparts);
        j0 = p * chunk;
        j1 = j0 + chunk < n2 ? j0 + chunk : n2;
        return p < nparts;
    }
    __device__ __forceinline__ static double dot8(const ldouble *a, int sa, const ldouble *b, int j0, int j1) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int j = j0;
        for (; j + 8 <= j1; j += 8) {
            double m[8], xx[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { m[u] = a[(j + u) * sa]; xx[u] = b[j + u]; }
            s0 += m[0] * xx[0]; s1 += m[1] * xx[1]; s2 += m[2] * xx[2]; s3 += m[3] * xx[3];
            s0 += m[4] * xx[4]; s1 += m[5] * xx[5]; s2 += m[6] * xx[6]; s3 += m[7] * xx[7];
        }
        for (; j < j1; j++) s0 += a[j * sa] * b[j];
        return (s0 + s1) + (s2 + s3);
    }
    // ---- several waves per problem (L > 64): one workgroup barrier per product or group of independent products (xgemv_w)
    static constexpr int NW = L > 64 ? L / 64 : 1;
    template <bool TR, class EPI = XNoEpi>
    __device__ __forceinline__ void gemv_w(const ldouble *M, int l, int nrows, int ncols, const ldouble *xv, double alpha,
                                           double beta, const ldouble *base, ldouble *out, int w0, int nw, EPI epi = EPI()) {
        xgemv_w<TR, EPI>(M, l, nrows, ncols, xv, alpha, beta, base, out, w0, nw, part, lane, epi);
    }

    // out[c] = sum_r M[c*l + r] * xv[r]      (lane per column; eight rows per trip: their 16 LDS reads
    // are issued before the first multiply, four independent partial sums)
    __device__ __forceinline__ void gemv_t(const ldouble *M, int l, int nrows, int ncols, const ldouble *xv, ldouble *out) {
        if constexpr (L > 64) {
            gemv_w<true>(M, l, nrows, ncols, xv, 1.0, 0.0, nullptr, out, 0, NW);
            SYNC();
            return;
        }
        PFOR(c, ncols) out[c] = dot8(M + c * l, 1, xv, 0, nrows);
        SYNC();
    }
    // out[r] = beta * base[r] + alpha * sum_c M[c*l + r] * wv[c]     (lane per row)
    __device__ __forceinline__ void gemv_n(const ldouble *M, int l, int nrows, int ncols, const ldouble *wv, double alpha,
                                           double beta, const ldouble *base, ldouble *out) {
        if constexpr (L > 64) {
            gemv_w<false>(M, l, nrows, ncols, wv, alpha, beta, base, out, 0, NW);
            SYNC();
            return;
        }
        PFOR(r, nrows) out[r] = (base ? beta * base[r] : 0.0) + alpha * dot8(M + r, l, wv, 0, ncols);
        SYNC();
    }
    // M[c*l + r] += coef * t[r] * v[c]
    __device__ __forceinline__ void ger(ldouble *M, int l, int nrows, int ncols, const ldouble *t, const ldouble *v, double coef) {
        if constexpr (L > 64) {
            xger_w(M, l, nrows, ncols, t, v, coef, L, lane);
        } else {
            PFOR(rr, nrows) {
                ldouble *row = M + rr;
                const double tr = coef * t[rr];
                int c = 0;
                for (; c + 8 <= ncols; c += 8) {
                    double m[8], vv[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) { m[u] = row[(c + u) * l]; vv[u] = v[c + u]; }
#pragma unroll
                    for (int u = 0; u < 8; u++) row[(c + u) * l] = m[u] + tr * vv[u];
                }
                for (; c < ncols; c++) row[c * l] += tr * v[c];
            }
        }
        SYNC();
    }
    __device__ __forceinline__ void copyv(const ldouble *src, ldouble *dst, int n) {
        PFOR(i, n) dst[i] = src[i];
        SYNC();
    }
    __device__ __forceinline__ ldouble *Zc(int c) { return Z + c * ld; }
    __device__ __forceinline__ ldouble *Yc(int c) { return Y + c * ldy; }

    // ------------------------------------------------------------------ sparse products
    template <class IP, class DP>
    __device__ __forceinline__ static double sparse_dot(IP idx, DP val, const ldouble *v, int k0, int k1) {
        double s = 0.0;
        int k = k0;
        for (; k + 4 <= k1; k += 4) {
            const int i0 = idx[k], i1 = idx[k + 1], i2 = idx[k + 2], i3 = idx[k + 3];
            const double b0 = val[k], b1 = val[k + 1], b2 = val[k + 2], b3 = val[k + 3];
            const double v0 = v[i0], v1 = v[i1], v2 = v[i2], v3 = v[i3];
            s += b0 * v0; s += b1 * v1; s += b2 * v2; s += b3 * v3;
        }
        for (; k < k1; k++) s += val[k] * v[idx[k]];
        return s;
    }
    __device__ __forceinline__ void A_times(const ldouble *v, ldouble *out) {
        if constexpr (DENSE_MATS) { gemv_n(Ad, nC, nC, nV, v, 1.0, 0.0, nullptr, out); return; }
        PFOR(r, nC) out[r] = sparse_dot(Aci, Arv, v, Arp[r], Arp[r + 1]);
        SYNC();
    }
    __device__ __forceinline__ void AT_times(const ldouble *yc, ldouble *out) {
        if constexpr (DENSE_MATS) { gemv_t(Ad, nC, nC, nV, yc, out); return; }
        PFOR(c, nV) out[c] = sparse_dot(Air, Aval, yc, Ajc[c], Ajc[c + 1]);
        SYNC();
    }
    __device__ __forceinline__ void H_times(const ldouble *v, ldouble *out) {
        if constexpr (DENSE_MATS) {
            h_tail_zero(out);
            gemv_t(Hd, nV, hdim, hdim, v, out);    // H is symmetric: column sums = row sums
            if (hreg != 0.0) { PFOR(c, nV) out[c] += hreg * v[c]; SYNC(); }
            return;
        }
        PFOR(c, nV) {
            const double s = haveH ? sparse_dot(Hir, Hval, v, Hjc[c], Hjc[c + 1]) : 0.0;
            out[c] = s + hreg * v[c];
        }
        SYNC();
    }
    // independent products of one stage, side by side on disjoint waves (several waves per problem, dense copies)
    static constexpr bool FUSED = L > 64 && MAT_LDS;
    // outA = A v, outH = (H + hreg I) v
    __device__ __forceinline__ void AH_times(const ldouble *v, ldouble *outA, ldouble *outH) { AH_times(v, outA, outH, XNoEpi(), XNoEpi()); }
    // epiA / epiH ride on the two products (FUSED builds; the callers of the other builds keep their separate loops)
    template <class EA, class EH>
    __device__ __forceinline__ void AH_times(const ldouble *v, ldouble *outA, ldouble *outH, EA epiA, EH epiH) {
        if constexpr (FUSED) {
            // waves in proportion to the work: A has nC x nV entries, the Hessian block hdim x hdim
            int wa = nC > 0 ? 1 : 0;
            if (nC > 0 && 3 * hdim * hdim < nC * nV) wa = NW - 1;
            gemv_w<false>(Ad, nC, nC, nV, v, 1.0, 0.0, nullptr, outA, 0, wa, epiA);
            gemv_w<true>(Hd, nV, hdim, hdim, v, 1.0, 0.0, nullptr, outH, wa, NW - wa, epiH);
            h_tail_zero(outH, epiH);
            SYNC();
            if (hreg != 0.0) { PFOR(c, nV) outH[c] += hreg * v[c]; SYNC(); }
        } else {
            A_times(v, outA);
            H_times(v, outH);
        }
    }
    // Ax = A xv, aty = A' yc, hx = (H + hreg I) xv
    __device__ __forceinline__ void AAtH_times(const ldouble *xv, const ldouble *yc, ldouble *outA, ldouble *aty, ldouble *hx) {
        if constexpr (FUSED) {
            const int wa = nC > 0 ? 1 : 0;
            const int wh = (nC > 0 && 3 * hdim * hdim < nC * nV) ? 1 : NW - wa - 1;   // waves of the Hessian block
            gemv_w<false>(Ad, nC, nC, nV, xv, 1.0, 0.0, nullptr, outA, 0, wa);
            gemv_w<true>(Ad, nC, nC, nV, yc, 1.0, 0.0, nullptr, aty, wa, NW - wa - wh);   // inner dimension nC: short sums
            gemv_w<true>(Hd, nV, hdim, hdim, xv, 1.0, 0.0, nullptr, hx, NW - wh, wh);
            h_tail_zero(hx);
            SYNC();
            if (hreg != 0.0) { PFOR(c, nV) hx[c] += hreg * xv[c]; SYNC(); }
        } else {
            A_times(xv, outA);
            AT_times(yc, aty);
            H_times(xv, hx);
        }
    }
    __device__ __forceinline__ void row_of_A(int i, ldouble *a, bool all) {
        if constexpr (DENSE_MATS) {
            PFOR(v, nV) a[v] = (all || Sb[v] == 0) ? Ad[i + v * nC] : 0.0;
            SYNC();
            return;
        }
        PFOR(v, nV) a[v] = 0.0;
        SYNC();
        for (int k = Arp[i] + lane; k < Arp[i + 1]; k += L) {
            int c = Aci[k];
            if (all || Sb[c] == 0) a[c] = Arv[k];
        }
        SYNC();
    }

    // ------------------------------------------------------------------ Householder onto the LAST component
    // v = w, v[n-1] += sgn(w[n-1]) |w|; beta = 1 / (|w| (|w| + |w[n-1]|)); P = I - beta v v' maps w to sgi |w| e_last
    __device__ __forceinline__ void house(const ldouble *w, int n, ldouble *v, double &alpha, double &beta, double &sgi) {
        const double s = dot(w, w, n);
        alpha = sqrt(s);
        const double wl = n > 0 ? w[n - 1] : 0.0, sg = wl >= 0.0 ? 1.0 : -1.0;
        SYNC();
        PFOR(i, n) v[i] = w[i] + (i == n - 1 ? sg * alpha : 0.0);
        beta = alpha > 0.0 ? 1.0 / (alpha * (alpha + fabs(wl))) : 0.0;
        sgi = -sg;
        SYNC();
    }

    // reflection of Z that puts the direction Z w (w = wz1, length nZ) into the last column; Wz follows
    // and loses its last row / column. Returns |w| and the sign of the image.
    __device__ __forceinline__ void z_reflect_and_shrink(bool wz_enabled, double &alpha, double &sgi) {
        double beta;
        house(wz1, nZ, wz2, alpha, beta, sgi);
        STAMP(19);
        if constexpr (FUSED) {
            const int wt = wz_enabled ? NW / 2 : NW;
            gemv_w<false>(Z, ld, nV, nZ, wz2, 1.0, 0.0, nullptr, w5, 0, wt);                        // t = Z v
            if (wz_enabled) gemv_w<false>(Wz, ld, nZ, nZ, wz2, 1.0, 0.0, nullptr, wz3, wt, NW - wt); // s = Wz v
            SYNC();
            STAMP(20);
            ger(Z, ld, nV, nZ, w5, wz2, -beta);                   // Z -= beta t v'
            STAMP(21);
            if (!wz_enabled) return;
        } else {
        gemv_n(Z, ld, nV, nZ, wz2, 1.0, 0.0, nullptr, w5);        // t = Z v
        ger(Z, ld, nV, nZ, w5, wz2, -beta);                       // Z -= beta t v'
        if (!wz_enabled) return;
        gemv_n(Wz, ld, nZ, nZ, wz2, 1.0, 0.0, nullptr, wz3);      // s = Wz v
        }
        const double theta = dot(wz2, wz3, nZ);
        const int l = nZ - 1;
        const double vl = wz2[l], sl = wz3[l];
        PFOR(a, nZ) w6[a] = Wz[l * ld + a] - beta * wz3[a] * vl - beta * wz2[a] * sl + beta * beta * theta * wz2[a] * vl;
        SYNC();
        STAMP(22);
        const double w22 = w6[l];
        {
            // one division per row (the HBM engine divides per element; same value up to rounding)
            int a = lane, b0 = 0, b1 = l, np = 1;
            bool on = lane < l;
            if constexpr (L > 64) { if (l > 0 && l <= L / 2) on = split2d(l, l, a, b0, b1, np); }
            if (np > 1) {
                if (on) {
                    const double sa = beta * wz3[a], va = beta * wz2[a], vt = beta * beta * theta * wz2[a], cw = w6[a] / w22;
                    for (int b = b0; b < b1; b++) Wz[b * ld + a] += -sa * wz2[b] - va * wz3[b] + vt * wz2[b] - cw * w6[b];
                }
            } else {
                PFOR(aa, l) {
                    const double sa = beta * wz3[aa], va = beta * wz2[aa], vt = beta * beta * theta * wz2[aa], cw = w6[aa] / w22;
                    for (int b = 0; b < l; b++) Wz[b * ld + aa] += -sa * wz2[b] - va * wz3[b] + vt * wz2[b] - cw * w6[b];
                }
            }
        }
        SYNC();
        STAMP(23);
    }

    // new row nAC: -(wY' Minv)/eta ; new column nAC: 0 ; corner 1/eta   (wY in a1)
    __device__ __forceinline__ void minv_append(double eta) {
        gemv_t(Minv, ldm, nAC, nAC, a1, a2);
        PFOR(j, nAC + 1) {
            if (j < nAC) { Minv[j * ldm + nAC] = -a2[j] / eta; Minv[nAC * ldm + j] = 0.0; }
            else Minv[nAC * ldm + nAC] = 1.0 / eta;
        }
        SYNC();
        STAMP(24);
    }

    // products of constraint row r with the bases: w1 = a_FR, wz1 = Z'a, a1 = Y'a (rows of fixed
    // variables are zero in both bases, so the sparse row is used as it is)
    __device__ __forceinline__ void constraint_products(int r, double &na2, double &wz2n) {
        row_of_A(r, w1, false);
        STAMP(16);
        if constexpr (FUSED) {
            int wz = nAC == 0 ? NW : (nZ == 0 ? 0 : fdiv_small(NW * nZ + ((nZ + nAC) >> 1), nZ + nAC));
            if (nZ > 0 && wz < 1) wz = 1;
            if (nAC > 0 && wz > NW - 1) wz = NW - 1;
            gemv_w<true>(Z, ld, nV, nZ, w1, 1.0, 0.0, nullptr, wz1, 0, wz);
            gemv_w<true>(Y, ldy, nV, nAC, w1, 1.0, 0.0, nullptr, a1, wz, NW - wz);
            SYNC();
        } else if constexpr (DENSE_MATS) {
            gemv_t(Z, ld, nV, nZ, w1, wz1);
            gemv_t(Y, ldy, nV, nAC, w1, a1);
        } else {
            const int k0 = Arp[r], k1 = Arp[r + 1];
            PFOR(c, nZ) wz1[c] = sparse_dot(Aci, Arv, Zc(c), k0, k1);
            PFOR(c, nAC) a1[c] = sparse_dot(Aci, Arv, Yc(c), k0, k1);
            SYNC();
        }
        STAMP(17);
        na2 = dot(w1, w1, nV);
        wz2n = dot(wz1, wz1, nZ);
        STAMP(18);
    }
    __device__ __forceinline__ void bound_products(int v, double &na2, double &wz2n) {
        PFOR(c, nZ) wz1[c] = Z[c * ld + v];
        PFOR(c, nAC) a1[c] = Y[c * ldy + v];
        SYNC();
        na2 = 1.0;
        wz2n = dot(wz1, wz1, nZ);
    }
    __device__ __forceinline__ bool is_LI(double na2, double wz2n) {
        return nZ > 0 && na2 > 0.0 && sqrt(wz2n) > RSQP_EPS_LI * sqrt(na2);
    }

    // prerequisites: constraint_products(r). skipZ: exchange / flip -- the row is orthogonal to all
    // null-space columns but the last, which becomes the new Y column as it is
    __device__ __forceinline__ void add_constraint(int r, int side, bool skipZ, bool wz_enabled) {
        double eta;
        if (!skipZ) {
            double alpha, sgi;
            z_reflect_and_shrink(wz_enabled, alpha, sgi);
            copyv(Zc(nZ - 1), Yc(nAC), nV);
            eta = sgi * alpha;
        } else {
            copyv(Zc(nZ - 1), Yc(nAC), nV);
            eta = dot(w1, Zc(nZ - 1), nV);
        }
        nZ--;
        minv_append(eta);
        if (lane == 0) { AC[nAC] = r; posAC[r] = nAC; Sc[r] = side; }
        nAC++;
        SYNC();
    }

    // prerequisites: bound_products(v)
    __device__ __forceinline__ void add_bound(int v, int side, bool skipZ) {
        if (!skipZ) { double alpha, sgi; z_reflect_and_shrink(true, alpha, sgi); }
        const ldouble *zs = Zc(nZ - 1);
        nZ--;
        // q~ = [qY ; q*], |q~| = 1: vt = q~ with last += sgn(q*); beta~ = 1/(1+|q*|); gamma = beta~/|q*|
        const double qs = zs[v], sg = qs >= 0.0 ? 1.0 : -1.0, aq = fabs(qs);
        const double beta = 1.0 / (1.0 + aq), gamma = beta / aq, vlast = qs + sg;
        gemv_n(Y, ldy, nV, nAC, a1, 1.0, 0.0, nullptr, w5);        // t = Y vY
        PFOR(i, nV) w5[i] += vlast * zs[i];                       //   + zs vlast
        SYNC();
        ger(Y, ldy, nV, nAC, w5, a1, -beta);                       // Y -= beta~ t vY'
        gemv_t(Minv, ldm, nAC, nAC, a1, a2);
        ger(Minv, ldm, nAC, nAC, a1, a2, gamma);                  // Minv += gamma vY (vY' Minv)
        PFOR(c, nAC) Y[c * ldy + v] = 0.0;
        PFOR(c, nZ) Z[c * ld + v] = 0.0;
        if (lane == 0) Sb[v] = side;
        nFR--;
        SYNC();
        STAMP(25);
    }

    // grow Wz by the null-space column Z[:, nZ]; false = not positive definite (nZ unchanged)
    __device__ __forceinline__ bool wz_grow() {
        const ldouble *z = Zc(nZ);
        H_times(z, w2);
        const double kappa = dot(z, w2, nV);
        gemv_t(Z, ld, nV, nZ, w2, wz1);                            // k = Z'Hz
        gemv_n(Wz, ld, nZ, nZ, wz1, 1.0, 0.0, nullptr, wz2);       // u = Wz k
        const double ku = nZ > 0 ? dot(wz1, wz2, nZ) : 0.0;
        const double rho2 = kappa - ku, thr = RSQP_EPS_PD_REL * (fabs(kappa) + fabs(ku)) + RSQP_EPS_PD_ABS;
        if (!(rho2 > thr)) return false;
        const double ir2 = 1.0 / rho2;
        {
            // Wz += (u / rho2) u': a lane per row walked the whole row (nZ of the 256 lanes busy, nZ dependent LDS round
            // trips each: 16 % of the four-wave kernel's time); now (row, column slice) pairs over all lanes, same
            // arithmetic per element
            const int n = nZ;
            int a = lane, b0 = 0, b1 = n, np = 1;
            bool on = lane < n;
            if constexpr (L > 64) { if (n > 0 && n <= L / 2) on = split2d(n, n, a, b0, b1, np); }
            if (np > 1) {
                if (on) {
                    const double ua = wz2[a] / rho2;
                    for (int b = b0; b < b1; b++) Wz[b * ld + a] += ua * wz2[b];
                }
            } else {
                PFOR(aa, n) {
                    const double ua = wz2[aa] / rho2;
                    for (int b = 0; b < n; b++) Wz[b * ld + aa] += ua * wz2[b];
                }
            }
            PFOR(aa, n) {                       // new last column and row
                const double ua = wz2[aa] / rho2;
                Wz[n * ld + aa] = -ua;
                Wz[aa * ld + n] = -ua;
            }
            if (lane == 0) Wz[n * ld + n] = ir2;
        }
        nZ++;
        SYNC();
        STAMP(26);
        return true;
    }

    // Y loses the column that carries constraint position k; it lands in Z[:, nZ]
    __device__ __forceinline__ void remove_constraint_tq(int k) {
        const int r = AC[k];
        SYNC();
        copyv(Minv + k * ldm, a1, nAC);                            // u = Minv[:, k]
        double alpha, beta, sgi;
        house(a1, nAC, a2, alpha, beta, sgi);
        gemv_n(Y, ldy, nV, nAC, a2, 1.0, 0.0, nullptr, w5);         // t = Y v
        ger(Y, ldy, nV, nAC, w5, a2, -beta);
        gemv_t(Minv, ldm, nAC, nAC, a2, a3);                       // s' = v' Minv
        ger(Minv, ldm, nAC, nAC, a2, a3, -beta);
        copyv(Yc(nAC - 1), Zc(nZ), nV);
        if (k != nAC - 1) {
            copyv(Minv + (nAC - 1) * ldm, Minv + k * ldm, nAC);
            if (lane == 0) { const int rl = AC[nAC - 1]; AC[k] = rl; posAC[rl] = k; }
        }
        if (lane == 0) { posAC[r] = -1; Sc[r] = 0; }
        nAC--;
        SYNC();
        STAMP(27);
    }

    // variable v becomes free: the null space gains the column Z[:, nZ]
    __device__ __forceinline__ void remove_bound_tq(int v) {
        if (lane == 0) Sb[v] = 0;
        nFR++;
        ldouble *znew = Zc(nZ);
        PFOR(i, nV) znew[i] = 0.0;
        PFOR(j, nAC) a4[j] = 0.0;
        SYNC();
        if (nAC == 0) {
            if (lane == 0) znew[v] = 1.0;
            SYNC();
            return;
        }
        if constexpr (DENSE_MATS) {
            PFOR(j, nAC) a4[j] = Ad[AC[j] + v * nC];
        } else {
            for (int k = Ajc[v] + lane; k < Ajc[v + 1]; k += L) {
                const int p = posAC[Air[k]];
                if (p >= 0) a4[p] = Aval[k];
            }
        }
        SYNC();
        gemv_n(Minv, ldm, nAC, nAC, a4, 1.0, 0.0, nullptr, a1);    // c = Minv a_v
        const double s = dot(a1, a1, nAC);
        SYNC();
        PFOR(i, nAC) a1[i] = -a1[i];                               // vY = -c
        const double nu = sqrt(1.0 + s), beta = 1.0 / (nu * (nu + 1.0)), vlast = 1.0 + nu;
        SYNC();
        gemv_n(Y, ldy, nV, nAC, a1, 1.0, 0.0, nullptr, w5);         // t = Y vY + e_v vlast
        if (lane == 0) w5[v] += vlast;
        SYNC();
        A_times(w5, c3);
        PFOR(j, nAC) a2[j] = c3[AC[j]];                            // p = A_AC t
        PFOR(i, nV) znew[i] = (i == v ? 1.0 : 0.0) - beta * w5[i] * vlast;
        SYNC();
        ger(Y, ldy, nV, nAC, w5, a1, -beta);
        gemv_n(Minv, ldm, nAC, nAC, a2, 1.0, 0.0, nullptr, a3);    // q1 = Minv p
        gemv_t(Minv, ldm, nAC, nAC, a1, a4);                       // q2' = vY' Minv
        const double d = dot(a1, a3, nAC);
        ger(Minv, ldm, nAC, nAC, a3, a4, beta / (1.0 - beta * d));
        STAMP(28);
    }

    // ------------------------------------------------------------------ removal with definiteness guard
    __device__ __forceinline__ int remove_with_guard(bool is_bound, int idx) {
        double na2, wz2n;
        if (is_bound) {
            const int old = Sb[idx];
            SYNC();
            remove_bound_tq(idx);
            if (lane == 0) y[idx] = 0.0;
            SYNC();
            if (wz_grow()) return RET_OK;
            const bool cant = (old == -1 && ubN[idx] >= RSQP_INFTY) || (old == 1 && lbN[idx] <= -RSQP_INFTY);
            nZ++;   // the candidate column is still Z[:, nZ]: treat it as the last null-space column
            bound_products(idx, na2, wz2n);
            add_bound(idx, cant ? old : -old, true);
            if (cant) return RET_UNBOUNDED;
            if (lane == 0) { if (old == -1) ub[idx] = x[idx]; else lb[idx] = x[idx]; }
            nflips++;
            SYNC();
            return RET_OK;
        } else {
            const int old = Sc[idx], k = posAC[idx];
            SYNC();
            remove_constraint_tq(k);
            if (lane == 0) y[nV + idx] = 0.0;
            SYNC();
            if (wz_grow()) return RET_OK;
            const bool cant = (old == -1 && ubAN[idx] >= RSQP_INFTY) || (old == 1 && lbAN[idx] <= -RSQP_INFTY);
            nZ++;
            constraint_products(idx, na2, wz2n);
            add_constraint(idx, cant ? old : -old, true, true);
            if (cant) return RET_UNBOUNDED;
            if (lane == 0) { if (old == -1) ubA[idx] = Ax[idx]; else lbA[idx] = Ax[idx]; }
            nflips++;
            SYNC();
            return RET_OK;
        }
    }

    // ------------------------------------------------------------------ exchange
    // incoming row in w4 (all variables), its Y-products in a1. Finds the partner, shifts the duals.
    __device__ __forceinline__ int ensure_LI(int side, double &y_new, int &pkind, int &pidx) {
        PFOR(i, nC) c1[i] = 0.0;
        SYNC();
        gemv_t(Minv, ldm, nAC, nAC, a1, a2);                       // xi[j] = sum_i Minv[i][j] wY[i]
        PFOR(j, nAC) c1[AC[j]] = a2[j];
        SYNC();
        AT_times(c1, w2);
        PFOR(v, nV) w3[v] = Sb[v] != 0 ? w4[v] - w2[v] : 0.0;      // xiB
        SYNC();
        const double sgn = side == 1 ? -1.0 : 1.0;
        double bt = RSQP_INFTY;
        int bid = 0x7fffffff;
        PFOR(i, nC) {
            if (Sc[i] != 0) {
                double xi = sgn * c1[i], yi = y[nV + i];
                double num = Sc[i] == -1 ? yi : -yi, den = Sc[i] == -1 ? xi : -xi;
                if (den > RSQP_EPS_DEN) {
                    double t = (num > 0.0 ? num : 0.0) / den;
                    if (t < bt || (t == bt && i < bid)) { bt = t; bid = i; }
                }
            }
        }
        PFOR(v, nV) {
            if (Sb[v] != 0) {
                double xi = sgn * w3[v], yi = y[v];
                double num = Sb[v] == -1 ? yi : -yi, den = Sb[v] == -1 ? xi : -xi;
                if (den > RSQP_EPS_DEN) {
                    double t = (num > 0.0 ? num : 0.0) / den;
                    if (t < bt || (t == bt && nC + v < bid)) { bt = t; bid = nC + v; }
                }
            }
        }
        block_argmin(bt, bid);
        if (bid == 0x7fffffff) return RET_INFEASIBLE;
        PFOR(i, nC) if (Sc[i] != 0) y[nV + i] -= bt * sgn * c1[i];
        PFOR(v, nV) if (Sb[v] != 0) y[v] -= bt * sgn * w3[v];
        SYNC();
        dirty_products = 1;                                        // the duals moved outside a step: A'y is recomputed
        y_new = sgn * bt;
        pkind = bid < nC ? 1 : 2;
        pidx = bid < nC ? bid : bid - nC;
        STAMP(29);
        return RET_OK;
    }

    __device__ __forceinline__ int change_active_set(const Blocking &b) {
        if (b.kind == 1) return remove_with_guard(false, b.idx);
        if (b.kind == 2) return remove_with_guard(true, b.idx);
        if (b.kind == 3 || b.kind == 4) {
            double ynew = 0.0, na2, wz2n;
            bool full = true;
            if (b.kind == 3) constraint_products(b.idx, na2, wz2n); else bound_products(b.idx, na2, wz2n);
            if (!is_LI(na2, wz2n)) {
                int pkind = 0, pidx = -1;
                if (b.kind == 3) row_of_A(b.idx, w4, true);
                else { PFOR(v, nV) w4[v] = v == b.idx ? 1.0 : 0.0; SYNC(); }
                const int rc_ = ensure_LI(b.side, ynew, pkind, pidx);
                if (rc_ != RET_OK) return rc_;
                if (pkind == 1) {
                    const int k = posAC[pidx];
                    SYNC();
                    remove_constraint_tq(k);
                    if (lane == 0) y[nV + pidx] = 0.0;
                } else {
                    remove_bound_tq(pidx);
                    if (lane == 0) y[pidx] = 0.0;
                }
                SYNC();
                full = wz_grow();
                if (!full) nZ++;   // keep the candidate column as the last null-space column
                if (b.kind == 3) constraint_products(b.idx, na2, wz2n); else bound_products(b.idx, na2, wz2n);
            }
            if (b.kind == 3) {
                add_constraint(b.idx, b.side, !full, true);
                if (lane == 0) y[nV + b.idx] = ynew;
            } else {
                add_bound(b.idx, b.side, !full);
                if (lane == 0) y[b.idx] = ynew;
            }
            SYNC();
        }
        return RET_OK;
    }

    // ------------------------------------------------------------------ auxiliary QP
    __device__ __forceinline__ static double clampinf(double v) {
        return v > RSQP_INFTY ? RSQP_INFTY : (v < -RSQP_INFTY ? -RSQP_INFTY : v);
    }
    __device__ __forceinline__ void store_targets(const double *g_, const double *lb_, const double *ub_,
                                                  const double *lbA_, const double *ubA_) {
        PFOR(v, nV) { gN[v] = g_[v]; lbN[v] = clampinf(lb_[v]); ubN[v] = clampinf(ub_[v]); }
        PFOR(i, nC) { lbAN[i] = clampinf(lbA_[i]); ubAN[i] = clampinf(ubA_[i]); }
        SYNC();
    }
    __device__ __forceinline__ bool bounds_inconsistent() {
        double bad = 0.0;
        PFOR(v, nV) if (lbN[v] > ubN[v] + RSQP_EPS) bad += 1.0;
        PFOR(i, nC) if (lbAN[i] > ubAN[i] + RSQP_EPS) bad += 1.0;
        return block_sum(bad) > 0.0;
    }

    // warm-start inputs staged by the caller: x0 in wv4 (= w4), y0 in dy, guessed bound status in wq
    // (= wz1) and guessed constraint status in wc1 (= c1), as doubles
    __device__ __forceinline__ int setup_aux(bool x0, bool y0, bool guess_b, bool guess_c, bool cy0 = false) {
        status = QPS_PREPARINGAUXILIARYQP;
        infeasible = unbounded = 0;
        PFOR(v, nV) {
            double xv = x0 ? wv4[v] : 0.0;
            int s;
            if (guess_b) s = (int)wq[v];
            else if (x0) s = xv <= lbN[v] + RSQP_BOUND_TOLERANCE ? -1 : (xv >= ubN[v] - RSQP_BOUND_TOLERANCE ? 1 : 0);
            else if (y0) s = dy[v] > RSQP_EPS ? -1 : (dy[v] < -RSQP_EPS ? 1 : 0);
            else s = -1;
            if (s == -1 && lbN[v] <= -RSQP_INFTY) s = (ubN[v] < RSQP_INFTY && !x0 && !guess_b) ? 1 : 0;
            if (s == 1 && ubN[v] >= RSQP_INFTY) s = 0;
            wv4[v] = xv;
            wq[v] = (double)s;
        }
        if (!y0) { PFOR(i, nV + nC) dy[i] = 0.0; }
        if (!guess_c) { PFOR(i, nC) wc1[i] = 0.0; }
        SYNC();
        PFOR(v, nV) { x[v] = wv4[v]; Sb[v] = (int)wq[v]; }
        PFOR(i, nV + nC) y[i] = dy[i];
        for (int k = lane; k < ld * nV; k += L) { Z[k] = 0.0; Wz[k] = 0.0; }
        for (int k = lane; k < sizeT * ldm; k += L) Minv[k] = 0.0;
        PFOR(i, nC) { Sc[i] = 0; posAC[i] = -1; c2[i] = wc1[i]; }   // the guess moves out of c1 (scratch of the products)
        SYNC();
        if (lane == 0) {
            int n = 0;
            for (int v = 0; v < nV; v++)
                if (Sb[v] == 0) Z[(n++) * ld + v] = 1.0;
            iscal[0] = n;
        }
        SYNC();
        nFR = nZ = iscal[0];
        nAC = 0;
        A_times(x, Ax);
        for (int i = 0; i < nC; i++) {
            int s = 0;
            if (guess_c) s = (int)c2[i];
            else if (y0 && (!x0 || cy0)) s = y[nV + i] > RSQP_EPS ? -1 : (y[nV + i] < -RSQP_EPS ? 1 : 0);
            else if (x0) s = Ax[i] <= lbAN[i] + RSQP_BOUND_TOLERANCE ? -1 : (Ax[i] >= ubAN[i] - RSQP_BOUND_TOLERANCE ? 1 : 0);
            if (s == -1 && lbAN[i] <= -RSQP_INFTY) s = 0;
            if (s == 1 && ubAN[i] >= RSQP_INFTY) s = 0;
            if (s != 0) {
                double na2, wz2n;
                constraint_products(i, na2, wz2n);
                if (is_LI(na2, wz2n)) add_constraint(i, s, false, false);
            }
        }
        // Wz = (Z'HZ)^-1 by bordering over the final null-space columns
        const int nZf = nZ;
        nZ = 0;
        for (int k = 0; k < nZf; k++)
            if (!wz_grow()) return RET_SETUP_FAILED;
        PFOR(v, nV) {
            double yv = y[v];
            if (Sb[v] == 0 || (Sb[v] == -1 && yv < 0.0) || (Sb[v] == 1 && yv > 0.0)) y[v] = 0.0;
        }
        PFOR(i, nC) {
            double yi = y[nV + i];
            if (Sc[i] == 0 || (Sc[i] == -1 && yi < 0.0) || (Sc[i] == 1 && yi > 0.0)) y[nV + i] = 0.0;
        }
        SYNC();
        AT_times(y + nV, w1);
        H_times(x, w2);
        PFOR(v, nV) {
            double xv = x[v];
            g[v] = w1[v] + y[v] - w2[v];
            lb[v] = Sb[v] == -1 ? xv : fmin(lbN[v], xv - RSQP_BOUND_RELAXATION);
            ub[v] = Sb[v] == 1 ? xv : fmax(ubN[v], xv + RSQP_BOUND_RELAXATION);
        }
        PFOR(i, nC) {
            double ax = Ax[i];
            lbA[i] = Sc[i] == -1 ? ax : fmin(lbAN[i], ax - RSQP_BOUND_RELAXATION);
            ubA[i] = Sc[i] == 1 ? ax : fmax(ubAN[i], ax + RSQP_BOUND_RELAXATION);
        }
        SYNC();
        status = QPS_AUXILIARYQPSOLVED;
        return RET_OK;
    }

    // A hot start on a state the tableau kernel left behind (qp_small_g.h wrote iterate, multipliers, homotopy
    // data and working set in this engine's layout, its own tableau G elsewhere) that that kernel could not finish:
    // build Z, Y, Minv, Wz for the stored working set and leave everything else -- x, y, g, lb, ub, lbA, ubA, A x -- as it is,
    // so that the homotopy continues exactly where a hot start of this engine would (the bases differ, the step directions
    // do not depend on them). Mirrors the factor part of setup_aux.
    static constexpr bool K_IMAGE = true;
    __device__ __forceinline__ int rebuild_factors() {
        for (int k = lane; k < ld * nV; k += L) { Z[k] = 0.0; Wz[k] = 0.0; }
        for (int k = lane; k < sizeT * ldm; k += L) Minv[k] = 0.0;
        PFOR(i, nC) { c2[i] = (double)Sc[i]; Sc[i] = 0; posAC[i] = -1; }
        SYNC();
        if (lane == 0) {
            int n = 0;
            for (int v = 0; v < nV; v++)
                if (Sb[v] == 0) Z[(n++) * ld + v] = 1.0;
            iscal[0] = n;
        }
        SYNC();
        nFR = nZ = iscal[0];
        nAC = 0;
        for (int i = 0; i < nC; i++) {
            const int s = (int)c2[i];
            if (s != 0) {
                double na2, wz2n;
                constraint_products(i, na2, wz2n);
                if (!is_LI(na2, wz2n)) return RET_SETUP_FAILED;
                add_constraint(i, s, false, false);
            }
        }
        const int nZf = nZ;
        nZ = 0;
        for (int k = 0; k < nZf; k++)
            if (!wz_grow()) return RET_SETUP_FAILED;
        status = QPS_AUXILIARYQPSOLVED;
        return RET_OK;
    }

    // ------------------------------------------------------------------ step direction
    __device__ __forceinline__ static double delta_of(double target, double cur) {
        return (fabs(target) >= RSQP_INFTY && fabs(cur) >= RSQP_INFTY) ? 0.0 : target - cur;
    }
    // The step direction of the several-waves build: every element-wise loop that followed a product (right-hand side of
    // the active constraints, tmpg, the sum H xY + tmpg, the merge of dx on the free variables, H dx + dg, the scatter
    // of the active multipliers, the multipliers of the fixed variables) rides on the product that feeds it, and dx_FX /
    // dy = 0 ride on the drift correction of the previous change: 17 barrier-separated stages become 10. Same arithmetic
    // per element as the plain version below.
    __device__ __forceinline__ void step_direction_fused() {
        if (!dx_ready) {
            PFOR(v, nV) dx[v] = Sb[v] == -1 ? delta_of(lbN[v], lb[v]) : (Sb[v] == 1 ? delta_of(ubN[v], ub[v]) : 0.0);
            PFOR(i, nV + nC) dy[i] = 0.0;
            SYNC();
        }
        dx_ready = 0;
        STAMP(10);
        auto epi_bA = [&](int r, double val) {                    // bA of an active constraint, at its working-set position
            const int pj = posAC[r];
            if (pj >= 0) a1[pj] = (Sc[r] == -1 ? delta_of(lbAN[r], lbA[r]) : delta_of(ubAN[r], ubA[r])) - val;
        };
        auto epi_tmpg = [&](int v, double val) { w1[v] = val + (gN[v] - g[v]); };
        if (nZ > 0) AH_times(dx, c1, w2, epi_bA, epi_tmpg);        // A dx_FX -> bA, H dx_FX -> tmpg
        else { gemv_w<false>(Ad, nC, nC, nV, dx, 1.0, 0.0, nullptr, c1, 0, NW, epi_bA); SYNC(); }
        STAMP(11);
        STAMP(12);
        // range space: wY = Minv bA ; xY = Y wY
        gemv_n(Minv, ldm, nAC, nAC, a1, 1.0, 0.0, nullptr, a2);
        gemv_n(Y, ldy, nV, nAC, a2, 1.0, 0.0, nullptr, w3);
        STAMP(13);
        // null space: wZ = -Wz Z'(tmpg + H xY) ; dx_FR = xY + Z wZ
        auto epi_merge = [&](int v, double val) { if (Sb[v] == 0) dx[v] = val; };
        if (nZ > 0) {
            auto epi_sum = [&](int v, double val) { w2[v] = 1.0 * val + 1.0 * w1[v]; };
            gemv_w<true>(Hd, nV, hdim, hdim, w3, 1.0, 0.0, nullptr, w2, 0, NW, epi_sum);
            h_tail_zero(w2, epi_sum);
            SYNC();
            gemv_t(Z, ld, nV, nZ, w2, wz1);
            gemv_n(Wz, ld, nZ, nZ, wz1, -1.0, 0.0, nullptr, wz2);
            gemv_w<false>(Z, ld, nV, nZ, wz2, 1.0, 1.0, w3, w4, 0, NW, epi_merge);
            SYNC();
        } else {
            PFOR(v, nV) { const double t = w3[v]; w4[v] = t; if (Sb[v] == 0) dx[v] = t; }
            SYNC();
        }
        STAMP(14);
        STAMP(15);
        // multipliers: dyAC = Minv' Y'(H dx + dg)
        auto epi_dg = [&](int v, double val) { w2[v] = val + (gN[v] - g[v]); };
        AH_times(dx, dAx, w5, XNoEpi(), epi_dg);                    // A dx (for the ratio tests) rides along with H dx
        gemv_t(Y, ldy, nV, nAC, w2, a1);
        auto epi_dyc = [&](int j, double val) { dy[nV + AC[j]] = val; };
        gemv_w<true>(Minv, ldm, nAC, nAC, a1, 1.0, 0.0, nullptr, a2, 0, NW, epi_dyc);
        SYNC();
        auto epi_dyb = [&](int v, double val) { dy[v] = Sb[v] != 0 ? w2[v] - val : 0.0; };
        gemv_w<true>(Ad, nC, nC, nV, dy + nV, 1.0, 0.0, nullptr, w3, 0, NW, epi_dyb);
        SYNC();
    }

    __device__ __forceinline__ void step_direction() {
        if constexpr (FUSED) { if (hreg == 0.0) { step_direction_fused(); return; } }   // (regularised LPs keep the plain version)
        dx_ready = 0;
        double anynz = 0.0;
        PFOR(v, nV) {
            const double d = Sb[v] == -1 ? delta_of(lbN[v], lb[v]) : (Sb[v] == 1 ? delta_of(ubN[v], ub[v]) : 0.0);
            dx[v] = d;
            if (d != 0.0) anynz = 1.0;
        }
        PFOR(i, nV + nC) dy[i] = 0.0;
        // dx_FX is the move of the ACTIVE bounds along the homotopy: once the bounds a problem keeps active do not move any
        // more (slack bounds of the [J I -I] formulation: lb = lbN = 0) it is exactly zero, and so are A dx_FX and H dx_FX
        // (tested per problem where a problem is at most one wave: 23 x 6 members 0.92 -> 0.85 ms; across the four waves of
        // a 69 x 28 member the two extra barriers cost what the skipped stage saves, so that build always multiplies)
        bool moving = true;
        if constexpr (L <= 64) moving = block_sum(anynz) > 0.0;
        SYNC();
        STAMP(10);
        if (moving) {
            if (nZ > 0) AH_times(dx, c1, w2);                       // A dx_FX and H dx_FX
            else A_times(dx, c1);                                   // (no null space: the projected-gradient part is empty)
        } else {
            PFOR(i, nC) c1[i] = 0.0;
            PFOR(v, nV) w2[v] = 0.0;
            SYNC();
        }
        STAMP(11);
        PFOR(j, nAC) {
            const int r = AC[j];
            a1[j] = (Sc[r] == -1 ? delta_of(lbAN[r], lbA[r]) : delta_of(ubAN[r], ubA[r])) - c1[r];   // bA
        }
        if (nZ > 0) { PFOR(v, nV) w1[v] = w2[v] + (gN[v] - g[v]); }  // tmpg
        SYNC();
        // range space: wY = Minv bA ; xY = Y wY
        STAMP(12);
        gemv_n(Minv, ldm, nAC, nAC, a1, 1.0, 0.0, nullptr, a2);
        gemv_n(Y, ldy, nV, nAC, a2, 1.0, 0.0, nullptr, w3);
        STAMP(13);
        // null space: wZ = -Wz Z'(tmpg + H xY) ; dx_FR = xY + Z wZ
        if (nZ > 0) {
            H_times(w3, w2);
            PFOR(v, nV) w2[v] = 1.0 * w2[v] + 1.0 * w1[v];
            SYNC();
            gemv_t(Z, ld, nV, nZ, w2, wz1);
            gemv_n(Wz, ld, nZ, nZ, wz1, -1.0, 0.0, nullptr, wz2);
            gemv_n(Z, ld, nV, nZ, wz2, 1.0, 1.0, w3, w4);
        } else {
            copyv(w3, w4, nV);
        }
        STAMP(14);
        PFOR(v, nV) if (Sb[v] == 0) dx[v] = w4[v];
        SYNC();
        STAMP(15);
        // multipliers: dyAC = Minv' Y'(H dx + dg)
        AH_times(dx, dAx, w5);                                      // A dx (for the ratio tests) rides along with H dx
        PFOR(v, nV) w2[v] = w5[v] + (gN[v] - g[v]);
        SYNC();
        gemv_t(Y, ldy, nV, nAC, w2, a1);
        gemv_t(Minv, ldm, nAC, nAC, a1, a2);
        PFOR(j, nAC) dy[nV + AC[j]] = a2[j];
        SYNC();
        AT_times(dy + nV, w3);
        PFOR(v, nV) dy[v] = Sb[v] != 0 ? w2[v] - w3[v] : 0.0;
        SYNC();
    }

    // ------------------------------------------------------------------ ratio tests (as Engine)
    __device__ __forceinline__ static void cand(double num, double den, int id, double &bt, int &bid) {
        if (den >= RSQP_EPS_DEN) {
            double t = (num > 0.0 ? num : 0.0) / den;
            if (t < bt || (t == bt && id < bid)) { bt = t; bid = id; }
        }
    }
    __device__ __forceinline__ Blocking ratio_tests() {
        double bt = 1.0;
        int bid = 0x7fffffff;
        // one candidate (one f64 division) per lane: the items are (constraint, lower or active), (constraint, upper),
        // (variable, lower or active), (variable, upper) -- a lane per constraint / variable evaluated up to two
        // candidates one after the other on the first nC lanes. The lexicographic minimum does not depend on the order.
        const int items = 2 * (nC + nV);
        for (int it = lane; it < items; it += L) {
            // branch-free up to the division: the lanes of a wave hold items of every kind (constraint / variable, lower /
            // upper / active), and a division inside each of eight divergent paths was 2 k cycles of the 4.6 k this
            // function took. Operands through selected base pointers (one round of loads), both candidate forms computed,
            // one division site.
            const bool isc = it < 2 * nC;
            const int jt = isc ? it : it - 2 * nC, n = isc ? nC : nV;
            const bool upper = jt >= n;
            const int i = upper ? jt - n : jt;
            const ldouble *pcur = isc ? Ax : x, *pdl = isc ? dAx : dx, *plo = isc ? lbA : lb, *ploN = isc ? lbAN : lbN,
                          *pup = isc ? ubA : ub, *pupN = isc ? ubAN : ubN;
            const lint *pS = isc ? Sc : Sb;
            const int yoff = isc ? nV + i : i;
            const double cur = pcur[i], d = pdl[i], lo = plo[i], loN = ploN[i], up = pup[i], upN = pupN[i], yi = y[yoff], dyi = dy[yoff];
            const int sflag = pS[i];
            const bool act = sflag != 0;
            // active: the multiplier reaches zero; inactive: the lower / upper side is reached
            const double num = act ? (sflag == -1 ? yi : -yi) : (upper ? up - cur : cur - lo);
            const double den = act ? (sflag == -1 ? -dyi : dyi) : (upper ? d - delta_of(upN, up) : delta_of(loN, lo) - d);
            const int id = act ? (isc ? i : nC + i)
                               : (upper ? (isc ? 2 * nC + nV : 3 * nC + 2 * nV) + i : (isc ? nC + nV : 3 * nC + nV) + i);
            const bool ok = act ? !upper : (upper ? upN < RSQP_INFTY : loN > -RSQP_INFTY);
            if (ok) cand(num, den, id, bt, bid);
        }
        if (!(bt < 1.0)) { bt = 1.0; bid = 0x7fffffff; }
        block_argmin(bt, bid);
        Blocking b;
        b.tau = bt; b.kind = 0; b.idx = -1; b.side = 0;
        if (bid != 0x7fffffff) {
            if (bid < nC) { b.kind = 1; b.idx = bid; }
            else if (bid < nC + nV) { b.kind = 2; b.idx = bid - nC; }
            else if (bid < 2 * nC + nV) { b.kind = 3; b.idx = bid - nC - nV; b.side = -1; }
            else if (bid < 3 * nC + nV) { b.kind = 3; b.idx = bid - 2 * nC - nV; b.side = 1; }
            else if (bid < 3 * nC + 2 * nV) { b.kind = 4; b.idx = bid - 3 * nC - nV; b.side = -1; }
            else { b.kind = 4; b.idx = bid - 3 * nC - 2 * nV; b.side = 1; }
        }
        return b;
    }

    // ------------------------------------------------------------------ homotopy (as Engine)
    // A x, A'y_C and (H + hreg I) x follow the iterate by axpy (their increments are by-products of the step direction)
    // and are recomputed exactly every REFRESH working-set changes and whenever an exchange moved the duals outside a
    // step -- the scheme of the HBM-resident engine (qp_large.hip refresh_products / drift_correction). Before, every
    // change paid the three products (11 % of the four-wave kernel's time).
#ifndef RSQP_X_REFRESH
#define RSQP_X_REFRESH 8
#endif
    static constexpr int REFRESH = RSQP_X_REFRESH;
    __device__ __forceinline__ void refresh_products() {
        AAtH_times(x, y + nV, Ax, ATy, Hx);
        dirty_products = 0; since_refresh = 0;
    }
    __device__ __forceinline__ void drift_correction() {
        PFOR(v, nV) if (Sb[v] != 0) x[v] = Sb[v] == -1 ? lb[v] : ub[v];
        SYNC();
        if (dirty_products || ++since_refresh >= REFRESH) refresh_products();
        PFOR(i, nC) { if (Sc[i] == -1) lbA[i] = Ax[i]; else if (Sc[i] == 1) ubA[i] = Ax[i]; }
        PFOR(v, nV) g[v] = ATy[v] + y[v] - Hx[v];
        if constexpr (FUSED) {      // first loop of the next step direction: it reads nothing this function writes but Sb / lb / ub
            PFOR(v, nV) dx[v] = Sb[v] == -1 ? delta_of(lbN[v], lb[v]) : (Sb[v] == 1 ? delta_of(ubN[v], ub[v]) : 0.0);
            PFOR(i, nV + nC) dy[i] = 0.0;
            dx_ready = 1;
        }
        SYNC();
    }
    __device__ __forceinline__ int homotopy(int maxit, int &nWSR) {
        int iter = 0, rcode = RET_OK;
        status = QPS_PERFORMINGHOMOTOPY;
        PFOR(v, nV) {
            if (Sb[v] != -1 && lb[v] <= -RSQP_INFTY && lbN[v] > -RSQP_INFTY) lb[v] = fmin(lbN[v], x[v] - RSQP_BOUND_RELAXATION);
            if (Sb[v] != 1 && ub[v] >= RSQP_INFTY && ubN[v] < RSQP_INFTY) ub[v] = fmax(ubN[v], x[v] + RSQP_BOUND_RELAXATION);
        }
        PFOR(i, nC) {
            if (Sc[i] != -1 && lbA[i] <= -RSQP_INFTY && lbAN[i] > -RSQP_INFTY) lbA[i] = fmin(lbAN[i], Ax[i] - RSQP_BOUND_RELAXATION);
            if (Sc[i] != 1 && ubA[i] >= RSQP_INFTY && ubAN[i] < RSQP_INFTY) ubA[i] = fmax(ubAN[i], Ax[i] + RSQP_BOUND_RELAXATION);
        }
        SYNC();
        PFOR(i, nC) c3[i] = Ax[i];                  // (the relaxation above used the Ax of the set-up: keep that copy bit for bit)
        SYNC();
        dx_ready = 0;
        refresh_products();                         // Hx, ATy (and Ax) of the starting point
        PFOR(i, nC) Ax[i] = c3[i];
        SYNC();
        for (;;) {
            STAMP(7);
            step_direction();
            STAMP(3);
            Blocking b = ratio_tests();
            STAMP(4);
            double tau = b.tau;
            bool done = b.kind == 0;
            PFOR(v, nV) {
                if (done) {
                    g[v] = gN[v]; lb[v] = lbN[v]; ub[v] = ubN[v];
                    x[v] = Sb[v] == -1 ? lb[v] : (Sb[v] == 1 ? ub[v] : x[v] + tau * dx[v]);
                } else {
                    x[v] += tau * dx[v];
                    g[v] += tau * (gN[v] - g[v]);
                    lb[v] += tau * delta_of(lbN[v], lb[v]);
                    ub[v] += tau * delta_of(ubN[v], ub[v]);
                }
            }
            PFOR(i, nV + nC) y[i] += tau * dy[i];
            if (!done) { PFOR(v, nV) { Hx[v] += tau * w5[v]; ATy[v] += tau * w3[v]; } }   // H dx and A'dy_C of the step direction
            PFOR(i, nC) {
                if (done) { lbA[i] = lbAN[i]; ubA[i] = ubAN[i]; }
                else {
                    lbA[i] += tau * delta_of(lbAN[i], lbA[i]); ubA[i] += tau * delta_of(ubAN[i], ubA[i]);
                    // A x follows the step by its increment (a by-product of the step direction): between here and the
                    // exact product in drift_correction only Ax[blocking] / Ax[flipped] are read, and only into bounds
                    // that drift_correction overwrites with the exact product once the constraint is active
                    Ax[i] += tau * dAx[i];
                }
            }
            SYNC();
            if (done) A_times(x, Ax);
            STAMP(5);
            if (done) { status = QPS_SOLVED; break; }
            if (iter >= maxit) { rcode = RET_MAX_NWSR; break; }
            if (lane == 0) {
                if (b.kind == 3) { if (b.side == -1) lbA[b.idx] = Ax[b.idx]; else ubA[b.idx] = Ax[b.idx]; }
                else if (b.kind == 4) { if (b.side == -1) lb[b.idx] = x[b.idx]; else ub[b.idx] = x[b.idx]; }
            }
            SYNC();
            rcode = change_active_set(b);
            STAMP(6);
            if (rcode == RET_INFEASIBLE) { infeasible = 1; break; }
            if (rcode == RET_UNBOUNDED) { unbounded = 1; break; }
            iter++;
            drift_correction();
        }
        nWSR = iter;
        return rcode;
    }
    __device__ __forceinline__ void restore(int nFR_, int nAC_, int status_) { nFR = nFR_; nAC = nAC_; nZ = nFR_ - nAC_; status = status_; }
    __device__ __forceinline__ double objective() {
        H_times(x, w2);
        double a = dot(x, w2, nV), b = dot(gN, x, nV), c = dot(x, x, nV);
        return 0.5 * (a - hreg * c) + b;
    }
};
