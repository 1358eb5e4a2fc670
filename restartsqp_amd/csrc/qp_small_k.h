// qp_small_k.h -- included by qp_small.hip inside its anonymous namespace (after qp_small_x.h).
//
// small_qpk_kernel: the LDS / register-resident engine in the EXPLICIT-KKT-INVERSE formulation. Same homotopy, ratio tests,
// tie breaks and drift correction as the other engines and the CPU restatement (the step directions of an active-set
// iteration do not depend on how the KKT system is solved), but the whole linear algebra of a working-set change is ONE
// symmetric matrix
//        M = K^-1,   K = [ H_FR,FR   A_AC,FR' ]      over the index set S = free variables + active constraints,
//                        [ A_AC,FR   0        ]
// kept current by bordering (an index enters S: a variable is freed, a constraint becomes active) and by a Schur-complement
// step (an index leaves S): ONE rank-1 update each. The step direction is ONE product with M instead of the ~10 dependent
// products and solves of the null-space formulations, a working-set change 7-8 barrier-separated phases instead of ~50
//  -- and a phase is what a change costs on this chip: the problems are a few thousand multiply-adds, the
// latency of a dependent LDS round trip + reduction + barrier is what adds up.
//
// MI355X mapping (one problem per workgroup of 256 lanes = 4 waves, lane = 8 bi + bj: 32 row blocks x 8 column blocks):
//   * EVERY matrix lives in REGISTERS: lane (bi, bj) holds the RV x CV block of H, the RC x CV block of A, the RV x CC block
//     of A' and the (RV + RC) x (CV + CC) block of M that belong to variable rows bi RV + a / constraint rows bi RC + c and
//     variable columns bj CV + b / constraint columns bj CC + c. Slots of M are FIXED (slot of variable v, slot of
//     constraint i; rows and columns of indices outside S hold zeros or rounding-size residue and every consumer masks by
//     the working set), so nothing is ever compacted or moved, every loop has compile-time bounds and every register index
//     is a constant.
//   * a product = the LDS reads of the input slice, (rows x cols) FMAs from registers, a sum over the 8 lanes of a row block
//     (3 DPP steps: quad_perm, quad_perm, row_half_mirror -- no LDS traffic), one LDS write per row by the lane that owns
//     it, with the element-wise work that consumes the result done by that lane in the same phase. Few rows per lane keep
//     the reductions short (4 sums per product with M at 69 x 28, where a 16 x 16 grid had 7 sums over 16 lanes).
//   * phases of one working-set change (a barrier after each):  [A x | H x | A'y (drift correction) + A dx_FX | H dx_FX
//     -> right-hand side]  [M r -> dx_FR, dy_AC]  [A dx | H dx - A'dy -> dy_FX]  [ratio tests]  [homotopy step + k of the
//     change]  [u = M k]  ([A'xi: independence test of an incoming row])  [pivot, rank-1 update of M, working set].
//   * what the formulation does not carry -- a removal that would leave Z'HZ not positive definite (flipping bounds),
//     pivots of rounding size, an independence test it cannot decide, LPs, a free variable in the cold working set --
//     ends the kernel for that problem with RET_BAIL; the launcher then runs the null-space kernel (EngineX) on exactly
//     those members (it re-solves them from the same inputs: same results as before this engine existed).
// Validated first as a CPU prototype against the CPU restatement of the null-space method (tools/proto_k: 2000 random convex
// QPs and the 512-QP hs0xx batch identical in status, working sets and nWSR; non-convex inputs bail).

#define KSYNC() __syncthreads()

// a value every lane of the workgroup holds identically (result of a reduction, a decision derived from one). Moving such
// values to scalar registers (v_readfirstlane) turns the control flow that depends on them into scalar branches -- and was
// measured SLOWER on this kernel (69 x 28 members: 0.85 ms as is, 0.91 ms with the integer control values in SGPRs, 0.93 ms
// with the doubles too: the extra SGPR pressure spills, AGPR traffic grows from 88 to 187 registers), so it is a no-op
// unless built with -DRSQP_K_UNI
#ifdef RSQP_K_UNI
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
#else
__device__ __forceinline__ int uni(int v) { return v; }
__device__ __forceinline__ double uni(double v) { return v; }
#endif
template <int S> __device__ __forceinline__ double kmin_f64(double v) {        // minimum over the wave, every lane gets it
    if constexpr (S == 4) {
        const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
        return fmin(fmin(r0, r1), fmin(r2, r3));
    } else { v = fmin(v, xchg_f64<S>(v)); return kmin_f64<S + 1>(v); }
}
template <int S> __device__ __forceinline__ int kmin_i32(int v) {
    if constexpr (S == 4) {
        const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16), r2 = __builtin_amdgcn_readlane(v, 32),
                  r3 = __builtin_amdgcn_readlane(v, 48);
        return min(min(r0, r1), min(r2, r3));
    } else { v = min(v, xchg_i32<S>(v)); return kmin_i32<S + 1>(v); }
}

// GJ_ = lanes per row block: 8 (256 lanes, one wave per SIMD) or 16 (512 lanes, two waves per SIMD: half the matrix block per
// lane, and a wave's waits overlap with the other wave's issue)
template <int RV, int RC, int CV, int CC, int GJ_ = 8>
struct EngineK {
    static constexpr int GI = 32, GJ = GJ_, NT = GI * GJ, LGJ = GJ == 8 ? 3 : 4, NW = NT / 64;
    static_assert(GJ == 8 || GJ == 16, "row blocks of 8 or 16 lanes (inside a DPP row)");
    static constexpr int NVP = GI * RV > GJ * CV ? GI * RV : GJ * CV, NCP = GI * RC > GJ * CC ? GI * RC : GJ * CC;
    static constexpr int MAXV = GI * RV < GJ * CV ? GI * RV : GJ * CV, MAXC = GI * RC < GJ * CC ? GI * RC : GJ * CC;   // largest nV / nC
    static_assert(RV + RC <= GJ, "row owners of a row block must fit its 8 lanes");
    static constexpr int NVEC_V = 19, NVEC_C = 12;
    // LDS: vectors of NVP / NCP doubles, integer working sets, reduction slots, then the dense copies of A and H
    __host__ __device__ static long long lds_bytes(int nV, int nC) {
        return 8LL * (NVEC_V * NVP + NVEC_C * NCP + 16) + 4LL * (NVP + NCP + 16) + 8LL * ((long long)nC * nV + (long long)nV * nV);
    }
    // ---- LDS
    ldouble *x, *g, *lb, *ub, *gN, *lbN, *ubN, *dx, *yB, *dyB, *rV, *uV, *kV, *tV, *xiB, *aful;
    ldouble *gy, *pH, *hdv;    // carried: A'y_C - H x of the iterate, H dx_FX; H dx - A'dy_C of the last step direction
    ldouble *Ax, *lbA, *ubA, *lbAN, *ubAN, *dAx, *yC, *dyC, *rC, *uC, *kC;
    ldouble *pA;               // carried: A dx_FX
    ldouble *red;
    LDS int *Sb, *Sc, *ired;
    ldouble *Ad, *Hd;
    // ---- registers
    double Hb[RV][CV], Ab[RC][CV], Tb[RV][CC];
    double Mvv[RV][CV], Mvc[RV][CC], Mcv[RC][CV], Mcc[RC][CC];
    int nV, nC, tid, bi, bj, wave;
    int nFR, nAC, status, infeasible, bail_reason, parity;
    int debug_bail;             // >= 0: a hot start bails out before its debug_bail-th change (tests of the hand-over); else -1
    int since_refresh;          // working-set changes since the carried products were last formed exactly; >= REFRESH: do it now
    static constexpr int REFRESH = 8;
    double hscale;
    double wV[RV], wC[RC];     // one-hot: wV[a] = (bj == a), wC[c] = (bj - RV == c)
    long long tlast;   // (-DRSQP_STAMPS builds: cycles per phase of block 0, tools/stamp_k_kernel.py)

    __device__ __forceinline__ void carve(lchar *base, int nV_, int nC_) {
        nV = nV_; nC = nC_;
        tid = (int)threadIdx.x; bi = tid >> LGJ; bj = tid & (GJ - 1); wave = tid >> 6;
        ldouble *p = (ldouble *)base;
#define KV_(name) name = p; p += NVP
#define KC_(name) name = p; p += NCP
        KV_(x); KV_(g); KV_(lb); KV_(ub); KV_(gN); KV_(lbN); KV_(ubN); KV_(dx); KV_(yB); KV_(dyB); KV_(rV); KV_(uV); KV_(kV);
        KV_(tV); KV_(xiB); KV_(aful); KV_(gy); KV_(pH); KV_(hdv);
        KC_(Ax); KC_(lbA); KC_(ubA); KC_(lbAN); KC_(ubAN); KC_(dAx); KC_(yC); KC_(dyC); KC_(rC); KC_(uC); KC_(kC); KC_(pA);
#undef KV_
#undef KC_
        red = p; p += 16;
        LDS int *ip = (LDS int *)p;
        Sb = ip; ip += NVP; Sc = ip; ip += NCP; ired = ip; ip += 16;
        Ad = (ldouble *)ip;
        Hd = Ad + nC * nV;
        parity = 0;
#pragma unroll
        for (int a = 0; a < RV; a++) wV[a] = bj == a ? 1.0 : 0.0;
#pragma unroll
        for (int c = 0; c < RC; c++) wC[c] = bj - RV == c ? 1.0 : 0.0;
    }

    // ------------------------------------------------------------------ building blocks
    template <int R, int Cn> __device__ __forceinline__ static void mv(const double (&B)[R][Cn], const double (&xv)[Cn], double (&acc)[R]) {
#pragma unroll
        for (int a = 0; a < R; a++)
#pragma unroll
            for (int b = 0; b < Cn; b++) acc[a] = fma(B[a][b], xv[b], acc[a]);
    }
    template <int R, int Cn> __device__ __forceinline__ static void mvsub(const double (&B)[R][Cn], const double (&xv)[Cn], double (&acc)[R]) {
#pragma unroll
        for (int a = 0; a < R; a++)
#pragma unroll
            for (int b = 0; b < Cn; b++) acc[a] = fma(-B[a][b], xv[b], acc[a]);
    }
    template <int R> __device__ __forceinline__ static void rowsum(double (&acc)[R]) {     // over the 8 lanes of a row block
#pragma unroll
        for (int a = 0; a < R; a++) acc[a] = allreduce_sum<LGJ>(acc[a]);
    }
    template <int R> __device__ __forceinline__ static void zero(double (&acc)[R]) {
#pragma unroll
        for (int a = 0; a < R; a++) acc[a] = 0.0;
    }
    // row owners of a row block: lane bj < RV owns variable row bi RV + bj, lane RV <= bj < RV + RC constraint row bi RC + bj - RV
    __device__ __forceinline__ bool ownsV() const { return bj < RV; }
    __device__ __forceinline__ bool ownsC() const { return bj >= RV && bj < RV + RC; }
    __device__ __forceinline__ int rowV() const { return bi * RV + bj; }
    __device__ __forceinline__ int rowC() const { return bi * RC + (bj - RV); }
    // acc[k] for the lane-dependent k of a row owner, as the product with the lane's one-hot weights (wV / wC, set by carve):
    // exact (x * 1 + 0 * ...), and -- unlike a chain of selects, which the compiler turns back into an indexed load from a
    // copy of acc in SCRATCH memory (a ~500-cycle round trip in every phase: measured) -- it stays in registers
    template <int R> __device__ __forceinline__ static double pick(const double (&acc)[R], const double (&w)[R]) {
        double v = acc[0] * w[0];
#pragma unroll
        for (int a = 1; a < R; a++) v = fma(acc[a], w[a], v);
        return v;
    }
    template <int Cn> __device__ __forceinline__ void ldcols(const ldouble *v, double (&xv)[Cn]) const {
#pragma unroll
        for (int b = 0; b < Cn; b++) xv[b] = v[bj * Cn + b];
    }
    template <int R> __device__ __forceinline__ void ldrows(const ldouble *v, double (&xv)[R]) const {
#pragma unroll
        for (int a = 0; a < R; a++) xv[a] = v[bi * R + a];
    }
    // the other engines' delta_of(target, cur) = 0 when both are infinite, else target - cur. Here every target is clamped to
    // +-1e20 on entry and an infinite side of the current QP holds exactly that value (min / max with the relaxed limit), so
    // the plain difference IS that function: 1e20 - 1e20 = 0 -- and the two comparisons per call are gone
    __device__ __forceinline__ static double delta_of(double target, double cur) { return target - cur; }
    __device__ __forceinline__ static double clampinf(double v) { return v > RSQP_INFTY ? RSQP_INFTY : (v < -RSQP_INFTY ? -RSQP_INFTY : v); }
    // sums over the workgroup's vectors, formed by EVERY wave from the published LDS operands: all lanes agree, no barrier
    // (trip counts are compile-time constants: the loads of all trips are issued together)
    __device__ __forceinline__ double wsumV(const ldouble *a, const ldouble *b, double s) const {
        const int l = tid & 63;
#pragma unroll
        for (int k = 0; k < (NVP + 63) / 64; k++) { const int i = l + 64 * k; if (i < NVP) s = fma(a[i], b[i], s); }
        return s;
    }
    __device__ __forceinline__ double wsumC(const ldouble *a, const ldouble *b, double s) const {
        const int l = tid & 63;
#pragma unroll
        for (int k = 0; k < (NCP + 63) / 64; k++) { const int i = l + 64 * k; if (i < NCP) s = fma(a[i], b[i], s); }
        return s;
    }
    __device__ __forceinline__ double wdotV(const ldouble *a, const ldouble *b) const { return uni(allreduce_sum<6>(wsumV(a, b, 0.0))); }
    __device__ __forceinline__ double wdotVC(const ldouble *a, const ldouble *b, const ldouble *c, const ldouble *d) const {   // a'b + c'd
        return uni(allreduce_sum<6>(wsumC(c, d, wsumV(a, b, 0.0))));
    }
    // lexicographic minimum of (t, id) over the workgroup, ONE barrier (the exchange slots alternate): minimum of t over the
    // wave, then the lowest id among the lanes that hold it; the four wave results meet in LDS
    __device__ __forceinline__ void block_argmin(double &t, int &id) {
        const double tm = kmin_f64<0>(t);
        const int im = kmin_i32<0>(t == tm ? id : 0x7fffffff);
        parity ^= 8;
        if ((tid & 63) == 0) { red[parity + wave] = tm; ired[parity + wave] = im; }
        KSYNC();
        t = red[parity]; id = ired[parity];
#pragma unroll
        for (int w = 1; w < NW; w++) {
            const double t2 = red[parity + w]; const int id2 = ired[parity + w];
            if (t2 < t || (t2 == t && id2 < id)) { t = t2; id = id2; }
        }
        t = uni(t); id = uni(id);
    }

    // ------------------------------------------------------------------ staging
    __device__ __forceinline__ void stage(const int *gAjc, const int *gAir, const double *gAval, const int *gHjc, const int *gHir,
                                          const double *gHval, const double *g_, const double *lb_, const double *ub_,
                                          const double *lbA_, const double *ubA_) {
        for (int k = tid; k < nC * nV + nV * nV; k += NT) Ad[k] = 0.0;
        for (int v = tid; v < NVP; v += NT) {
            const bool in = v < nV;
            x[v] = 0.0; g[v] = 0.0; dx[v] = 0.0; yB[v] = 0.0; dyB[v] = 0.0; rV[v] = 0.0; uV[v] = 0.0; kV[v] = 0.0; tV[v] = 0.0; xiB[v] = 0.0;
            aful[v] = 0.0; gy[v] = 0.0; pH[v] = 0.0; hdv[v] = 0.0;
            gN[v] = in ? g_[v] : 0.0; lbN[v] = in ? clampinf(lb_[v]) : 0.0; ubN[v] = in ? clampinf(ub_[v]) : 0.0;
            lb[v] = 0.0; ub[v] = 0.0; Sb[v] = -1;
        }
        for (int i = tid; i < NCP; i += NT) {
            const bool in = i < nC;
            Ax[i] = 0.0; dAx[i] = 0.0; yC[i] = 0.0; dyC[i] = 0.0; rC[i] = 0.0; uC[i] = 0.0; kC[i] = 0.0; pA[i] = 0.0;
            lbAN[i] = in ? clampinf(lbA_[i]) : -RSQP_INFTY; ubAN[i] = in ? clampinf(ubA_[i]) : RSQP_INFTY;
            lbA[i] = -RSQP_INFTY; ubA[i] = RSQP_INFTY; Sc[i] = 0;
        }
        KSYNC();
        for (int c = tid; c < nV; c += NT) {
            for (int k = gAjc[c]; k < gAjc[c + 1]; k++) Ad[gAir[k] + c * nC] = gAval[k];
            for (int k = gHjc[c]; k < gHjc[c + 1]; k++) Hd[gHir[k] + c * nV] = gHval[k];
        }
        KSYNC();
#pragma unroll
        for (int a = 0; a < RV; a++) {
            const int r = bi * RV + a;
#pragma unroll
            for (int b = 0; b < CV; b++) {
                const int c = bj * CV + b;
                Hb[a][b] = (r < nV && c < nV) ? Hd[r + c * nV] : 0.0;
            }
#pragma unroll
            for (int c_ = 0; c_ < CC; c_++) {
                const int i = bj * CC + c_;                             // A' block: variable rows of bi, constraint columns of bj
                Tb[a][c_] = (i < nC && r < nV) ? Ad[i + r * nC] : 0.0;
            }
        }
#pragma unroll
        for (int c_ = 0; c_ < RC; c_++) {
            const int i = bi * RC + c_;
#pragma unroll
            for (int b = 0; b < CV; b++) {
                const int v = bj * CV + b;                              // A block: constraint rows of bi, variable columns of bj
                Ab[c_][b] = (i < nC && v < nV) ? Ad[i + v * nC] : 0.0;
            }
        }
        double hm = 0.0;
        for (int v = tid & 63; v < nV; v += 64) hm = fmax(hm, fabs(Hd[v + v * nV]));
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) hm = fmax(hm, __shfl_xor(hm, s));
        // H has to be symmetric (products run over rows, borderings read columns); some of the reference's own inputs
        // (test/unsolved_QPs/*.hpp) are not -- those go to the null-space kernel. Every lane checks its own block.
        int asym = 0;
#pragma unroll
        for (int a = 0; a < RV; a++)
#pragma unroll
            for (int b = 0; b < CV; b++) {
                const int r = bi * RV + a, c = bj * CV + b;
                if (r < nV && c < nV && Hb[a][b] != Hd[c + r * nV]) asym = 1;
            }
        // (the workgroup-wide OR through this kernel's own LDS slots: __syncthreads_or brings static LDS, and with it the
        //  160 KB dynamic-LDS attribute of the kernel is refused)
        const int wany = __any(asym) ? 1 : 0;
        if ((tid & 63) == 0) ired[wave] = wany;
        KSYNC();
        int any = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) any |= ired[w];
        KSYNC();
        hscale = any ? 0.0 : hm;      // (hscale = 0 makes the kernel bail)
    }
    __device__ __forceinline__ bool bounds_inconsistent() const {
        double bad = 0.0;
        for (int v = tid & 63; v < nV; v += 64) if (lbN[v] > ubN[v] + RSQP_EPS) bad += 1.0;
        for (int i = tid & 63; i < nC; i += 64) if (lbAN[i] > ubAN[i] + RSQP_EPS) bad += 1.0;
        return allreduce_sum<6>(bad) > 0.0;
    }

    // cold start: every variable on a finite bound (lower first), no constraint active; x = 0, y = 0
    __device__ __forceinline__ int setup_cold() {
        double nofin = 0.0;
        for (int v = tid & 63; v < nV; v += 64) if (lbN[v] <= -RSQP_INFTY && ubN[v] >= RSQP_INFTY) nofin += 1.0;
        if (allreduce_sum<6>(nofin) > 0.0) { bail_reason = 11; return RET_BAIL; }   // a free variable in the cold working set
        for (int v = tid; v < NVP; v += NT) {
            const int s = (v >= nV || lbN[v] > -RSQP_INFTY) ? -1 : 1;
            Sb[v] = s;
            double l = s == -1 ? 0.0 : fmin(lbN[v], -RSQP_BOUND_RELAXATION), u = s == 1 ? 0.0 : fmax(ubN[v], RSQP_BOUND_RELAXATION);
            if (v >= nV) { l = 0.0; u = 0.0; }
            lb[v] = l; ub[v] = u;
            dx[v] = s == -1 ? delta_of(lbN[v], l) : delta_of(ubN[v], u);       // dx on the fixed variables = the move of their bounds
        }
        for (int i = tid; i < nC; i += NT) { lbA[i] = fmin(lbAN[i], -RSQP_BOUND_RELAXATION); ubA[i] = fmax(ubAN[i], RSQP_BOUND_RELAXATION); }
        nFR = nAC = 0;
        KSYNC();
        return RET_OK;
    }

    // ------------------------------------------------------------------ phase: drift correction + right-hand side, EXACT
    // (every REFRESH working-set changes, after an exchange, and at the start; in between these four products follow the
    // iterate by their increments -- the scheme of the other engines -- and the phase is the element-wise tail of enter_leave)
    // A x, H x, A'y_C of the iterate (x already exactly on its active bounds, see the end of enter_leave): active constraint
    // limits := A x, gradient of the current QP from stationarity; A dx_FX, H dx_FX (dx holds the move of the active bounds) ->
    // right-hand side r of the KKT system of the step
    // keep_data (first phase of a hot start): g, A x and the limits of the active constraints stay what the previous solve left
    // (its targets, exactly), as in a hot start of the other engines; only the carried products and the right-hand side are formed
    __device__ __forceinline__ void drift_and_rhs(bool keep_data) {
        double xv[CV], dv[CV], yc[CC], ax[RC], ad[RC], gyr[RV], hd[RV];
        ldcols<CV>(x, xv); ldcols<CV>(dx, dv); ldcols<CC>(yC, yc);
        zero<RC>(ax); zero<RC>(ad); zero<RV>(gyr); zero<RV>(hd);
        mv<RC, CV>(Ab, xv, ax); mv<RC, CV>(Ab, dv, ad);
        mv<RV, CC>(Tb, yc, gyr); mvsub<RV, CV>(Hb, xv, gyr);      // A'y_C - H x
        mv<RV, CV>(Hb, dv, hd);
        rowsum<RC>(ax); rowsum<RC>(ad); rowsum<RV>(gyr); rowsum<RV>(hd);
        if (ownsV()) {
            const int v = rowV();
            const double gyv = pick<RV>(gyr, wV), phv = pick<RV>(hd, wV), gv = keep_data ? g[v] : gyv + yB[v];
            gy[v] = keep_data ? gv - yB[v] : gyv; pH[v] = phv;
            g[v] = gv;
            rV[v] = Sb[v] == 0 ? -((gN[v] - gv) + phv) : 0.0;
        } else if (ownsC()) {
            const int i = rowC(), s = Sc[i];
            const double a = keep_data ? Ax[i] : pick<RC>(ax, wC), pa = pick<RC>(ad, wC);
            Ax[i] = a; pA[i] = pa;
            // (lbA | ubA and lbAN | ubAN are adjacent arrays: the side picks an OFFSET -- a select between the two pointers
            //  becomes a table of LDS addresses in scratch memory and a ~500-cycle load from it)
            const int side_off = i + (s == 1 ? NCP : 0);
            const double lim = keep_data ? lbA[side_off] : a;
            if (s != 0) lbA[side_off] = lim;
            rC[i] = s != 0 ? (lbAN[side_off] - lim) - pa : 0.0;
        }
        since_refresh = 0;
    }
    // ------------------------------------------------------------------ products with M
    // (outV, outC) = M (inV, inC); the caller closes with the barrier. SD: the step-direction epilogue (dx on the free variables,
    // dy of the active constraints) instead of the plain store; patch: -1 is added to the slot (pisc, pid) of the result
    template <bool SD> __device__ __forceinline__ void m_times(const ldouble *inV, const ldouble *inC, bool patch, bool pisc, int pid) {
        double xv[CV], xc[CC], sv[RV], sc[RC];
        ldcols<CV>(inV, xv); ldcols<CC>(inC, xc);
        zero<RV>(sv); zero<RC>(sc);
        mv<RV, CV>(Mvv, xv, sv); mv<RV, CC>(Mvc, xc, sv);
        mv<RC, CV>(Mcv, xv, sc); mv<RC, CC>(Mcc, xc, sc);
        rowsum<RV>(sv); rowsum<RC>(sc);
        if (ownsV()) {
            const int r = rowV();
            const double val = pick<RV>(sv, wV);
            if constexpr (SD) { if (Sb[r] == 0) dx[r] = val; }
            else uV[r] = val - ((patch && !pisc && r == pid) ? 1.0 : 0.0);
        } else if (ownsC()) {
            const int r = rowC();
            const double val = pick<RC>(sc, wC);
            if constexpr (SD) dyC[r] = Sc[r] != 0 ? -val : 0.0;
            else uC[r] = val - ((patch && pisc && r == pid) ? 1.0 : 0.0);
        }
    }
    // M += coef * (uV; uC)(uV; uC)' on this lane's block
    __device__ __forceinline__ void rank1(double coef) {
        double rv[RV], rc[RC], cv[CV], cc[CC];
        ldrows<RV>(uV, rv); ldrows<RC>(uC, rc); ldcols<CV>(uV, cv); ldcols<CC>(uC, cc);
#pragma unroll
        for (int a = 0; a < RV; a++) {
            const double t = coef * rv[a];
#pragma unroll
            for (int b = 0; b < CV; b++) Mvv[a][b] = fma(t, cv[b], Mvv[a][b]);
#pragma unroll
            for (int c_ = 0; c_ < CC; c_++) Mvc[a][c_] = fma(t, cc[c_], Mvc[a][c_]);
        }
#pragma unroll
        for (int c_ = 0; c_ < RC; c_++) {
            const double t = coef * rc[c_];
#pragma unroll
            for (int b = 0; b < CV; b++) Mcv[c_][b] = fma(t, cv[b], Mcv[c_][b]);
#pragma unroll
            for (int d_ = 0; d_ < CC; d_++) Mcc[c_][d_] = fma(t, cc[d_], Mcc[c_][d_]);
        }
    }

    // ------------------------------------------------------------------ elementary operations on S
    // One routine for the four elementary operations, so that the product with M and the rank-1 update exist ONCE in the
    // kernel (inlined per call site the kernel was 73 KB of code -- more than the 64 KB instruction cache two CUs share):
    //   OP_LEAVE_C  constraint i leaves S:  u = M e_i,        M -= u u' / u_i                      (Schur complement step)
    //   OP_ENTER_B  variable v gets fixed:  u = M e_v,        M -= u u' / u_v
    //   OP_LEAVE_B  variable v is freed:    u = M k - e_v,    M += u u' / (H_vv - k'u),  k = column v of [H; A] on S
    //   OP_ENTER_C  constraint i is added:  u = M [a; 0] - e_i,  M += u u' / (-a'u),     a = free part of row i
    // With u' = u - e_p the bordering [[M + u u'/s, -u/s], [-u'/s, 1/s]] IS the rank-1 update M + u' u''/s (row / column p
    // of M are zero before). A slot that left S keeps rounding-size residue in its row and column instead of exact zeros
    // (M_qp - m_q (m_p / m_p)): every consumer masks by the working set, and a later bordering of that slot absorbs it.
    enum { OP_LEAVE_C = 0, OP_LEAVE_B = 1, OP_ENTER_C = 2, OP_ENTER_B = 3 };
    __device__ __forceinline__ void build_k(int op, int id) {
        const bool isc = op == OP_LEAVE_C || op == OP_ENTER_C;
        // (one predicated trip: NVP + NCP <= NT; the variables on the lower lanes, the constraints on the upper waves)
        static_assert(NVP <= NT / 2 && NCP <= NT / 2, "build_k lane map");
        const int u = tid, j = tid - NT / 2;
        if (op == OP_LEAVE_B) {
            if (u < NVP) kV[u] = (u < nV && Sb[u] == 0) ? Hd[(u < nV ? u : 0) + id * nV] : 0.0;
            if (j >= 0 && j < NCP) kC[j] = (j < nC && Sc[j] != 0) ? Ad[(j < nC ? j : 0) + id * nC] : 0.0;
        } else if (op == OP_ENTER_C) {
            if (u < NVP) { const double a = u < nV ? Ad[id + (u < nV ? u : 0) * nC] : 0.0; aful[u] = a; kV[u] = Sb[u] == 0 ? a : 0.0; }
            if (j >= 0 && j < NCP) kC[j] = 0.0;
        } else {
            if (u < NVP) { const double a = (!isc && u == id) ? 1.0 : 0.0; kV[u] = a; aful[u] = a; }
            if (j >= 0 && j < NCP) kC[j] = (isc && j == id) ? 1.0 : 0.0;
        }
    }
    // k of (op0, id0) is published (build_k by the caller, before its barrier)
    __device__ __forceinline__ int enter_leave(int op0, int id0, int side) {
        int op = op0, id = id0;
        bool li_known = false;
        double ynew = 0.0;
        for (;;) {
            op = uni(op); id = uni(id);
            const bool isc = op == OP_LEAVE_C || op == OP_ENTER_C, unit = op == OP_LEAVE_C || op == OP_ENTER_B;
            const int sb_old = isc ? 0 : uni(Sb[id]);      // (read before the barrier below: the owner rewrites it at the end of the pass)
            m_times<false>(kV, kC, !unit, isc, id);
            KSYNC();
            STAMP(37);
            if ((op == OP_ENTER_C || op == OP_ENTER_B) && !li_known) {
                // independence of the incoming row a from the working set, from the residual of its representation by the
                // active rows: r = a_FR - A_AC,FR' xi_C with xi_C = the constraint part of M [a; 0]. K M = I gives r = H P a
                // and Z'r = Z'a, so |Z'a| <= |r| <= cond(Z'HZ) |Z'a|: a first-order quantity like the |Z'a| > 1e-9 |a| of the
                // null-space engines (a'Pa is of second order and drowns in rounding below ~1e-7)
                {
                    double yc[CC], at[RV];
#pragma unroll
                    for (int c_ = 0; c_ < CC; c_++) { const int i = bj * CC + c_; yc[c_] = Sc[i] != 0 ? uC[i] : 0.0; }
                    zero<RV>(at);
                    mv<RV, CC>(Tb, yc, at);
                    rowsum<RV>(at);
                    if (ownsV()) {
                        const int v = rowV();
                        const double r = aful[v] - pick<RV>(at, wV);
                        const bool fr = Sb[v] == 0;
                        tV[v] = fr ? r : 0.0;       // residual on the free variables
                        xiB[v] = fr ? 0.0 : r;      // coefficients of the fixed variables in the dependency
                    }
                }
                KSYNC();
                const double rn2 = wdotV(tV, tV), na2 = wdotV(kV, kV);
                int li;
                if (nFR - nAC <= 0 || !(na2 > 0.0)) li = 0;
                else { const double rel = sqrt(rn2 / na2); li = rel > 1e-7 ? 1 : (rel < 1e-9 ? 0 : -1); }
                if (li < 0) { bail_reason = 3; return RET_BAIL; }
                li_known = true;
                STAMP(38);
                if (li == 0) {
                    // ---- exchange: shift the multipliers along the dependency until one of them reaches zero; that one leaves
                    const double sgn = side == 1 ? -1.0 : 1.0;
                    double bt = RSQP_INFTY;
                    int bid = 0x7fffffff;
                    for (int it = tid; it < nC + nV; it += NT) {
                        const bool c_ = it < nC;
                        const int i = c_ ? it : it - nC;
                        const int s = c_ ? Sc[i] : Sb[i];
                        if (s != 0) {
                            const double xi = sgn * (c_ ? uC[i] : xiB[i]), yi = c_ ? yC[i] : yB[i];
                            const double num = s == -1 ? yi : -yi, den = s == -1 ? xi : -xi;
                            if (den > RSQP_EPS_DEN) {
                                const double t = (num > 0.0 ? num : 0.0) / den;
                                if (t < bt || (t == bt && it < bid)) { bt = t; bid = it; }
                            }
                        }
                    }
                    block_argmin(bt, bid);
                    if (bid == 0x7fffffff) return RET_INFEASIBLE;
                    for (int i = tid; i < nC; i += NT) if (Sc[i] != 0) yC[i] -= bt * sgn * uC[i];
                    for (int v = tid; v < nV; v += NT) if (Sb[v] != 0) yB[v] -= bt * sgn * xiB[v];
                    ynew = sgn * bt;
                    if (bid < nC) { op = OP_LEAVE_C; id = bid; } else { op = OP_LEAVE_B; id = bid - nC; }
                    build_k(op, id);    // (writes kV / kC / aful: every wave is past its dots -- block_argmin's barrier)
                    KSYNC();
                    STAMP(39);
                    continue;           // the partner leaves first; then the incoming row is taken up again
                }
            }
            // ---- pivot and rank-1 update
            double coef;
            if (unit) {
                const double mu = uni(isc ? uC[id] : uV[id]);
                bool ok;
                if (isc) { const double d2 = wdotV(uV, uV); ok = d2 > 0.0 && -mu > 1e-8 * hscale * d2; }   // the released direction has curvature -mu
                else ok = mu > 1e-10 / hscale;
                if (!ok) { bail_reason = op == OP_LEAVE_C ? (op0 == op ? 1 : 4) : 7; return RET_BAIL; }
                coef = -1.0 / mu;
            } else {
                const double ku = wdotVC(kV, uV, kC, uC);       // (k is zero in the patched slot)
                const double kappa = op == OP_LEAVE_B ? uni(Hd[id + id * nV]) : 0.0, sigma = kappa - ku;
                bool ok;
                if (op == OP_LEAVE_B) ok = sigma > 1e-8 * hscale;                       // Z'HZ stays positive definite
                else { const double na2 = wdotV(kV, kV); ok = -sigma > 1e-10 * na2 / hscale; }
                if (!ok) { bail_reason = op == OP_LEAVE_B ? (op0 == op ? 2 : 5) : 6; return RET_BAIL; }
                coef = 1.0 / sigma;
            }
            STAMP(41);
            rank1(coef);
            STAMP(42);
            const bool partner = op != op0 || id != id0;
            if (partner || op != op0) since_refresh = REFRESH;     // an exchange moved the multipliers outside a step: exact products next
            const bool tail = !partner && since_refresh + 1 < REFRESH;   // the element-wise drift correction + right-hand side rides here
            // a variable changed sides of the working set: its move d leaves (freed) or enters (fixed) dx_FX, and with it
            // d x column of A / H the carried products A dx_FX / H dx_FX
            double dchg = 0.0;
            if (tail && !isc) {
                const int so = id + ((op == OP_LEAVE_B ? sb_old : side) == 1 ? NVP : 0);      // lb | ub, lbN | ubN are adjacent
                const double d = uni(lbN[so] - lb[so]);
                dchg = op == OP_LEAVE_B ? -d : d;
            }
            // the working set, and -- behind the last operation -- x exactly on its active bounds, dx_FX of the next step
            if (ownsV()) {
                const int v = rowV();
                int s = Sb[v];
                if (!isc && v == id) {
                    s = op == OP_LEAVE_B ? 0 : side;
                    Sb[v] = s;
                    yB[v] = op == OP_LEAVE_B ? 0.0 : ynew;
                }
                if (!partner) {
                    const double l = lb[v], u = ub[v];
                    if (s != 0) x[v] = s == -1 ? l : u;
                    dx[v] = s == -1 ? delta_of(lbN[v], l) : (s == 1 ? delta_of(ubN[v], u) : 0.0);
                    if (tail) {
                        double ph = pH[v];
                        if (!isc && v < nV) { ph = fma(Hd[v + id * nV], dchg, ph); pH[v] = ph; }
                        const double gv = gy[v] + yB[v];
                        g[v] = gv;
                        rV[v] = s == 0 ? -((gN[v] - gv) + ph) : 0.0;
                    }
                }
            } else if (ownsC()) {
                const int i = rowC();
                int s = Sc[i];
                if (isc && i == id) { s = op == OP_LEAVE_C ? 0 : side; Sc[i] = s; yC[i] = op == OP_LEAVE_C ? 0.0 : ynew; }
                if (tail) {
                    double pa = pA[i];
                    if (!isc && i < nC) { pa = fma(Ad[i + id * nC], dchg, pa); pA[i] = pa; }
                    const double a = Ax[i];
                    const int side_off = i + (s == 1 ? NCP : 0);
                    if (s != 0) lbA[side_off] = a;
                    rC[i] = s != 0 ? (lbAN[side_off] - a) - pa : 0.0;
                }
            }
            if (op == OP_LEAVE_C) nAC--; else if (op == OP_LEAVE_B) nFR++; else if (op == OP_ENTER_C) nAC++; else nFR--;
            since_refresh++;
            if (!partner) break;
            KSYNC();
            op = op0; id = id0;
            build_k(op, id);
            KSYNC();
        }
        KSYNC();
        STAMP(40);
        return RET_OK;
    }

    // ------------------------------------------------------------------ ratio tests (as the other engines)
    __device__ __forceinline__ static void cand(double num, double den, int id, double &bt, int &bid) {
        if (den >= RSQP_EPS_DEN) {
            const double t = (num > 0.0 ? num : 0.0) / den;
            if (t < bt || (t == bt && id < bid)) { bt = t; bid = id; }
        }
    }
    // one candidate per lane, classes aligned to the waves: the first 2 NCP lanes = the constraints (lower side / multiplier of
    // an active one, then upper side), the next 2 NVP = the variables -- the arrays a lane reads are wave-uniform, the
    // side picks an offset into adjacent arrays (no per-lane pointer selects)
    static_assert(2 * (NCP + NVP) <= NT && (2 * NCP) % 64 == 0, "ratio-test lane map: one lane per candidate, the constraints fill whole waves");
    __device__ __forceinline__ Blocking ratio_tests() {
        double bt = 1.0;
        int bid = 0x7fffffff;
        if (tid < 2 * NCP) {
            const bool upper = tid >= NCP;
            const int i = upper ? tid - NCP : tid;
            if (i < nC) {
                const int s = Sc[i], o = i + (upper ? NCP : 0);
                const double cur = Ax[i], d = dAx[i], yi = yC[i], dyi = dyC[i], bnd = lbA[o], bndN = lbAN[o];
                const bool act = s != 0;
                const double sg = upper ? -1.0 : 1.0;
                const double num = act ? (s == -1 ? yi : -yi) : sg * (cur - bnd);
                const double den = act ? (s == -1 ? -dyi : dyi) : sg * ((bndN - bnd) - d);
                const int id = act ? i : (upper ? 2 * nC + nV : nC + nV) + i;
                const bool ok = act ? !upper : (upper ? bndN < RSQP_INFTY : bndN > -RSQP_INFTY);
                if (ok) cand(num, den, id, bt, bid);
            }
        } else {
            const int t2 = tid - 2 * NCP;
            const bool upper = t2 >= NVP;
            const int v = upper ? t2 - NVP : t2;
            if (v < nV) {
                const int s = Sb[v], o = v + (upper ? NVP : 0);
                const double cur = x[v], d = dx[v], yi = yB[v], dyi = dyB[v], bnd = lb[o], bndN = lbN[o];
                const bool act = s != 0;
                const double sg = upper ? -1.0 : 1.0;
                const double num = act ? (s == -1 ? yi : -yi) : sg * (cur - bnd);
                const double den = act ? (s == -1 ? -dyi : dyi) : sg * ((bndN - bnd) - d);
                const int id = act ? nC + v : (upper ? 3 * nC + 2 * nV : 3 * nC + nV) + v;
                const bool ok = act ? !upper : (upper ? bndN < RSQP_INFTY : bndN > -RSQP_INFTY);
                if (ok) cand(num, den, id, bt, bid);
            }
        }
        if (!(bt < 1.0)) { bt = 1.0; bid = 0x7fffffff; }
        STAMP(43);
        block_argmin(bt, bid);
        STAMP(44);
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

    __device__ __forceinline__ int homotopy(int maxit, int &nWSR, bool hot) {
        int iter = 0, rcode = RET_OK;
        status = QPS_PERFORMINGHOMOTOPY;
        since_refresh = REFRESH;
        for (;;) {
            if (since_refresh >= REFRESH) {
                drift_and_rhs(hot && iter == 0);
                KSYNC();
            }
            STAMP(30);
            m_times<true>(rV, rC, false, false, 0);            // (dx_FR ; -dy_AC) = M r
            KSYNC();
            STAMP(31);
            {   // A dx (ratio tests); H dx + dg - A'dy_C = multipliers of the fixed variables
                double xv[CV], yc[CC], aa[RC], hd[RV];
                ldcols<CV>(dx, xv); ldcols<CC>(dyC, yc);
                zero<RC>(aa); zero<RV>(hd);
                mv<RC, CV>(Ab, xv, aa); mv<RV, CV>(Hb, xv, hd); mvsub<RV, CC>(Tb, yc, hd);
                rowsum<RC>(aa); rowsum<RV>(hd);
                if (ownsV()) { const int v = rowV(); const double h = pick<RV>(hd, wV); hdv[v] = h; dyB[v] = Sb[v] != 0 ? h + (gN[v] - g[v]) : 0.0; }
                else if (ownsC()) dAx[rowC()] = pick<RC>(aa, wC);
            }
            KSYNC();
            STAMP(32);
            Blocking b = ratio_tests();
            b.kind = uni(b.kind); b.idx = uni(b.idx); b.side = uni(b.side);
            STAMP(33);
            const double tau = uni(b.tau);
            const bool done = b.kind == 0;
            // homotopy step; the blocking quantity sits exactly on its limit (its owner writes that); k of the change
            if (tid < nV) {                                    // (nV <= MAXV < NT / 2)
                const int v = tid;
                const int s = Sb[v];
                if (done) {
                    g[v] = gN[v]; lb[v] = lbN[v]; ub[v] = ubN[v];
                    x[v] = s == -1 ? lbN[v] : (s == 1 ? ubN[v] : x[v] + tau * dx[v]);
                } else {
                    const double xn = x[v] + tau * dx[v];
                    x[v] = xn;
                    g[v] += tau * (gN[v] - g[v]);
                    gy[v] -= tau * hdv[v];                      // A'y_C - H x follows the step
                    pH[v] *= 1.0 - tau;                         // the active bounds have 1 - tau of their way left
                    const double l = lb[v] + tau * delta_of(lbN[v], lb[v]), u = ub[v] + tau * delta_of(ubN[v], ub[v]);
                    lb[v] = (b.kind == 4 && b.side == -1 && v == b.idx) ? xn : l;
                    ub[v] = (b.kind == 4 && b.side == 1 && v == b.idx) ? xn : u;
                }
                yB[v] += tau * dyB[v];
            }
            if (tid >= NT / 2 && tid - NT / 2 < nC) {           // (the upper waves: the lower ones hold the variables)
                const int i = tid - NT / 2;
                yC[i] += tau * dyC[i];
                if (done) { lbA[i] = lbAN[i]; ubA[i] = ubAN[i]; }
                else {
                    const double an = Ax[i] + tau * dAx[i];     // A x follows the step
                    Ax[i] = an;
                    pA[i] *= 1.0 - tau;
                    const double l = lbA[i] + tau * delta_of(lbAN[i], lbA[i]), u = ubA[i] + tau * delta_of(ubAN[i], ubA[i]);
                    lbA[i] = (b.kind == 3 && b.side == -1 && i == b.idx) ? an : l;
                    ubA[i] = (b.kind == 3 && b.side == 1 && i == b.idx) ? an : u;
                }
            }
            if (done || iter >= maxit) {
                // A x of the final iterate, exactly (the carried copy is one step behind at a full step): the next hot start
                // reads it in its ratio tests
                KSYNC();
                double xv[CV], aa[RC];
                ldcols<CV>(x, xv);
                zero<RC>(aa);
                mv<RC, CV>(Ab, xv, aa);
                rowsum<RC>(aa);
                if (ownsC()) Ax[rowC()] = pick<RC>(aa, wC);
                KSYNC();
                if (done) status = QPS_SOLVED; else rcode = RET_MAX_NWSR;
                break;
            }
            const int op = b.kind == 1 ? OP_LEAVE_C : (b.kind == 2 ? OP_LEAVE_B : (b.kind == 3 ? OP_ENTER_C : OP_ENTER_B));
            build_k(op, b.idx);
            KSYNC();
            STAMP(34);
            if (hot && iter == debug_bail) { bail_reason = 13; rcode = RET_BAIL; break; }      // (test hook, see P.k_debug_bail)
            rcode = enter_leave(op, b.idx, b.side);
            if (rcode == RET_INFEASIBLE) { infeasible = 1; break; }
            if (rcode != RET_OK) break;
            iter++;
        }
        nWSR = iter;
        return rcode;
    }
    // ------------------------------------------------------------------ persistent state (hot starts)
    // The state lives in the layout of the explicit-inverse engine (qp_small_x.h carve: factors | x g lb ub | A x lbA ubA | y,
    // then Sb Sc AC posAC iscal as ints), so that engine's hot-start modes work on it unchanged (new matrices / warm re-init
    // rebuild their factors anyway; a plain hot start rebuilds them when iscal[4] says the factors are not its own) -- and
    // M = K^-1 goes BEHIND that image, one slot per variable and constraint ((nV + nC)^2 doubles, rsqp_state_bytes).
    __device__ __forceinline__ void store_state(double *img) const {
        const long long ldx = rsqp_ld(nV), sT = nV < nC ? nV : nC, voff = 2 * ldx * nV + sT * ldx;
        double *pv = img + voff;
        if (tid < nV) { const int v = tid; pv[v] = x[v]; pv[nV + v] = g[v]; pv[2 * nV + v] = lb[v]; pv[3 * nV + v] = ub[v]; pv[4 * nV + 3 * nC + v] = yB[v]; }
        if (tid >= NT / 2 && tid - NT / 2 < nC) {
            const int i = tid - NT / 2;
            pv[4 * nV + i] = Ax[i]; pv[4 * nV + nC + i] = lbA[i]; pv[4 * nV + 2 * nC + i] = ubA[i]; pv[5 * nV + 3 * nC + i] = yC[i];
        }
        int *pi = reinterpret_cast<int *>(img + voff + 5LL * nV + 4LL * nC);      // = persist_doubles of that engine
        if (tid < nV) pi[tid] = Sb[tid];
        if (tid >= NT / 2 && tid - NT / 2 < nC) pi[nV + tid - NT / 2] = Sc[tid - NT / 2];
        if (tid == 0) { int *isc = pi + nV + 3 * nC; isc[1] = nFR; isc[2] = nAC; isc[3] = status; isc[4] = 1; }
        double *pm = img + rsqp_image_bytes(nV, nC) / 8;
        const int N = nV + nC;
#pragma unroll
        for (int a = 0; a < RV; a++) {
            const int r = bi * RV + a;
            if (r < nV) {
#pragma unroll
                for (int b = 0; b < CV; b++) { const int c = bj * CV + b; if (c < nV) pm[r + c * N] = Mvv[a][b]; }
#pragma unroll
                for (int c_ = 0; c_ < CC; c_++) { const int c = bj * CC + c_; if (c < nC) pm[r + (nV + c) * N] = Mvc[a][c_]; }
            }
        }
#pragma unroll
        for (int c_ = 0; c_ < RC; c_++) {
            const int r = bi * RC + c_;
            if (r < nC) {
#pragma unroll
                for (int b = 0; b < CV; b++) { const int c = bj * CV + b; if (c < nV) pm[nV + r + c * N] = Mcv[c_][b]; }
#pragma unroll
                for (int d_ = 0; d_ < CC; d_++) { const int c = bj * CC + d_; if (c < nC) pm[nV + r + (nV + c) * N] = Mcc[c_][d_]; }
            }
        }
    }
    // M: zero (cold start) or the stored factor (hot start; pm = the extension behind the null-space image). ONE definition
    // site for the register blocks: with a second one (zeros in stage(), loads in load_state()) the register allocator kept
    // 204 instead of ~100 AGPRs busy through the whole kernel
    __device__ __forceinline__ void init_M(const double *pm) {
        const int N = nV + nC;
        const bool ld_ = pm != nullptr;
#pragma unroll
        for (int a = 0; a < RV; a++) {
            const int r = bi * RV + a;
#pragma unroll
            for (int b = 0; b < CV; b++) { const int c = bj * CV + b; Mvv[a][b] = (ld_ && r < nV && c < nV) ? pm[r + c * N] : 0.0; }
#pragma unroll
            for (int c_ = 0; c_ < CC; c_++) { const int c = bj * CC + c_; Mvc[a][c_] = (ld_ && r < nV && c < nC) ? pm[r + (nV + c) * N] : 0.0; }
        }
#pragma unroll
        for (int c_ = 0; c_ < RC; c_++) {
            const int r = bi * RC + c_;
#pragma unroll
            for (int b = 0; b < CV; b++) { const int c = bj * CV + b; Mcv[c_][b] = (ld_ && r < nC && c < nV) ? pm[nV + r + c * N] : 0.0; }
#pragma unroll
            for (int d_ = 0; d_ < CC; d_++) { const int c = bj * CC + d_; Mcc[c_][d_] = (ld_ && r < nC && c < nC) ? pm[nV + r + (nV + c) * N] : 0.0; }
        }
    }
    // hot start: false = the stored state is not one this kernel wrote (the caller bails: the null-space kernel takes the member)
    __device__ __forceinline__ bool load_state(const double *img) {
        const long long ldx = rsqp_ld(nV), sT = nV < nC ? nV : nC, voff = 2 * ldx * nV + sT * ldx;
        const double *pv = img + voff;
        const int *pi = reinterpret_cast<const int *>(img + voff + 5LL * nV + 4LL * nC);
        const int *isc = pi + nV + 3 * nC;
        if (isc[4] != 1 || isc[3] == QPS_NOTINITIALISED) return false;
        nFR = isc[1]; nAC = isc[2]; status = isc[3];
        if (tid < nV) { const int v = tid; x[v] = pv[v]; g[v] = pv[nV + v]; lb[v] = pv[2 * nV + v]; ub[v] = pv[3 * nV + v]; yB[v] = pv[4 * nV + 3 * nC + v]; Sb[v] = pi[v]; }
        if (tid >= NT / 2 && tid - NT / 2 < nC) {
            const int i = tid - NT / 2;
            Ax[i] = pv[4 * nV + i]; lbA[i] = pv[4 * nV + nC + i]; ubA[i] = pv[4 * nV + 2 * nC + i]; yC[i] = pv[5 * nV + 3 * nC + i]; Sc[i] = pi[nV + i];
        }
        KSYNC();
        // (as the homotopy of the other engines begins) an inactive side that was infinite and now has a finite target only has
        // to stay clear of the iterate; then dx on the fixed variables = the move of their bounds
        if (tid < NVP) {
            const int v = tid;
            const int s = Sb[v];
            if (v < nV) {
                if (s != -1 && lb[v] <= -RSQP_INFTY && lbN[v] > -RSQP_INFTY) lb[v] = fmin(lbN[v], x[v] - RSQP_BOUND_RELAXATION);
                if (s != 1 && ub[v] >= RSQP_INFTY && ubN[v] < RSQP_INFTY) ub[v] = fmax(ubN[v], x[v] + RSQP_BOUND_RELAXATION);
            }
            dx[v] = s == -1 ? lbN[v] - lb[v] : (s == 1 ? ubN[v] - ub[v] : 0.0);
        }
        if (tid >= NT / 2 && tid - NT / 2 < nC) {
            const int i = tid - NT / 2, s = Sc[i];
            if (s != -1 && lbA[i] <= -RSQP_INFTY && lbAN[i] > -RSQP_INFTY) lbA[i] = fmin(lbAN[i], Ax[i] - RSQP_BOUND_RELAXATION);
            if (s != 1 && ubA[i] >= RSQP_INFTY && ubAN[i] < RSQP_INFTY) ubA[i] = fmax(ubAN[i], Ax[i] + RSQP_BOUND_RELAXATION);
        }
        KSYNC();
        return true;
    }

    // 0.5 x'Hx + gN'x (H x by one product, into tV)
    __device__ __forceinline__ double objective() {
        double xv[CV], ah[RV];
        ldcols<CV>(x, xv);
        zero<RV>(ah);
        mv<RV, CV>(Hb, xv, ah);
        rowsum<RV>(ah);
        if (ownsV()) tV[rowV()] = pick<RV>(ah, wV);
        KSYNC();
        return 0.5 * wdotV(x, tV) + wdotV(gN, x);
    }
};

// STATEFUL = false: cold starts of batches that keep no hot-start state -- a build without the state I/O (not instantiated:
// it measures the same); true: cold starts that may leave a state behind, and hot starts on it
template <int RV, int RC, int CV, int CC, int GJ_, bool STATEFUL>
__global__ void __launch_bounds__(32 * GJ_, 1) small_qpk_kernel(QPPools P, int nq, int mode_in, int maxWSR) {
    const int mode = STATEFUL ? mode_in : 0;
    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    const int q = (int)blockIdx.x;
    if (q >= nq) return;
    const QPDesc d = P.desc[q];
    typedef EngineK<RV, RC, CV, CC, GJ_> ENG;
    ENG E;
    E.carve((lchar *)smem_generic, d.nV, d.nC);
    E.nFR = E.nAC = 0; E.status = QPS_NOTINITIALISED; E.infeasible = 0; E.bail_reason = 0; E.debug_bail = P.k_debug_bail;
#ifdef RSQP_STAMPS
    E.tlast = clock64();
#endif
    int rcode = RET_OK, nWSR = 0;
    double obj = 0.0;
    const bool eligible = d.haveH && d.hreg == 0.0 && d.nV <= ENG::MAXV && d.nC <= ENG::MAXC && 2 * (d.nV + d.nC) <= 4 * ENG::NT;
    if (!eligible) {
        rcode = RET_BAIL; E.bail_reason = 10;
    } else {
        E.stage(P.Ajc + d.offAjc, P.Air + d.offAnz, P.Aval + d.offAnz, P.Hjc + d.offHjc, P.Hir + d.offHnz, P.Hval + d.offHnz,
                P.g + d.offV, P.lb + d.offV, P.ub + d.offV, P.lbA + d.offC, P.ubA + d.offC);
        E.init_M((STATEFUL && mode != 0) ? P.state + d.offState + rsqp_image_bytes(d.nV, d.nC) / 8 : nullptr);
        if (!(E.hscale > 0.0)) { rcode = RET_BAIL; E.bail_reason = 10; }
        else if (STATEFUL && mode != 0 && !E.load_state(P.state + d.offState)) { rcode = RET_BAIL; E.bail_reason = 12; }   // not this kernel's state
        else if (E.bounds_inconsistent()) {
            // (a hot start keeps the stored iterate: what the null-space kernels return in that case)
            E.infeasible = 1; rcode = RET_INFEASIBLE;
        } else {
            // (ONE call site of the homotopy: inlined twice -- hot and cold -- the kernel doubled in code and in AGPR traffic)
            const bool hot = STATEFUL && mode != 0;
            if (!hot) {
                E.status = QPS_PREPARINGAUXILIARYQP;
                rcode = E.setup_cold();
                if (rcode == RET_OK) E.status = QPS_AUXILIARYQPSOLVED;
            }
            if (rcode == RET_OK) rcode = E.homotopy(maxWSR, nWSR, hot);
        }
        if (rcode != RET_BAIL) obj = E.objective();
    }
    const int tid = (int)threadIdx.x;
    if (rcode == RET_BAIL) {
        if (tid == 0) { P.ret[q] = RET_BAIL; P.nflips[q] = E.bail_reason; P.nwsr[q] = 1000 + E.bail_reason; }    // the null-space kernel takes this member over
        //                                  (nwsr: overwritten by it; read by tools/bail_hist.py under RSQP_SMALL_KKT_ONLY=1)
        return;
    }
    for (int v = tid; v < d.nV; v += ENG::NT) { P.x[d.offV + v] = E.x[v]; P.ws_b[d.offV + v] = E.Sb[v]; P.y[d.offV + d.offC + v] = E.yB[v]; }
    for (int i = tid; i < d.nC; i += ENG::NT) { P.y[d.offV + d.offC + d.nV + i] = E.yC[i]; P.ws_c[d.offC + i] = E.Sc[i]; }
    if (tid == 0) {
        const int st = E.status;
        P.status[q] = E.infeasible ? 100 + st : st;
        P.ret[q] = rcode;
        P.nwsr[q] = nWSR;
        P.nflips[q] = 0;
        P.obj[q] = obj;
        if (!STATEFUL || !P.keep_state) {
            // no hot-start state wanted: mark the persistent image "not initialised" (layout of the explicit-inverse engine:
            // persist_doubles doubles, then the integer image Sb | Sc | AC | posAC | iscal, status in iscal[3])
            const long long xnp = EngineX<256, true>::persist_doubles(d.nV, d.nC);
            reinterpret_cast<int *>(P.state + d.offState + xnp)[d.nV + 3 * d.nC + 3] = QPS_NOTINITIALISED;
        }
    }
    if constexpr (STATEFUL) { if (P.keep_state) E.store_state(P.state + d.offState); }
}
