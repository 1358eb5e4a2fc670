// qp_small_g.h -- included by qp_small.hip inside its anonymous namespace (after qp_small_x.h).
//
// small_qpg_kernel: the register-resident engine for mid-size problems in the TABLEAU formulation (round 4; it replaces the
// explicit-KKT-inverse kernel of round 3, qp_small_k.h, whose products and pivots it no longer needs). Same homotopy, ratio
// tests, tie breaks and drift correction as the other engines and the CPU restatement -- the step directions of an
// active-set iteration do not depend on how the KKT system is solved -- but the whole linear algebra is ONE symmetric
// (nV + nC) x (nV + nC) matrix in FIXED slots (slot of variable v = v, of constraint i = nV + i):
//        G = - SWEEP_S(K),   K = [H A'; A 0],   S = free variables + active constraints,
//        G_SS = K_SS^-1,   G_SN = -K_SS^-1 K_SN,   G_NN = -(K_NN - K_NS K_SS^-1 K_SN)      (N = all other slots),
// i.e. the relation [z_S; -r_N] = G [r_S; z_N] for K z = r. What that buys per working-set change:
//   * the step direction of EVERY quantity is ONE product: in = (-dg on free variables | bound moves on fixed variables |
//     limit moves on active constraints | 0 on inactive constraints) gives out = (dx_FR | -(H dx - A'dy) on fixed variables |
//     -dy on active constraints | -A dx on inactive constraints). No second product, no carried A dx_FX / H dx_FX.
//   * a working-set change is ONE principal pivot on the slot q that changes sides: with u = column q of G, pi = G_qq,
//     G <- G0 - (1 / pi) u~ u~', G0 = G with row and column q zeroed, u~ = u except u~_q = +1 (q enters S) / -1 (q leaves S).
//     The column is READ from the registers that hold it: no product u = M k, no dots for the pivot (pi is an entry).
//   * an exchange (the incoming row depends on the working set) is ONE 2 x 2 block pivot on (partner, incoming): defined
//     whenever the exchange is -- also when the partner alone would leave Z'HZ singular (hs071-like QPs: no curvature on
//     the slacks), the case the round-3 kernel left to a second launch. The dependency coefficients of the active
//     constraints AND of the fixed variables are the pivot column itself.
//   * a removal that would leave Z'HZ not positive definite is a flip to the opposite side ("flipping bounds"): G unchanged.
// The price: G is kept current by updates only -- no entry is ever re-derived from H and A -- so rounding accumulates over
// the changes (CPU prototype: worst |dy| 1.4e-9 after 63 changes). ONE step of iterative refinement on the final KKT system,
// with residuals formed from the data, ends every solve (worst error of 2000 random QPs: 5e-14).
// Phases of a working-set change (a barrier after each):
//   A  out = G in; the owner of a row turns its entry into dx / dy / A dx and its ratio-test candidates; block argmin
//   B  homotopy step by the row owners; the lanes that hold row q publish it (= column q: G is symmetric)
//   C  decision (pivot tests, independence from |P a|, exchange, flip), rank-1 / rank-2 update of the register blocks, working
//      set, drift correction and the input vector of the next product by the row owners
//   (+ every 8 changes and after exchanges / flips: A x, A'y - H x from the data; + 2 phases for an exchange)
// MI355X mapping (one problem per workgroup of 256 lanes = 4 waves, lane = 8 bi + bj: 32 row blocks x 8 column blocks): lane
// (bi, bj) holds the (RV + RC) x (CV + CC) block of G for variable rows bi RV + a / constraint rows bi RC + c and variable
// columns bj CV + b / constraint columns bj CC + c, and the same blocks of H, A, A' (exact products) -- all in REGISTERS;
// a product = LDS reads of the input slice, FMAs from registers, a sum over the 8 lanes of a row block by 3 DPP steps.
// What the formulation does not carry ends the kernel for that problem with RET_BAIL and the null-space kernel (EngineX)
// solves it in a second launch: a non-symmetric H, LPs, pivots inside a rounding band, an exchange without a partner (the
// verdict "infeasible" is left to the engine whose A dx is formed from the data), a free variable in the cold working set.
// Validated first as a CPU prototype (tools/proto_k/proto_g.cpp + check.py against the CPU restatement): the 512-QP hs0xx
// batch, 2000 random convex QPs and 14 of the reference's 18 dumps identical, 18 of 3000 degenerate inputs on another path.

#define GSYNC() __syncthreads()

template <int S> __device__ __forceinline__ double gmin_f64(double v) {        // minimum over the wave, every lane gets it
    if constexpr (S == 4) {
        const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
        return fmin(fmin(r0, r1), fmin(r2, r3));
    } else { v = fmin(v, xchg_f64<S>(v)); return gmin_f64<S + 1>(v); }
}
template <int S> __device__ __forceinline__ int gmin_i32(int v) {
    if constexpr (S == 4) {
        const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16), r2 = __builtin_amdgcn_readlane(v, 32),
                  r3 = __builtin_amdgcn_readlane(v, 48);
        return min(min(r0, r1), min(r2, r3));
    } else { v = min(v, xchg_i32<S>(v)); return gmin_i32<S + 1>(v); }
}

// a value the compiler must not look through: a one-hot weight (a == k ? 1.0 : 0.0) that multiplies register-array entries is
// otherwise recognised as a select and turned into an INDEXED load from a copy of the array in scratch memory
__device__ __forceinline__ double opaque(double v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ int opaque_i(int v) { asm volatile("" : "+v"(v)); return v; }

template <int RV, int RC, int CV, int CC>
struct EngineG {
    static constexpr int GI = 32, GJ = 8, NT = GI * GJ, LGJ = 3, NW = NT / 64;
    static constexpr int NVP = GI * RV > GJ * CV ? GI * RV : GJ * CV, NCP = GI * RC > GJ * CC ? GI * RC : GJ * CC;
    static constexpr int MAXV = GI * RV < GJ * CV ? GI * RV : GJ * CV, MAXC = GI * RC < GJ * CC ? GI * RC : GJ * CC;   // largest nV / nC
    static_assert(RV + RC <= GJ, "row owners of a row block must fit its 8 lanes");
    static_assert(NVP + NCP <= NT, "one lane per slot (exchange candidates)");
    static constexpr int NVEC_V = 13, NVEC_C = 9;
    // LDS: vectors of NVP / NCP doubles, integer working sets, reduction slots, then ZERO-PADDED dense copies of A (NCP x NVP,
    // column major) and H (NVP x NVP): the exact products (every 8th change, the end of a solve) read their blocks from there
    // with compile-time trip counts and no bounds tests -- held in registers as well (round 3) they cost 96 VGPRs that the
    // pivots need (the first build of this kernel spilled 150 registers to scratch)
    static constexpr int LDA = NCP, LDH = NVP;
    static constexpr int LDS_BYTES = 8 * (NVEC_V * NVP + NVEC_C * NCP + 16) + 4 * (NVP + NCP + 16) + 8 * (LDA * NVP + LDH * NVP);
    // ---- LDS (lb | ub, lbN | ubN, lbA | ubA, lbAN | ubAN are ADJACENT arrays: a side picks an offset, never a pointer)
    ldouble *x, *g, *lb, *ub, *gN, *lbN, *ubN, *yB, *gy, *inV, *uV, *u2V, *tV;
    ldouble *Ax, *lbA, *ubA, *lbAN, *ubAN, *yC, *inC, *uC, *u2C;
    ldouble *red;
    LDS int *Sb, *Sc, *ired;
    ldouble *Ad, *Hd;
    // ---- registers
    static constexpr int R = RV + RC, CN = CV + CC;
    double G[R][CN];           // rows: the RV variable rows, then the RC constraint rows of bi; columns: CV variable, then CC constraint columns of bj
    // (ONE array: with four blocks the compiler fused the structurally equal branches "row q is a variable row" / "a constraint
    //  row" of the <2,2,8,8> build into one body that SELECTS the block's address -- which put all of G into scratch memory)
    int nV, nC, tid, bi, bj, wave;
    int nFR, nAC, status, infeasible, unbounded, nflips, bail_reason, parity;
    int debug_bail;             // >= 0: a hot start bails out before its debug_bail-th change (tests of the hand-over); else -1
    int since_refresh;          // working-set changes since A x and A'y - H x were last formed from the data; >= REFRESH: do it now
    static constexpr int REFRESH = 8;
    double hscale;
    double wV[RV], wC[RC];     // one-hot: wV[a] = (bj == a), wC[c] = (bj - RV == c)
    long long tlast;   // (-DRSQP_STAMPS builds: cycles per phase of block 0, tools/stamp_k_kernel.py)

    __device__ __forceinline__ void carve(lchar *base, int nV_, int nC_) {
        nV = nV_; nC = nC_;
        tid = (int)threadIdx.x; bi = tid >> LGJ; bj = tid & (GJ - 1); wave = tid >> 6;
        ldouble *p = (ldouble *)base;
#define GV_(name) name = p; p += NVP
#define GC_(name) name = p; p += NCP
        // (slot vectors -- variable part, then constraint part ADJACENT: a lane that serves "slot s" indexes ONE array with s or
        //  NVP + i; a select between two LDS arrays becomes a table of their addresses in scratch memory)
        GV_(x); GV_(g); GV_(lb); GV_(ub); GV_(gN); GV_(lbN); GV_(ubN); GV_(gy); GV_(tV);
        GV_(yB); GC_(yC); GV_(inV); GC_(inC); GV_(uV); GC_(uC); GV_(u2V); GC_(u2C);
        GC_(Ax); GC_(lbA); GC_(ubA); GC_(lbAN); GC_(ubAN);
#undef GV_
#undef GC_
        red = p; p += 16;
        LDS int *ip = (LDS int *)p;
        Sb = ip; ip += NVP; Sc = ip; ip += NCP; ired = ip; ip += 16;
        Ad = (ldouble *)ip;
        Hd = Ad + LDA * NVP;
        parity = 0;
#pragma unroll
        for (int a = 0; a < RV; a++) wV[a] = opaque(bj == a ? 1.0 : 0.0);
#pragma unroll
        for (int c = 0; c < RC; c++) wC[c] = opaque(bj - RV == c ? 1.0 : 0.0);
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
    // acc[k] for the lane-dependent k of a row owner, as the product with the lane's one-hot weights: exact, and -- unlike a
    // chain of selects, which the compiler turns back into an indexed load from a copy of acc in SCRATCH memory -- in registers
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
    __device__ __forceinline__ static double clampinf(double v) { return v > RSQP_INFTY ? RSQP_INFTY : (v < -RSQP_INFTY ? -RSQP_INFTY : v); }
    // sums over the workgroup's vectors, formed by EVERY wave from the published LDS operands: all lanes agree, no barrier
    __device__ __forceinline__ double wdotV(const ldouble *a, const ldouble *b) const {
        const int l = tid & 63;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < (NVP + 63) / 64; k++) { const int i = l + 64 * k; if (i < NVP) s = fma(a[i], b[i], s); }
        return allreduce_sum<6>(s);
    }
    // |u_FR|^2 and |a_FR|^2 of the pivot column / the incoming row over the FREE variables (a: row `arow` of A, or e_arow when
    // arow < 0 encodes the unit vector of variable -arow - 1)
    __device__ __forceinline__ void free_norms(int arow, double &pn2, double &na2) const {
        const int l = tid & 63;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < (NVP + 63) / 64; k++) {
            const int v = l + 64 * k;
            if (v < nV && Sb[v] == 0) {
                const double u = uV[v];
                const double a = arow >= 0 ? Ad[arow + v * LDA] : (v == -arow - 1 ? 1.0 : 0.0);
                s1 = fma(u, u, s1); s2 = fma(a, a, s2);
            }
        }
        pn2 = allreduce_sum<6>(s1); na2 = allreduce_sum<6>(s2);
    }
    // lexicographic minimum of (t, id) over the workgroup, ONE barrier (the exchange slots alternate)
    __device__ __forceinline__ void block_argmin(double &t, int &id) {
        const double tm = gmin_f64<0>(t);
        const int im = gmin_i32<0>(t == tm ? id : 0x7fffffff);
        parity ^= 8;
        if ((tid & 63) == 0) { red[parity + wave] = tm; ired[parity + wave] = im; }
        GSYNC();
        t = red[parity]; id = ired[parity];
#pragma unroll
        for (int w = 1; w < NW; w++) {
            const double t2 = red[parity + w]; const int id2 = ired[parity + w];
            if (t2 < t || (t2 == t && id2 < id)) { t = t2; id = id2; }
        }
    }
    __device__ __forceinline__ static void cand(double num, double den, int id, bool ok, double &bt, int &bid) {
        if (ok && den >= RSQP_EPS_DEN) {
            const double t = (num > 0.0 ? num : 0.0) / den;
            if (t < bt || (t == bt && id < bid)) { bt = t; bid = id; }
        }
    }

    // ------------------------------------------------------------------ staging
    __device__ __forceinline__ void stage(const int *gAjc, const int *gAir, const double *gAval, const int *gHjc, const int *gHir,
                                          const double *gHval, const double *g_, const double *lb_, const double *ub_,
                                          const double *lbA_, const double *ubA_) {
        for (int k = tid; k < LDA * NVP + LDH * NVP; k += NT) Ad[k] = 0.0;
        for (int v = tid; v < NVP; v += NT) {
            const bool in = v < nV;
            x[v] = 0.0; g[v] = 0.0; yB[v] = 0.0; gy[v] = 0.0; inV[v] = 0.0; uV[v] = 0.0; u2V[v] = 0.0; tV[v] = 0.0;
            gN[v] = in ? g_[v] : 0.0; lbN[v] = in ? clampinf(lb_[v]) : 0.0; ubN[v] = in ? clampinf(ub_[v]) : 0.0;
            lb[v] = 0.0; ub[v] = 0.0; Sb[v] = -1;
        }
        for (int i = tid; i < NCP; i += NT) {
            const bool in = i < nC;
            Ax[i] = 0.0; yC[i] = 0.0; inC[i] = 0.0; uC[i] = 0.0; u2C[i] = 0.0;
            lbAN[i] = in ? clampinf(lbA_[i]) : -RSQP_INFTY; ubAN[i] = in ? clampinf(ubA_[i]) : RSQP_INFTY;
            lbA[i] = -RSQP_INFTY; ubA[i] = RSQP_INFTY; Sc[i] = 0;
        }
        GSYNC();
        for (int c = tid; c < nV; c += NT) {
            for (int k = gAjc[c]; k < gAjc[c + 1]; k++) Ad[gAir[k] + c * LDA] = gAval[k];
            for (int k = gHjc[c]; k < gHjc[c + 1]; k++) Hd[gHir[k] + c * LDH] = gHval[k];
        }
        GSYNC();
        double hm = 0.0;
        for (int v = tid & 63; v < nV; v += 64) hm = fmax(hm, fabs(Hd[v + v * LDH]));
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) hm = fmax(hm, __shfl_xor(hm, s));
        // H has to be symmetric (G is kept symmetric by construction); some of the reference's own inputs
        // (test/unsolved_QPs/*.hpp) are not -- those go to the null-space kernel. Every lane checks its own block.
        int asym = 0;
#pragma unroll
        for (int a = 0; a < RV; a++)
#pragma unroll
            for (int b = 0; b < CV; b++) {
                const int r = bi * RV + a, c = bj * CV + b;
                if (Hd[r + c * LDH] != Hd[c + r * LDH]) asym = 1;
            }
        const int wany = __any(asym) ? 1 : 0;
        if ((tid & 63) == 0) ired[wave] = wany;
        GSYNC();
        int any = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) any |= ired[w];
        GSYNC();
        hscale = any ? 0.0 : hm;      // (hscale = 0 makes the kernel bail)
    }
    __device__ __forceinline__ bool bounds_inconsistent() const {
        double bad = 0.0;
        for (int v = tid & 63; v < nV; v += 64) if (lbN[v] > ubN[v] + RSQP_EPS) bad += 1.0;
        for (int i = tid & 63; i < nC; i += 64) if (lbAN[i] > ubAN[i] + RSQP_EPS) bad += 1.0;
        return allreduce_sum<6>(bad) > 0.0;
    }
    // G: -K (cold start: S is empty) or the stored tableau (hot start; pm = the extension behind the null-space image). ONE
    // definition site for the register blocks (a second one doubled the registers the allocator kept busy)
    __device__ __forceinline__ void init_G(const double *pm) {
        const int N = nV + nC;
        const bool ld_ = pm != nullptr;
#pragma unroll
        for (int a = 0; a < R; a++) {
            const bool rv = a < RV;
            const int r = rv ? bi * RV + a : bi * RC + (a - RV);           // variable / constraint number of the row
            const bool rin = rv ? r < nV : r < nC;
            const int rs = rv ? r : nV + r;
#pragma unroll
            for (int b = 0; b < CN; b++) {
                const bool cv = b < CV;
                const int c = cv ? bj * CV + b : bj * CC + (b - CV);
                const bool cin = cv ? c < nV : c < nC;
                const int cs = cv ? c : nV + c;
                // -K: -H | -A' | -A | 0 from the zero-padded copies
                const double k0 = rv ? (cv ? -Hd[r + c * LDH] : -Ad[c + r * LDA]) : (cv ? -Ad[r + c * LDA] : 0.0);
                G[a][b] = (ld_ && rin && cin) ? pm[rs + cs * N] : k0;
            }
        }
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
        }
        for (int i = tid; i < nC; i += NT) { lbA[i] = fmin(lbAN[i], -RSQP_BOUND_RELAXATION); ubA[i] = fmax(ubAN[i], RSQP_BOUND_RELAXATION); }
        nFR = nAC = 0;
        GSYNC();
        return RET_OK;
    }

    // ------------------------------------------------------------------ the row owners' element-wise work
    // drift correction (gradient of the current QP from stationarity, active limits := A x) and the input of the next product
    // keep_data (first pass of a hot start): g, A x and the limits of the active constraints stay what the previous solve left
    __device__ __forceinline__ void make_input(bool keep_data) {
        if (ownsV()) {
            const int v = rowV(), s = Sb[v];
            const double gv = keep_data ? g[v] : gy[v] + yB[v];
            if (keep_data) gy[v] = gv - yB[v];
            g[v] = gv;
            const int so = v + (s == 1 ? NVP : 0);
            inV[v] = v < nV ? (s == 0 ? -(gN[v] - gv) : lbN[so] - lb[so]) : 0.0;
        } else if (ownsC()) {
            const int i = rowC(), s = Sc[i];
            const int so = i + (s == 1 ? NCP : 0);
            if (s != 0 && !keep_data) lbA[so] = Ax[i];
            inC[i] = s != 0 ? lbAN[so] - lbA[so] : 0.0;
        }
    }
    // A x and A'y_C - H x of the iterate from the data (every REFRESH changes, after an exchange or a flip, at the start)
    __device__ __forceinline__ void refresh_exact(bool keep_data) {
        if (!keep_data) {
            double ax[RC], hx[RV], aty[RV];
            exact_products(ax, hx, aty);
            if (ownsV()) gy[rowV()] = pick<RV>(aty, wV) - pick<RV>(hx, wV);
            else if (ownsC()) Ax[rowC()] = pick<RC>(ax, wC);
        }
        since_refresh = 0;
    }
    // A x, H x, A'y_C of the iterate from the DATA (the zero-padded dense copies in LDS), summed over the row block
    __device__ __forceinline__ void exact_products(double (&ax)[RC], double (&hx)[RV], double (&aty)[RV]) const {
        double xv[CV], yc[CC];
        ldcols<CV>(x, xv); ldcols<CC>(yC, yc);
        zero<RC>(ax); zero<RV>(hx); zero<RV>(aty);
#pragma unroll
        for (int b = 0; b < CV; b++) {
            const int c = bj * CV + b;
#pragma unroll
            for (int a = 0; a < RC; a++) ax[a] = fma(Ad[bi * RC + a + c * LDA], xv[b], ax[a]);
#pragma unroll
            for (int a = 0; a < RV; a++) hx[a] = fma(Hd[bi * RV + a + c * LDH], xv[b], hx[a]);
        }
#pragma unroll
        for (int b = 0; b < CC; b++) {
            const int i = bj * CC + b;
#pragma unroll
            for (int a = 0; a < RV; a++) aty[a] = fma(Ad[i + (bi * RV + a) * LDA], yc[b], aty[a]);
        }
        rowsum<RC>(ax); rowsum<RV>(hx); rowsum<RV>(aty);
    }

    // ------------------------------------------------------------------ pivots
    // rows / columns of this lane: slot numbers (padding rows / columns get -1: they never match a pivot), published entries
    __device__ __forceinline__ int rslot(int a) const {
        if (a < RV) { const int r = bi * RV + a; return r < nV ? r : -1; }
        const int r = bi * RC + (a - RV); return r < nC ? nV + r : -1;
    }
    __device__ __forceinline__ int cslot(int b) const {
        if (b < CV) { const int c = bj * CV + b; return c < nV ? c : -1; }
        const int c = bj * CC + (b - CV); return c < nC ? nV + c : -1;
    }
    __device__ __forceinline__ static double rowof(const ldouble *oV, const ldouble *oC, int bi_, int a) { return a < RV ? oV[bi_ * RV + a] : oC[bi_ * RC + (a - RV)]; }
    __device__ __forceinline__ static double colof(const ldouble *oV, const ldouble *oC, int bj_, int b) { return b < CV ? oV[bj_ * CV + b] : oC[bj_ * CC + (b - CV)]; }
    // the lanes that hold row q of G publish it (= column q): slot q = variable q (< nV) or constraint q - nV
    __device__ __forceinline__ void publish_row(int q, ldouble *oV, ldouble *oC) {
        const bool qv = q < nV;
        const int qq = qv ? q : q - nV, rb = qv ? qq / RV : qq / RC;
        if (bi == rb) {
            const int aq = qv ? qq - rb * RV : RV + (qq - rb * RC);
            double w[R];
#pragma unroll
            for (int a = 0; a < R; a++) w[a] = opaque(a == aq ? 1.0 : 0.0);
#pragma unroll
            for (int b = 0; b < CN; b++) {
                double v = G[0][b] * w[0];
#pragma unroll
                for (int a = 1; a < R; a++) v = fma(G[a][b], w[a], v);
                if (b < CV) oV[bj * CV + b] = v; else oC[bj * CC + (b - CV)] = v;
            }
        }
    }

    // principal pivot on slot q with the published column (uV | uC), pi = its entry q, sgn = +1 (q enters S) / -1 (leaves)
    __device__ __forceinline__ void pivot1(int q, double sgn, double pi) {
        const double c = -1.0 / pi;
        double tr[R], uc[CN], ck[CN];
        bool rowhit = false;
#pragma unroll
        for (int a = 0; a < R; a++) { const bool h = rslot(a) == q; rowhit |= h; tr[a] = c * (h ? sgn : rowof(uV, uC, bi, a)); }
#pragma unroll
        for (int b = 0; b < CN; b++) { const bool h = cslot(b) == q; uc[b] = h ? sgn : colof(uV, uC, bj, b); ck[b] = h ? 0.0 : 1.0; }
        if (rowhit) {      // the 8 lanes of ONE row block: the old row q goes (the other waves skip this)
#pragma unroll
            for (int a = 0; a < R; a++) {
                const double k = rslot(a) == q ? 0.0 : 1.0;
#pragma unroll
                for (int b = 0; b < CN; b++) G[a][b] *= k;
            }
        }
#pragma unroll
        for (int a = 0; a < R; a++)
#pragma unroll
            for (int b = 0; b < CN; b++) G[a][b] = fma(tr[a], uc[b], G[a][b] * ck[b]);
    }
    // 2 x 2 block pivot on (p, q): columns (u2V | u2C) of p and (uV | uC) of q published; W = P^-1 = [w11 w12; w12 w22] of
    // P = [G_pp G_pq; G_pq G_qq]; sp / sq = +1 (enters S) / -1 (leaves S):  G <- G00 - U~ W U~',  U~ = [u_p u_q] with rows p, q = diag(sp, sq)
    __device__ __forceinline__ void pivot2(int p, double sp, int q, double sq, double w11, double w12, double w22) {
        double rp[R], rq[R], rk[R];
        bool rowhit = false;
#pragma unroll
        for (int a = 0; a < R; a++) {
            const int s = rslot(a); const bool hp = s == p, hq = s == q; rowhit |= hp | hq;
            rp[a] = hp ? sp : (hq ? 0.0 : rowof(u2V, u2C, bi, a)); rq[a] = hq ? sq : (hp ? 0.0 : rowof(uV, uC, bi, a));
            rk[a] = (hp || hq) ? 0.0 : 1.0;
        }
        if (rowhit) {
#pragma unroll
            for (int a = 0; a < R; a++)
#pragma unroll
                for (int b = 0; b < CN; b++) G[a][b] *= rk[a];
        }
        // column by column (the per-column factors live for one column only: the rare path must not set the kernel's register peak)
#pragma unroll
        for (int b = 0; b < CN; b++) {
            const int s = cslot(b); const bool hp = s == p, hq = s == q;
            const double up = hp ? sp : (hq ? 0.0 : colof(u2V, u2C, bj, b)), uq = hq ? sq : (hp ? 0.0 : colof(uV, uC, bj, b));
            const double cp = fma(w11, up, w12 * uq), cq = fma(w12, up, w22 * uq), ck = (hp || hq) ? 0.0 : 1.0;
#pragma unroll
            for (int a = 0; a < R; a++) G[a][b] = fma(-rp[a], cp, fma(-rq[a], cq, G[a][b] * ck));
        }
    }

    // out = G in for this lane's rows (summed over the row block); the caller picks its own entry
    __device__ __forceinline__ void g_times(double (&sv)[RV], double (&sc)[RC]) const {
        double xin[CN], acc[R];
#pragma unroll
        for (int b = 0; b < CN; b++) xin[b] = colof(inV, inC, bj, b);
        zero<R>(acc);
        mv<R, CN>(G, xin, acc);
        rowsum<R>(acc);
#pragma unroll
        for (int a = 0; a < RV; a++) sv[a] = acc[a];
#pragma unroll
        for (int a = 0; a < RC; a++) sc[a] = acc[RV + a];
    }

    // ------------------------------------------------------------------ phase C: one working-set change
    // kind 1 constraint idx leaves | 2 bound of idx leaves | 3 constraint idx enters at `side` | 4 variable idx gets fixed at `side`;
    // the column of the slot is published in (uV | uC). Ends with the working set, x on its active bounds and the input of the
    // next product written by the row owners (the caller closes with the barrier) -- unless exact products are due.
    __device__ __forceinline__ int change(int kind, int idx, int side) {
        const int q = (kind == 1 || kind == 3) ? nV + idx : idx;
        const double pi = uV[q < nV ? q : NVP + (q - nV)];
        int pk = 0, pidx = -1;                // exchange partner: 1 constraint / 2 bound
        double ynew = 0.0;
        bool flip = false;
        if (kind == 1) {
            double d2, dummy;
            free_norms(-(nV + 1) - 1, d2, dummy);     // (no incoming row: a = 0)
            const bool ok = d2 > 0.0 && -pi > 1e-8 * hscale * d2;
            if (!ok) {
                if (d2 > 0.0 && !(-pi < 1e-11 * hscale * d2)) { bail_reason = 1; return RET_BAIL; }
                flip = true;
            }
        } else if (kind == 2) {
            const double sigma = -pi;
            if (!(sigma > 1e-8 * hscale)) {
                if (!(sigma < 1e-11 * hscale)) { bail_reason = 2; return RET_BAIL; }
                flip = true;
            }
        }
        STAMP(35);
        if (flip) {
            // the released direction has no curvature: the constraint / bound goes to its OPPOSITE side, G is unchanged
            // (two branches that must stay two: merged by the compiler they SELECT the array addresses -- a table in scratch memory;
            //  the opaque index keeps their code different)
            int old;
            double farside;
            if (kind == 1) { old = Sc[idx]; farside = lbAN[idx + (old == -1 ? NCP : 0)]; }
            else { const int iv = opaque_i(idx); old = Sb[iv]; farside = lbN[iv + (old == -1 ? NVP : 0)]; }
            if (fabs(farside) >= RSQP_INFTY) return RET_UNBOUNDED;
            GSYNC();                            // (every lane has read the old side)
            if (kind == 1) { if (ownsC() && rowC() == idx) { Sc[idx] = -old; lbA[idx + (old == -1 ? NCP : 0)] = Ax[idx]; yC[idx] = 0.0; } }
            else if (ownsV() && rowV() == idx) { const int iv = opaque_i(idx); Sb[iv] = -old; lb[iv + (old == -1 ? NVP : 0)] = x[iv]; yB[iv] = 0.0; }
            nflips++;
            since_refresh = REFRESH;
            return RET_OK;
        }
        if (kind >= 3) {
            // independence of the incoming row a from the working set. |P a| -- the free-variable part of the pivot column -- is of
            // first order in |Z'a| (|Z'a| / lmax <= |P a| <= |Z'a| / lmin); in the band between "clearly independent" and
            // "clearly dependent" the residual r = a_FR - A_AC,FR' xi_C of the row's representation by the active rows decides
            // (|Z'a| <= |r| <= cond |Z'a|, the test of qp_small_k.h)
            const double sg = kind == 3 ? -1.0 : 1.0;      // xi = -column for a constraint that enters S, +column for a variable that leaves it
            int li;
            double pn2, na2;
            free_norms(kind == 3 ? idx : -idx - 1, pn2, na2);
            if (nFR - nAC <= 0 || !(na2 > 0.0)) li = 0;
            else {
                const double rel = hscale * sqrt(pn2 / na2);
                li = rel > 1e-6 ? 1 : (rel < 1e-12 ? 0 : -1);
            }
            if (li < 0) {
                double at[RV];
                zero<RV>(at);
#pragma unroll
                for (int c_ = 0; c_ < CC; c_++) {
                    const int i = bj * CC + c_;
                    const double yc = Sc[i] != 0 ? sg * uC[i] : 0.0;
#pragma unroll
                    for (int a = 0; a < RV; a++) at[a] = fma(Ad[i + (bi * RV + a) * LDA], yc, at[a]);
                }
                rowsum<RV>(at);
                if (ownsV()) {
                    const int v = rowV();
                    const double a = v < nV ? (kind == 3 ? Ad[idx + v * LDA] : (v == idx ? 1.0 : 0.0)) : 0.0;
                    tV[v] = (v < nV && Sb[v] == 0) ? a - pick<RV>(at, wV) : 0.0;
                }
                GSYNC();
                const double rel = sqrt(wdotV(tV, tV) / na2);
                li = rel > 1e-7 ? 1 : (rel < 1e-9 ? 0 : -1);
                if (li < 0) { bail_reason = 3; return RET_BAIL; }
            }
            STAMP(36);
            if (li == 0) {
                // ---- exchange: shift the multipliers along the dependency until one of them reaches zero; that one leaves
                const double sgn = side == 1 ? -1.0 : 1.0;
                double bt = RSQP_INFTY;
                int bid = 0x7fffffff;
                if (tid < nC + nV) {
                    const int sl = tid < nC ? NVP + tid : tid - nC;      // (Sb | Sc, uV | uC, yB | yC are adjacent: one array, slot index)
                    const int s = Sb[sl];
                    if (s != 0) {
                        const double xi = sgn * sg * uV[sl], yi = yB[sl];
                        const double num = s == -1 ? yi : -yi, den = s == -1 ? xi : -xi;
                        if (den > RSQP_EPS_DEN) { bt = (num > 0.0 ? num : 0.0) / den; bid = tid; }
                    }
                }
                block_argmin(bt, bid);
                if (bid == 0x7fffffff) { bail_reason = 8; return RET_BAIL; }      // (no partner: the null-space engine decides "infeasible")
                if (bid < nC) { pk = 1; pidx = bid; } else { pk = 2; pidx = bid - nC; }
                if (ownsV()) { const int v = rowV(); if (v < nV && Sb[v] != 0) yB[v] -= bt * sgn * sg * uV[v]; }
                else if (ownsC()) { const int i = rowC(); if (i < nC && Sc[i] != 0) yC[i] -= bt * sgn * sg * uC[i]; }
                ynew = sgn * bt;
                const int p = pk == 1 ? nV + pidx : pidx;
                publish_row(p, u2V, u2C);
                GSYNC();
                const int psl = p < nV ? p : NVP + (p - nV);
                const double pp = u2V[psl], qq = pi, pq = uV[psl];
                const double det = pp * qq - pq * pq;
                if (!(det < 0.0) || !(-det > 1e-10 * fmax(fabs(pp * qq), pq * pq))) { bail_reason = 5; return RET_BAIL; }
                const double w11 = qq / det, w12 = -pq / det, w22 = pp / det;
                pivot2(p, pk == 1 ? -1.0 : 1.0, q, kind == 3 ? 1.0 : -1.0, w11, w12, w22);
                since_refresh = REFRESH;
            } else {
                if (kind == 3) { if (!(pi > 1e-10 * na2 / hscale)) { bail_reason = 6; return RET_BAIL; } }
                else if (!(pi > 1e-10 / hscale)) { bail_reason = 7; return RET_BAIL; }
            }
        }
        // (ONE call site of the single pivot: a slot ENTERS S when a bound leaves the working set or a constraint joins it)
        if (pk == 0) pivot1(q, (kind == 2 || kind == 3) ? 1.0 : -1.0, pi);
        STAMP(37);
        // ---- the working set (the owners of the changed rows)
        if (ownsV()) {
            const int v = rowV();
            if (kind == 2 && v == idx) { Sb[v] = 0; yB[v] = 0.0; }
            if (kind == 4 && v == idx) { Sb[v] = side; yB[v] = ynew; }
            if (pk == 2 && v == pidx) { Sb[v] = 0; yB[v] = 0.0; }
        } else if (ownsC()) {
            const int i = rowC();
            if (kind == 1 && i == idx) { Sc[i] = 0; yC[i] = 0.0; }
            if (kind == 3 && i == idx) { Sc[i] = side; yC[i] = ynew; }
            if (pk == 1 && i == pidx) { Sc[i] = 0; yC[i] = 0.0; }
        }
        if (kind == 1) nAC--; else if (kind == 2) nFR++; else if (kind == 3) nAC++; else nFR--;
        if (pk == 1) nAC--; else if (pk == 2) nFR++;
        return RET_OK;
    }

    __device__ __forceinline__ int homotopy(int maxit, int &nWSR, bool hot) {
        int iter = 0, rcode = RET_OK;
        status = QPS_PERFORMINGHOMOTOPY;
        since_refresh = REFRESH;
        for (;;) {
            // ---- x exactly on its active bounds; (exact products); drift correction + input of the product
            if (ownsV()) { const int v = rowV(), s = Sb[v]; if (s != 0 && v < nV) x[v] = s == -1 ? lb[v] : ub[v]; }
            const bool keep = hot && iter == 0;
            if (since_refresh >= REFRESH) {
                GSYNC();
                refresh_exact(keep);
                STAMP(30);
            }
            make_input(keep);
            GSYNC();
            STAMP(38);
            // ---- phase A: out = G in; the row owners form dx / dy / A dx and their ratio-test candidates
            double bt = 1.0, dxv = 0.0, dyv = 0.0, hdv_ = 0.0;
            int bid = 0x7fffffff;
            {
                double sv[RV], sc[RC];
                g_times(sv, sc);
                STAMP(31);
                if (ownsV()) {
                    const int v = rowV();
                    if (v < nV) {
                        const double out = pick<RV>(sv, wV);
                        const int s = Sb[v];
                        const double dg = gN[v] - g[v], xx = x[v], yi = yB[v];
                        if (s == 0) { dxv = out; dyv = 0.0; hdv_ = -dg; }
                        else { dxv = inV[v]; dyv = dg - out; hdv_ = -out; }
                        const double l = lb[v], u = ub[v], lN = lbN[v], uN = ubN[v];
                        // active: the multiplier reaches zero | inactive: the lower bound is hit
                        cand(s != 0 ? (s == -1 ? yi : -yi) : xx - l, s != 0 ? (s == -1 ? -dyv : dyv) : (lN - l) - dxv,
                             s != 0 ? nC + v : 3 * nC + nV + v, s != 0 || lN > -RSQP_INFTY, bt, bid);
                        cand(u - xx, dxv - (uN - u), 3 * nC + 2 * nV + v, s == 0 && uN < RSQP_INFTY, bt, bid);
                    }
                } else if (ownsC()) {
                    const int i = rowC();
                    if (i < nC) {
                        const double out = pick<RC>(sc, wC);
                        const int s = Sc[i];
                        const double ax = Ax[i], yi = yC[i];
                        if (s != 0) { dyv = -out; dxv = inC[i]; }      // (dxv holds A dx of the row, dyv its dy)
                        else { dyv = 0.0; dxv = -out; }
                        const double l = lbA[i], u = ubA[i], lN = lbAN[i], uN = ubAN[i];
                        cand(s != 0 ? (s == -1 ? yi : -yi) : ax - l, s != 0 ? (s == -1 ? -dyv : dyv) : (lN - l) - dxv,
                             s != 0 ? i : nC + nV + i, s != 0 || lN > -RSQP_INFTY, bt, bid);
                        cand(u - ax, dxv - (uN - u), 2 * nC + nV + i, s == 0 && uN < RSQP_INFTY, bt, bid);
                    }
                }
            }
            if (!(bt < 1.0)) { bt = 1.0; bid = 0x7fffffff; }
            STAMP(32);
            block_argmin(bt, bid);
            STAMP(33);
            // ---- phase B: decode, homotopy step by the row owners, the column of the change
            int kind = 0, idx = -1, side = 0;
            if (bid != 0x7fffffff) {
                if (bid < nC) { kind = 1; idx = bid; }
                else if (bid < nC + nV) { kind = 2; idx = bid - nC; }
                else if (bid < 2 * nC + nV) { kind = 3; idx = bid - nC - nV; side = -1; }
                else if (bid < 3 * nC + nV) { kind = 3; idx = bid - 2 * nC - nV; side = 1; }
                else if (bid < 3 * nC + 2 * nV) { kind = 4; idx = bid - 3 * nC - nV; side = -1; }
                else { kind = 4; idx = bid - 3 * nC - 2 * nV; side = 1; }
            }
            const double tau = bt;
            const bool done = kind == 0, cap = iter >= maxit;
            if (ownsV()) {
                const int v = rowV();
                if (v < nV) {
                    const int s = Sb[v];
                    if (done) {
                        g[v] = gN[v]; lb[v] = lbN[v]; ub[v] = ubN[v];
                        x[v] = s == -1 ? lbN[v] : (s == 1 ? ubN[v] : x[v] + tau * dxv);
                    } else {
                        const double xn = x[v] + tau * dxv;
                        x[v] = xn;
                        g[v] += tau * (gN[v] - g[v]);
                        gy[v] -= tau * hdv_;                          // A'y_C - H x follows the step
                        const double l = lb[v] + tau * (lbN[v] - lb[v]), u = ub[v] + tau * (ubN[v] - ub[v]);
                        lb[v] = (!cap && kind == 4 && side == -1 && v == idx) ? xn : l;      // the blocking quantity sits exactly on its limit
                        ub[v] = (!cap && kind == 4 && side == 1 && v == idx) ? xn : u;
                    }
                    yB[v] += tau * dyv;
                }
            } else if (ownsC()) {
                const int i = rowC();
                if (i < nC) {
                    yC[i] += tau * dyv;
                    if (done) { lbA[i] = lbAN[i]; ubA[i] = ubAN[i]; }
                    else {
                        const double an = Ax[i] + tau * dxv;          // A x follows the step
                        Ax[i] = an;
                        const double l = lbA[i] + tau * (lbAN[i] - lbA[i]), u = ubA[i] + tau * (ubAN[i] - ubA[i]);
                        lbA[i] = (!cap && kind == 3 && side == -1 && i == idx) ? an : l;
                        ubA[i] = (!cap && kind == 3 && side == 1 && i == idx) ? an : u;
                    }
                }
            }
            if (done || cap) {
                if (done) status = QPS_SOLVED; else rcode = RET_MAX_NWSR;
                break;
            }
            if (hot && iter == debug_bail) { bail_reason = 13; rcode = RET_BAIL; break; }      // (test hook, see P.k_debug_bail)
            publish_row((kind == 1 || kind == 3) ? nV + idx : idx, uV, uC);
            GSYNC();
            STAMP(34);
            // ---- phase C
            rcode = change(kind, idx, side);
            if (rcode == RET_UNBOUNDED) { unbounded = 1; break; }
            if (rcode != RET_OK) break;
            iter++;
            since_refresh++;
        }
        nWSR = iter;
        return rcode;
    }

    // ------------------------------------------------------------------ the end of a solve
    // solved: ONE step of iterative refinement on the final KKT system with residuals from the data (header), the multipliers
    // of the fixed variables from stationarity, A x of the final iterate; returns the objective 0.5 x'Hx + gN'x.
    // refine = false (iteration limit): only A x and the objective
    __device__ __forceinline__ double finish(bool refine) {
        GSYNC();
        // pass 0 (refine only): residuals from the data -> correction out = G in -> x_FR, y_AC; pass 1: A x, H x, A'y_C of the final
        // iterate -> multipliers of the fixed variables, stored A x, H x for the objective (ONE call site of the exact products)
        for (int pass = refine ? 0 : 1; pass < 2; pass++) {
            double ax[RC], hx[RV], aty[RV];
            exact_products(ax, hx, aty);
            if (pass == 0) {
                if (ownsV()) {
                    const int v = rowV();
                    inV[v] = (v < nV && Sb[v] == 0) ? -(gN[v] + pick<RV>(hx, wV) - pick<RV>(aty, wV)) : 0.0;
                } else if (ownsC()) {
                    const int i = rowC(), s = i < nC ? Sc[i] : 0;
                    inC[i] = s != 0 ? lbAN[i + (s == 1 ? NCP : 0)] - pick<RC>(ax, wC) : 0.0;
                }
                GSYNC();
                double sv[RV], sc[RC];
                g_times(sv, sc);
                if (ownsV()) { const int v = rowV(); if (v < nV && Sb[v] == 0) x[v] += pick<RV>(sv, wV); }
                else if (ownsC()) { const int i = rowC(); if (i < nC && Sc[i] != 0) yC[i] -= pick<RC>(sc, wC); }
                GSYNC();
            } else {
                if (ownsV()) {
                    const int v = rowV();
                    const double h = pick<RV>(hx, wV);
                    tV[v] = h;
                    if (refine && v < nV) yB[v] = Sb[v] != 0 ? gN[v] + h - pick<RV>(aty, wV) : 0.0;
                } else if (ownsC()) Ax[rowC()] = pick<RC>(ax, wC);
            }
        }
        GSYNC();
        return 0.5 * wdotV(x, tV) + wdotV(gN, x);
    }

    // ------------------------------------------------------------------ persistent state (hot starts)
    // The state lives in the layout of the explicit-inverse engine (qp_small_x.h carve: factors | x g lb ub | A x lbA ubA | y,
    // then Sb Sc AC posAC iscal as ints), so that engine's hot-start modes work on it unchanged (new matrices / warm re-init
    // rebuild their factors anyway; a plain hot start rebuilds them when iscal[4] says the factors are not its own) -- and the
    // tableau G goes BEHIND that image, one slot per variable and constraint ((nV + nC)^2 doubles, rsqp_state_bytes).
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
        if (tid == 0) { int *isc = pi + nV + 3 * nC; isc[1] = nFR; isc[2] = nAC; isc[3] = status; isc[4] = 2; }
        double *pm = img + rsqp_image_bytes(nV, nC) / 8;
        const int N = nV + nC;
#pragma unroll
        for (int a = 0; a < R; a++) {
            const int rs = rslot(a);
            if (rs >= 0) {
#pragma unroll
                for (int b = 0; b < CN; b++) { const int cs = cslot(b); if (cs >= 0) pm[rs + cs * N] = G[a][b]; }
            }
        }
    }
    // hot start: false = the stored state is not one this kernel wrote (the caller bails: the null-space kernel takes the member)
    __device__ __forceinline__ bool load_state(const double *img) {
        const long long ldx = rsqp_ld(nV), sT = nV < nC ? nV : nC, voff = 2 * ldx * nV + sT * ldx;
        const double *pv = img + voff;
        const int *pi = reinterpret_cast<const int *>(img + voff + 5LL * nV + 4LL * nC);
        const int *isc = pi + nV + 3 * nC;
        if (isc[4] != 2 || isc[3] == QPS_NOTINITIALISED) return false;
        nFR = isc[1]; nAC = isc[2]; status = isc[3];
        if (tid < nV) { const int v = tid; x[v] = pv[v]; g[v] = pv[nV + v]; lb[v] = pv[2 * nV + v]; ub[v] = pv[3 * nV + v]; yB[v] = pv[4 * nV + 3 * nC + v]; Sb[v] = pi[v]; }
        if (tid >= NT / 2 && tid - NT / 2 < nC) {
            const int i = tid - NT / 2;
            Ax[i] = pv[4 * nV + i]; lbA[i] = pv[4 * nV + nC + i]; ubA[i] = pv[4 * nV + 2 * nC + i]; yC[i] = pv[5 * nV + 3 * nC + i]; Sc[i] = pi[nV + i];
        }
        GSYNC();
        // (as the homotopy of the other engines begins) an inactive side that was infinite and now has a finite target only has
        // to stay clear of the iterate
        if (tid < nV) {
            const int v = tid;
            const int s = Sb[v];
            if (s != -1 && lb[v] <= -RSQP_INFTY && lbN[v] > -RSQP_INFTY) lb[v] = fmin(lbN[v], x[v] - RSQP_BOUND_RELAXATION);
            if (s != 1 && ub[v] >= RSQP_INFTY && ubN[v] < RSQP_INFTY) ub[v] = fmax(ubN[v], x[v] + RSQP_BOUND_RELAXATION);
        }
        if (tid >= NT / 2 && tid - NT / 2 < nC) {
            const int i = tid - NT / 2, s = Sc[i];
            if (s != -1 && lbA[i] <= -RSQP_INFTY && lbAN[i] > -RSQP_INFTY) lbA[i] = fmin(lbAN[i], Ax[i] - RSQP_BOUND_RELAXATION);
            if (s != 1 && ubA[i] >= RSQP_INFTY && ubAN[i] < RSQP_INFTY) ubA[i] = fmax(ubAN[i], Ax[i] + RSQP_BOUND_RELAXATION);
        }
        GSYNC();
        return true;
    }
};

template <int RV, int RC, int CV, int CC>
__global__ void __launch_bounds__(256, 1) small_qpg_kernel(QPPools P, int nq, int mode, int maxWSR) {
    typedef EngineG<RV, RC, CV, CC> ENG;
    // STATIC LDS (the image no longer depends on the problem's sizes): every vector's address is a link-time constant that
    // folds into the ds instructions -- with the dynamic window the ~25 array bases were register values, and the compiler
    // spilled some of them to scratch memory and re-loaded them inside the phases
    __shared__ __attribute__((aligned(16))) char smem_static[ENG::LDS_BYTES];
    const int q = (int)blockIdx.x;
    if (q >= nq) return;
    const QPDesc d = P.desc[q];
    ENG E;
    E.carve((lchar *)smem_static, d.nV, d.nC);
    E.nFR = E.nAC = 0; E.status = QPS_NOTINITIALISED; E.infeasible = E.unbounded = 0; E.nflips = 0; E.bail_reason = 0;
    E.debug_bail = P.k_debug_bail;
#ifdef RSQP_STAMPS
    E.tlast = clock64();
#endif
    int rcode = RET_OK, nWSR = 0;
    double obj = 0.0;
    const bool eligible = d.haveH && d.hreg == 0.0 && d.nV <= ENG::MAXV && d.nC <= ENG::MAXC;
    if (!eligible) {
        rcode = RET_BAIL; E.bail_reason = 10;
    } else {
        E.stage(P.Ajc + d.offAjc, P.Air + d.offAnz, P.Aval + d.offAnz, P.Hjc + d.offHjc, P.Hir + d.offHnz, P.Hval + d.offHnz,
                P.g + d.offV, P.lb + d.offV, P.ub + d.offV, P.lbA + d.offC, P.ubA + d.offC);
        E.init_G(mode != 0 ? P.state + d.offState + rsqp_image_bytes(d.nV, d.nC) / 8 : nullptr);
        if (!(E.hscale > 0.0)) { rcode = RET_BAIL; E.bail_reason = 10; }
        else if (mode != 0 && !E.load_state(P.state + d.offState)) { rcode = RET_BAIL; E.bail_reason = 12; }   // not this kernel's state
        else if (E.bounds_inconsistent()) {
            // (a hot start keeps the stored iterate: what the null-space kernels return in that case)
            E.infeasible = 1; rcode = RET_INFEASIBLE;
        } else {
            const bool hot = mode != 0;
            if (!hot) {
                E.status = QPS_PREPARINGAUXILIARYQP;
                rcode = E.setup_cold();
                if (rcode == RET_OK) E.status = QPS_AUXILIARYQPSOLVED;
            }
            if (rcode == RET_OK) rcode = E.homotopy(maxWSR, nWSR, hot);
        }
        if (rcode == RET_OK || rcode == RET_MAX_NWSR) obj = E.finish(rcode == RET_OK);
        else if (rcode != RET_BAIL) {     // infeasible / unbounded: the objective of the iterate the solve stopped at
            __syncthreads();
            obj = E.finish(false);
        }
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
        P.status[q] = E.infeasible ? 100 + st : (E.unbounded ? 200 + st : st);
        P.ret[q] = rcode;
        P.nwsr[q] = nWSR;
        P.nflips[q] = E.nflips;
        P.obj[q] = obj;
        if (!P.keep_state) {
            // no hot-start state wanted: mark the persistent image "not initialised" (layout of the explicit-inverse engine:
            // persist_doubles doubles, then the integer image Sb | Sc | AC | posAC | iscal, status in iscal[3])
            const long long xnp = EngineX<256, true>::persist_doubles(d.nV, d.nC);
            reinterpret_cast<int *>(P.state + d.offState + xnp)[d.nV + 3 * d.nC + 3] = QPS_NOTINITIALISED;
        }
    }
    if (P.keep_state) E.store_state(P.state + d.offState);
}
