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
//   A  out = G in; the two lanes that serve a row turn its entry into dx / dy / A dx and ONE ratio-test candidate each (the
//      lower side or the multiplier | the upper side): one division per lane; block argmin
//   B  homotopy step on the row state in REGISTERS; the lanes that hold row q publish it (= column q: G is symmetric)
//   C  decision (pivot tests, independence from |P a|, exchange, flip), rank-1 / rank-2 update of the register blocks, working
//      set, drift correction and the input of the next product -- all on registers, one LDS store per row
//   (+ every 8 changes and after exchanges / flips: A x, A'y - H x from the data; + 2 phases for an exchange)
// MI355X mapping (one problem per workgroup of 256 lanes = 4 waves, lane = 8 bi + bj: 32 row blocks x 8 column blocks): lane
// (bi, bj) holds the (RV + RC) x (CV + CC) block of G for variable rows bi RV + a / constraint rows bi RC + c and variable
// columns bj CV + b / constraint columns bj CC + c, and the same blocks of H, A, A' (exact products) -- all in REGISTERS;
// a product = LDS reads of the input slice, FMAs from registers, a sum over the 8 lanes of a row block by 3 DPP steps.
// The STATE of a row (x or A x, both limits and their targets, multiplier, working-set status, gradient data) lives in the
// registers of the two lanes bj = a and bj = a + 4 of its row block (a = 0..3: R = 4 rows per block), which update it
// identically; LDS only carries what crosses lanes: the input of the product, the published pivot row(s), the working set
// (masks), the argmin slots -- and x / y_C when a product with the data is due (first version of this kernel: every vector in
// LDS, ~30 dependent LDS round trips of the row owners per change; measured 15 k cycles per change, as much as round 3's).
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
    static constexpr int R = RV + RC, CN = CV + CC;
    static_assert(R == 4, "two lanes per row: lanes bj = a and bj = a + 4 serve row a of their row block");
    static constexpr int NVP = GI * RV > GJ * CV ? GI * RV : GJ * CV, NCP = GI * RC > GJ * CC ? GI * RC : GJ * CC;
    static constexpr int MAXV = GI * RV < GJ * CV ? GI * RV : GJ * CV, MAXC = GI * RC < GJ * CC ? GI * RC : GJ * CC;   // largest nV / nC
    // LDS: slot vectors (variable part, then constraint part ADJACENT: one array, index s or NVP + i -- a select between two
    // LDS arrays becomes a table of their addresses in scratch memory), integer working set, reduction slots, then ZERO-PADDED
    // dense copies of A (NCP x NVP, column major) and H (NVP x NVP): the exact products (every 8th change, the end of a solve)
    // read their blocks from there with compile-time trip counts and no bounds tests -- held in registers as well (round 3)
    // they cost 96 VGPRs that the pivots need
    // (leading dimensions ODD: with LDH = 96 / LDA = 32 the 8 column blocks of a wave hit the same LDS banks in every load of the
    //  exact products -- an 8-way conflict on 48 loads per lane, four waves sharing the LDS: ~3 k cycles per refresh)
    static constexpr int LDA = NCP + 1, LDH = NVP + 1, NS = NVP + NCP;
    static constexpr int LDS_BYTES = 8 * (4 * NS + NVP + 16) + 4 * (NS + 16) + 8 * (LDA * NVP + LDH * NVP);
    ldouble *xcol, *ycol, *inV, *uV, *u2V, *tV;      // xcol | ycol: x and y_C by column index (products with the data); in / u / u2: slot vectors
    ldouble *red;
    LDS int *Sb, *ired;                              // Sb | Sc adjacent: status by slot
    ldouble *Ad, *Hd;
    // ---- registers: the tableau block
    double G[R][CN];           // rows: the RV variable rows, then the RC constraint rows of bi; columns: CV variable, then CC constraint columns of bj
    // (ONE array: with four blocks the compiler fused the structurally equal branches "row q is a variable row" / "a constraint
    //  row" of the <2,2,8,8> build into one body that SELECTS the block's address -- which put all of G into scratch memory)
    // ---- registers: the state of my row (lanes bj and bj ^ 4 hold the same row and update it identically)
    double val, lo, up, loN, upN, ym, gcur, gtar, gyc, inm;     // x | A x, limits, their targets, multiplier, g, gN, A'y_C - H x, my input entry
    int st;                    // working-set status of my row: -1 lower, +1 upper, 0 free / inactive
    int my_a, my_idx, my_sl;   // row within the block, variable / constraint number, slot index into the adjacent LDS vectors
    bool isV, valid, lower;    // variable row | inside the problem | the lane that also serves the lower side and the multiplier
    double w4[R];              // one-hot over the rows of the block: my row
    int nV, nC, tid, bi, bj, wave;
    int nFR, nAC, status, infeasible, unbounded, nflips, bail_reason, parity;
    int debug_bail;             // >= 0: a hot start bails out before its debug_bail-th change (tests of the hand-over); else -1
    int since_refresh;          // working-set changes since A x and A'y - H x were last formed from the data; >= REFRESH: do it now
    static constexpr int REFRESH = 16;      // (8 in the null-space engines; the tableau forms A x and A'y - H x of the iterate from the data half as often: measured the same answers)
    double hscale;
    long long tlast;   // (-DRSQP_STAMPS builds: cycles per phase of block 0, tools/stamp_k_kernel.py)

    __device__ __forceinline__ void carve(lchar *base, int nV_, int nC_) {
        nV = nV_; nC = nC_;
        tid = (int)threadIdx.x; bi = tid >> LGJ; bj = tid & (GJ - 1); wave = tid >> 6;
        ldouble *p = (ldouble *)base;
        xcol = p; p += NVP; ycol = p; p += NCP;
        inV = p; p += NS; uV = p; p += NS; u2V = p; p += NS; tV = p; p += NVP;
        red = p; p += 16;
        LDS int *ip = (LDS int *)p;
        Sb = ip; ip += NS; ired = ip; ip += 16;
        Ad = (ldouble *)ip;
        Hd = Ad + LDA * NVP;
        parity = 0;
        my_a = bj & 3; lower = bj < 4;
        isV = my_a < RV;
        my_idx = isV ? bi * RV + my_a : bi * RC + (my_a - RV);
        valid = isV ? my_idx < nV : my_idx < nC;
        my_sl = isV ? my_idx : NVP + my_idx;
#pragma unroll
        for (int a = 0; a < R; a++) w4[a] = opaque(my_a == a ? 1.0 : 0.0);
    }

    // ------------------------------------------------------------------ building blocks
    template <int R_, int Cn> __device__ __forceinline__ static void mv(const double (&B)[R_][Cn], const double (&xv)[Cn], double (&acc)[R_]) {
#pragma unroll
        for (int a = 0; a < R_; a++)
#pragma unroll
            for (int b = 0; b < Cn; b++) acc[a] = fma(B[a][b], xv[b], acc[a]);
    }
    template <int R_> __device__ __forceinline__ static void rowsum(double (&acc)[R_]) {     // over the 8 lanes of a row block
#pragma unroll
        for (int a = 0; a < R_; a++) acc[a] = allreduce_sum<LGJ>(acc[a]);
    }
    template <int R_> __device__ __forceinline__ static void zero(double (&acc)[R_]) {
#pragma unroll
        for (int a = 0; a < R_; a++) acc[a] = 0.0;
    }
    // acc[my_a] as the product with the lane's one-hot weights: exact, and -- unlike a chain of selects, which the compiler
    // turns back into an indexed load from a copy of acc in SCRATCH memory -- in registers
    __device__ __forceinline__ double mine(const double (&acc)[R]) const {
        double v = acc[0] * w4[0];
#pragma unroll
        for (int a = 1; a < R; a++) v = fma(acc[a], w4[a], v);
        return v;
    }
    template <int Cn> __device__ __forceinline__ void ldcols(const ldouble *v, double (&xv)[Cn]) const {
#pragma unroll
        for (int b = 0; b < Cn; b++) xv[b] = v[bj * Cn + b];
    }
    __device__ __forceinline__ static double clampinf(double v) { return v > RSQP_INFTY ? RSQP_INFTY : (v < -RSQP_INFTY ? -RSQP_INFTY : v); }
    // sums over a workgroup vector, formed by EVERY wave from the published LDS operands: all lanes agree, no barrier
    __device__ __forceinline__ double wdotV(const ldouble *a, const ldouble *b) const {
        const int l = tid & 63;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < (NVP + 63) / 64; k++) { const int i = l + 64 * k; if (i < NVP) s = fma(a[i], b[i], s); }
        return allreduce_sum<6>(s);
    }
    // |u_FR|^2 and |a_FR|^2 of the pivot column / the incoming row over the FREE variables (a: row `arow` of A, the unit
    // vector of variable -arow - 2 when arow <= -2, nothing when arow == -1)
    __device__ __forceinline__ void free_norms(int arow, double &pn2, double &na2) const {
        const int l = tid & 63;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < (NVP + 63) / 64; k++) {
            const int v = l + 64 * k;
            if (v < nV && Sb[v] == 0) {
                const double u = uV[v];
                const double a = arow >= 0 ? Ad[arow + v * LDA] : (v == -arow - 2 ? 1.0 : 0.0);
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
    // true if the predicate holds in any lane of the workgroup (two barriers: rare paths only)
    __device__ __forceinline__ bool block_any(bool pred) {
        const int wany = __any(pred ? 1 : 0) ? 1 : 0;
        if ((tid & 63) == 0) ired[wave] = wany;
        GSYNC();
        int any = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) any |= ired[w];
        GSYNC();
        return any != 0;
    }
    // sum of one value per lane over the workgroup (fixed order: lanes by DPP tree, then waves 0..3)
    __device__ __forceinline__ double block_sum(double v) {
        const double ws = allreduce_sum<6>(v);
        parity ^= 8;
        if ((tid & 63) == 0) red[parity + wave] = ws;
        GSYNC();
        double s = red[parity];
#pragma unroll
        for (int w = 1; w < NW; w++) s += red[parity + w];
        return s;
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
        for (int k = tid; k < NS; k += NT) { inV[k] = 0.0; uV[k] = 0.0; u2V[k] = 0.0; Sb[k] = k < NVP ? -1 : 0; }
        for (int k = tid; k < NVP; k += NT) { xcol[k] = 0.0; tV[k] = 0.0; }
        for (int k = tid; k < NCP; k += NT) ycol[k] = 0.0;
        // the state of my row
        val = 0.0; ym = 0.0; gcur = 0.0; gyc = 0.0; inm = 0.0; st = isV ? -1 : 0; lo = 0.0; up = 0.0;
        gtar = (valid && isV) ? g_[my_idx] : 0.0;
        if (isV) { loN = valid ? clampinf(lb_[my_idx]) : 0.0; upN = valid ? clampinf(ub_[my_idx]) : 0.0; }
        else { loN = valid ? clampinf(lbA_[my_idx]) : -RSQP_INFTY; upN = valid ? clampinf(ubA_[my_idx]) : RSQP_INFTY; }
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
        bool asym = false;
#pragma unroll
        for (int a = 0; a < RV; a++)
#pragma unroll
            for (int b = 0; b < CV; b++) {
                const int r = bi * RV + a, c = bj * CV + b;
                if (Hd[r + c * LDH] != Hd[c + r * LDH]) asym = true;
            }
        hscale = block_any(asym) ? 0.0 : hm;      // (hscale = 0 makes the kernel bail)
    }
    __device__ __forceinline__ bool bounds_inconsistent() { return block_any(valid && loN > upN + RSQP_EPS); }
    // G: -K (cold start: S is empty) or the stored tableau (hot start; pm = the extension behind the null-space image). ONE
    // definition site for the register block (a second one doubled the registers the allocator kept busy)
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
        if (block_any(valid && isV && loN <= -RSQP_INFTY && upN >= RSQP_INFTY)) { bail_reason = 11; return RET_BAIL; }   // a free variable in the cold working set
        if (isV) {
            st = (!valid || loN > -RSQP_INFTY) ? -1 : 1;
            lo = (!valid || st == -1) ? 0.0 : fmin(loN, -RSQP_BOUND_RELAXATION);
            up = (!valid || st == 1) ? 0.0 : fmax(upN, RSQP_BOUND_RELAXATION);
        } else {
            st = 0;
            lo = fmin(loN, -RSQP_BOUND_RELAXATION); up = fmax(upN, RSQP_BOUND_RELAXATION);
        }
        if (lower) Sb[my_sl] = st;
        nFR = nAC = 0;
        return RET_OK;
    }

    // ------------------------------------------------------------------ products with the data
    // x and y_C by column index into LDS (the caller closes with the barrier)
    __device__ __forceinline__ void publish_iterate() {
        if (lower) { if (isV) xcol[my_idx] = val; else ycol[my_idx] = ym; }
    }
    // comb[a] = (A'y_C - H x) of variable row a | (A x) of constraint row a - RV; hx[a] = (H x) of variable row a: from the
    // zero-padded dense copies in LDS, summed over the row block
    __device__ __forceinline__ void exact_products(double (&comb)[R], double (&hx)[RV], double (&aty)[RV]) const {
        double xv[CV], yc[CC], ax[RC];
        ldcols<CV>(xcol, xv); ldcols<CC>(ycol, yc);
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
#pragma unroll
        for (int a = 0; a < RV; a++) comb[a] = aty[a] - hx[a];
#pragma unroll
        for (int a = 0; a < RC; a++) comb[RV + a] = ax[a];
    }
    // drift correction (gradient of the current QP from stationarity, active limits := A x) and my entry of the next input
    // keep_data (first pass of a hot start): g, A x and the limits of the active constraints stay what the previous solve left
    __device__ __forceinline__ void make_input(bool keep_data) {
        if (isV) {
            const double gv = keep_data ? gcur : gyc + ym;
            if (keep_data) gyc = gv - ym;
            gcur = gv;
            inm = valid ? (st == 0 ? -(gtar - gv) : (st == -1 ? loN - lo : upN - up)) : 0.0;
        } else {
            // (value selects, never "if (..) lo = ..; else up = ..": the compiler merges such stores into ONE store through a
            //  selected address, which pins the row state -- and every member near it -- in scratch memory)
            lo = (!keep_data && st == -1) ? val : lo;
            up = (!keep_data && st == 1) ? val : up;
            inm = st == 0 ? 0.0 : (st == -1 ? loN - lo : upN - up);
        }
        if (lower) inV[my_sl] = inm;
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
    __device__ __forceinline__ static double rowof(const ldouble *o, int bi_, int a) { return a < RV ? o[bi_ * RV + a] : o[NVP + bi_ * RC + (a - RV)]; }
    __device__ __forceinline__ static double colof(const ldouble *o, int bj_, int b) { return b < CV ? o[bj_ * CV + b] : o[NVP + bj_ * CC + (b - CV)]; }
    __device__ __forceinline__ int lds_slot(int q) const { return q < nV ? q : NVP + (q - nV); }
    // position of slot q in the lane grid: row block / row inside it, column block / column inside it
    struct Pos { int rb, a, cb, b; };
    __device__ __forceinline__ Pos pos_of(int q) const {
        const bool qv = q < nV;
        const int qq = qv ? q : q - nV;
        Pos p;
        p.rb = qv ? qq / RV : qq / RC; p.a = qv ? qq - p.rb * RV : RV + (qq - p.rb * RC);
        p.cb = qv ? qq / CV : qq / CC; p.b = qv ? qq - p.cb * CV : CV + (qq - p.cb * CC);
        return p;
    }
    // the 8 lanes that hold row q of G publish it (= column q: G is symmetric)
    __device__ __forceinline__ void publish_row(int q, ldouble *o) {
        const Pos P = pos_of(q);
        if (bi == P.rb) {
            double w[R];
#pragma unroll
            for (int a = 0; a < R; a++) w[a] = opaque(a == P.a ? 1.0 : 0.0);
#pragma unroll
            for (int b = 0; b < CN; b++) {
                double v = G[0][b] * w[0];
#pragma unroll
                for (int a = 1; a < R; a++) v = fma(G[a][b], w[a], v);
                if (b < CV) o[bj * CV + b] = v; else o[NVP + bj * CC + (b - CV)] = v;
            }
        }
    }
    // 1 / x by v_rcp_f64 and two Newton steps (~2^-52 relative; the same bits in every lane): the pivot's reciprocal is not compared
    // with anything, an IEEE division (a chain of ~12 dependent instructions) buys nothing here
    __device__ __forceinline__ static double recip(double x) {
        double y = __builtin_amdgcn_rcp(x);
        double e = fma(-x, y, 1.0); y = fma(y, e, y);
        e = fma(-x, y, 1.0); y = fma(y, e, y);
        return y;
    }
    // principal pivot on slot q with the published column u, pi = its entry q, sgn = +1 (q enters S) / -1 (q leaves):
    //   G <- G0 - (1 / pi) u~ u~',  G0 = G with row and column q zeroed, u~ = u except u~_q = sgn.
    // Row q is zeroed by the 8 lanes that hold it (a branch per row: the other waves skip it), column q by a 0 / 1 factor per
    // column inside the update (one extra multiplication per entry: a lane-varying register index would go through scratch)
    __device__ __forceinline__ void pivot1(int q, double sgn, double pi) {
        const Pos P = pos_of(q);
        const int aq = bi == P.rb ? P.a : -1, bq = bj == P.cb ? P.b : -1;
        const double c = -recip(pi);
        double tr[R], uc[CN], ck[CN];
#pragma unroll
        for (int a = 0; a < R; a++) tr[a] = c * (aq == a ? sgn : rowof(uV, bi, a));
#pragma unroll
        for (int b = 0; b < CN; b++) { uc[b] = bq == b ? sgn : colof(uV, bj, b); ck[b] = bq == b ? 0.0 : 1.0; }
#pragma unroll
        for (int a = 0; a < R; a++)
            if (aq == a) {
#pragma unroll
                for (int b = 0; b < CN; b++) G[a][b] = 0.0;
            }
#pragma unroll
        for (int a = 0; a < R; a++)
#pragma unroll
            for (int b = 0; b < CN; b++) G[a][b] = fma(tr[a], uc[b], G[a][b] * ck[b]);
    }
    // 2 x 2 block pivot on (p, q): columns u2 of p and u of q published; W = P^-1 = [w11 w12; w12 w22] of
    // P = [G_pp G_pq; G_pq G_qq]; sp / sq = +1 (enters S) / -1 (leaves S):  G <- G00 - U~ W U~',  U~ = [u_p u_q] with rows p, q = diag(sp, sq)
    __device__ __forceinline__ void pivot2(int p, double sp, int q, double sq, double w11, double w12, double w22) {
        double rp[R], rq[R], rk[R];
        bool rowhit = false;
#pragma unroll
        for (int a = 0; a < R; a++) {
            const int s = rslot(a); const bool hp = s == p, hq = s == q; rowhit |= hp | hq;
            rp[a] = hp ? sp : (hq ? 0.0 : rowof(u2V, bi, a)); rq[a] = hq ? sq : (hp ? 0.0 : rowof(uV, bi, a));
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
            const double up_ = hp ? sp : (hq ? 0.0 : colof(u2V, bj, b)), uq = hq ? sq : (hp ? 0.0 : colof(uV, bj, b));
            const double cp = fma(w11, up_, w12 * uq), cq = fma(w12, up_, w22 * uq), ck = (hp || hq) ? 0.0 : 1.0;
#pragma unroll
            for (int a = 0; a < R; a++) G[a][b] = fma(-rp[a], cp, fma(-rq[a], cq, G[a][b] * ck));
        }
    }
    // (G in) of my row: LDS reads of the input slice, FMAs from registers, sum over the row block
    __device__ __forceinline__ double g_times_mine() const {
        double xin[CN], acc[R];
#pragma unroll
        for (int b = 0; b < CN; b++) xin[b] = colof(inV, bj, b);
        zero<R>(acc);
        mv<R, CN>(G, xin, acc);
        rowsum<R>(acc);
        return mine(acc);
    }

    // ------------------------------------------------------------------ phase C: one working-set change
    // kind 1 constraint idx leaves | 2 bound of idx leaves | 3 constraint idx enters at `side` | 4 variable idx gets fixed at `side`;
    // the column of the slot is published in uV. Ends with the working set updated in the registers of the rows concerned
    // and in LDS (the masks of the next tests)
    __device__ __forceinline__ int change(int kind, int idx, int side) {
        const int q = (kind == 1 || kind == 3) ? nV + idx : idx;
        const double pi = uV[lds_slot(q)];
        const bool myrow = valid && my_idx == idx && (isV == (kind == 2 || kind == 4));      // my row is the one that changes
        int pk = 0, pidx = -1;                // exchange partner: 1 constraint / 2 bound
        double ynew = 0.0;
        bool flip = false;
        if (kind == 1) {
            double d2, dummy;
            free_norms(-1, d2, dummy);
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
            if (block_any(myrow && fabs(st == -1 ? upN : loN) >= RSQP_INFTY)) return RET_UNBOUNDED;
            up = (myrow && st == -1) ? val : up;
            lo = (myrow && st == 1) ? val : lo;
            ym = myrow ? 0.0 : ym;
            st = myrow ? -st : st;
            if (myrow && lower) Sb[my_sl] = st;
            nflips++;
            since_refresh = REFRESH;
            return RET_OK;
        }
        if (kind >= 3) {
            // independence of the incoming row a from the working set. |P a| -- the free-variable part of the pivot column -- is of
            // first order in |Z'a| (|Z'a| / lmax <= |P a| <= |Z'a| / lmin); in the band between "clearly independent" and
            // "clearly dependent" the residual r = a_FR - A_AC,FR' xi_C of the row's representation by the active rows decides
            // (|Z'a| <= |r| <= cond |Z'a|, the test of round 3's kernel)
            const double sg = kind == 3 ? -1.0 : 1.0;      // xi = -column for a constraint that enters S, +column for a variable that leaves it
            int li;
            double pn2, na2;
            free_norms(kind == 3 ? idx : -idx - 2, pn2, na2);
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
                    const double yc = Sb[NVP + i] != 0 ? sg * uV[NVP + i] : 0.0;
#pragma unroll
                    for (int a = 0; a < RV; a++) at[a] = fma(Ad[i + (bi * RV + a) * LDA], yc, at[a]);
                }
                rowsum<RV>(at);
                double comb[R];
#pragma unroll
                for (int a = 0; a < R; a++) comb[a] = a < RV ? at[a] : 0.0;
                if (lower && isV) {
                    const double a = valid ? (kind == 3 ? Ad[idx + my_idx * LDA] : (my_idx == idx ? 1.0 : 0.0)) : 0.0;
                    tV[my_idx] = (valid && st == 0) ? a - mine(comb) : 0.0;
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
                const double xi_m = (valid && st != 0) ? sgn * sg * uV[my_sl] : 0.0;     // (fixed variables and active constraints)
                if (lower && valid && st != 0) {
                    const double num = st == -1 ? ym : -ym, den = st == -1 ? xi_m : -xi_m;
                    if (den > RSQP_EPS_DEN) { bt = (num > 0.0 ? num : 0.0) / den; bid = isV ? nC + my_idx : my_idx; }
                }
                block_argmin(bt, bid);
                if (bid == 0x7fffffff) { bail_reason = 8; return RET_BAIL; }      // (no partner: the null-space engine decides "infeasible")
                if (bid < nC) { pk = 1; pidx = bid; } else { pk = 2; pidx = bid - nC; }
                ym -= bt * xi_m;
                ynew = sgn * bt;
                const int p = pk == 1 ? nV + pidx : pidx;
                publish_row(p, u2V);
                GSYNC();
                const int psl = lds_slot(p);
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
        // ---- the working set: the rows concerned, in registers and (masks of the next tests) in LDS
        const bool leaves = kind == 1 || kind == 2;
        const bool prow = pk != 0 && valid && my_idx == pidx && isV == (pk == 2);
        st = myrow ? (leaves ? 0 : side) : (prow ? 0 : st);
        ym = myrow ? (leaves ? 0.0 : ynew) : (prow ? 0.0 : ym);
        if ((myrow || prow) && lower) Sb[my_sl] = st;
        nAC += (kind == 3 ? 1 : 0) - (kind == 1 ? 1 : 0) - (pk == 1 ? 1 : 0);
        nFR += (kind == 2 ? 1 : 0) - (kind == 4 ? 1 : 0) + (pk == 2 ? 1 : 0);
        return RET_OK;
    }

    __device__ __forceinline__ int homotopy(int maxit, int &nWSR, bool hot) {
        int iter = 0, rcode = RET_OK;
        status = QPS_PERFORMINGHOMOTOPY;
        since_refresh = REFRESH;
        for (;;) {
            // ---- x exactly on its active bounds; (exact products); drift correction + input of the product
            if (isV && st != 0) val = st == -1 ? lo : up;
            const bool keep = hot && iter == 0;
            if (since_refresh >= REFRESH) {
                publish_iterate();
                GSYNC();
                if (!keep) {
                    double comb[R], hx[RV], aty[RV];
                    exact_products(comb, hx, aty);
                    const double m = mine(comb);
                    if (isV) gyc = m; else val = m;
                }
                since_refresh = 0;
                STAMP(30);
            }
            make_input(keep);
            GSYNC();
            STAMP(38);
            // ---- phase A: out = G in; my row's dx / dy / A dx and ONE ratio-test candidate per lane
            double bt = 1.0, dval, dy, hd;
            int bid = 0x7fffffff;
            {
                const double out = g_times_mine();
                STAMP(31);
                const double dg = gtar - gcur;
                if (isV) { dval = st == 0 ? out : inm; dy = st == 0 ? 0.0 : dg - out; hd = st == 0 ? -dg : -out; }
                else { dval = st != 0 ? inm : -out; dy = st != 0 ? -out : 0.0; hd = 0.0; }
                double num, den; int id; bool ok;
                if (lower) {
                    // the multiplier of an active row reaches zero | the lower limit of an inactive row is hit
                    num = st != 0 ? (st == -1 ? ym : -ym) : val - lo;
                    den = st != 0 ? (st == -1 ? -dy : dy) : (loN - lo) - dval;
                    id = st != 0 ? (isV ? nC + my_idx : my_idx) : (isV ? 3 * nC + nV + my_idx : nC + nV + my_idx);
                    ok = valid && (st != 0 || loN > -RSQP_INFTY);
                } else {
                    num = up - val; den = dval - (upN - up);
                    id = isV ? 3 * nC + 2 * nV + my_idx : 2 * nC + nV + my_idx;
                    ok = valid && st == 0 && upN < RSQP_INFTY;
                }
                cand(num, den, id, ok, bt, bid);
            }
            if (!(bt < 1.0)) { bt = 1.0; bid = 0x7fffffff; }
            STAMP(32);
            block_argmin(bt, bid);
            STAMP(33);
            // ---- phase B: decode, homotopy step on the row state, the column of the change
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
            ym += tau * dy;
            {
                const double vn = val + tau * dval;
                const double l = lo + tau * (loN - lo), u = up + tau * (upN - up);
                const bool hit = !done && !cap && valid && my_idx == idx && (isV ? kind == 4 : kind == 3);   // the blocking quantity sits exactly on its limit
                // (done: the data ARE the targets now, x exactly on its active bounds; A x of the constraint rows is formed from the data below)
                val = done ? (isV ? (st == -1 ? loN : (st == 1 ? upN : vn)) : val) : vn;
                gcur = done ? gtar : gcur + tau * (gtar - gcur);
                gyc -= tau * hd;                                  // A'y_C - H x follows the step
                lo = done ? loN : ((hit && side == -1) ? vn : l);
                up = done ? upN : ((hit && side == 1) ? vn : u);
            }
            if (done || cap) {
                if (done) status = QPS_SOLVED; else rcode = RET_MAX_NWSR;
                break;
            }
            if (hot && iter == debug_bail) { bail_reason = 13; rcode = RET_BAIL; break; }      // (test hook, see P.k_debug_bail)
            publish_row((kind == 1 || kind == 3) ? nV + idx : idx, uV);
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
    // refine = false (iteration limit, infeasible, unbounded): only A x and the objective
    __device__ __forceinline__ double finish(bool refine) {
        double hxm = 0.0;
        // pass 0 (refine only): residuals from the data -> correction out = G in -> x_FR, y_AC; pass 1: A x, H x, A'y_C of the final
        // iterate -> multipliers of the fixed variables, A x, H x for the objective (ONE call site of the exact products)
        for (int pass = refine ? 0 : 1; pass < 2; pass++) {
            publish_iterate();
            GSYNC();
            double comb[R], hx[RV], aty[RV], hx4[R];
            exact_products(comb, hx, aty);
#pragma unroll
            for (int a = 0; a < R; a++) hx4[a] = a < RV ? hx[a] : 0.0;
            const double m = mine(comb), h = mine(hx4);      // variable row: A'y_C - H x, H x; constraint row: A x
            if (pass == 0) {
                if (isV) inm = (valid && st == 0) ? -(gtar - m) : 0.0;             // -(gN + H x - A'y_C)
                else inm = st != 0 ? (st == 1 ? upN : loN) - m : 0.0;
                if (lower) inV[my_sl] = inm;
                GSYNC();
                const double out = g_times_mine();
                if (isV) { if (valid && st == 0) val += out; }
                else if (st != 0) ym -= out;
                GSYNC();
            } else {
                hxm = h;
                if (isV) { if (refine && valid) ym = st != 0 ? gtar - m : 0.0; }   // gN + H x - A'y_C on the fixed variables
                else val = m;
            }
        }
        return block_sum((lower && isV && valid) ? val * fma(0.5, hxm, gtar) : 0.0);
    }

    // ------------------------------------------------------------------ results and persistent state (hot starts)
    __device__ __forceinline__ void write_results(const QPPools &P, const QPDesc &d) const {
        if (lower && valid) {
            if (isV) { P.x[d.offV + my_idx] = val; P.ws_b[d.offV + my_idx] = st; P.y[d.offV + d.offC + my_idx] = ym; }
            else { P.y[d.offV + d.offC + nV + my_idx] = ym; P.ws_c[d.offC + my_idx] = st; }
        }
    }
    // The state lives in the layout of the explicit-inverse engine (qp_small_x.h carve: factors | x g lb ub | A x lbA ubA | y,
    // then Sb Sc AC posAC iscal as ints), so that engine's hot-start modes work on it unchanged (new matrices / warm re-init
    // rebuild their factors anyway; a plain hot start rebuilds them when iscal[4] says the factors are not its own) -- and the
    // tableau G goes BEHIND that image, one slot per variable and constraint ((nV + nC)^2 doubles, rsqp_state_bytes).
    __device__ __forceinline__ void store_state(double *img) const {
        const long long ldx = rsqp_ld(nV), sT = nV < nC ? nV : nC, voff = 2 * ldx * nV + sT * ldx;
        double *pv = img + voff;
        int *pi = reinterpret_cast<int *>(img + voff + 5LL * nV + 4LL * nC);      // = persist_doubles of that engine
        if (lower && valid) {
            if (isV) {
                const int v = my_idx;
                pv[v] = val; pv[nV + v] = gcur; pv[2 * nV + v] = lo; pv[3 * nV + v] = up; pv[4 * nV + 3 * nC + v] = ym; pi[v] = st;
            } else {
                const int i = my_idx;
                pv[4 * nV + i] = val; pv[4 * nV + nC + i] = lo; pv[4 * nV + 2 * nC + i] = up; pv[5 * nV + 3 * nC + i] = ym; pi[nV + i] = st;
            }
        }
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
        if (valid) {
            if (isV) {
                const int v = my_idx;
                val = pv[v]; gcur = pv[nV + v]; lo = pv[2 * nV + v]; up = pv[3 * nV + v]; ym = pv[4 * nV + 3 * nC + v]; st = pi[v];
            } else {
                const int i = my_idx;
                val = pv[4 * nV + i]; lo = pv[4 * nV + nC + i]; up = pv[4 * nV + 2 * nC + i]; ym = pv[5 * nV + 3 * nC + i]; st = pi[nV + i];
            }
            // (as the homotopy of the other engines begins) an inactive side that was infinite and now has a finite target only has
            // to stay clear of the iterate
            if (st != -1 && lo <= -RSQP_INFTY && loN > -RSQP_INFTY) lo = fmin(loN, val - RSQP_BOUND_RELAXATION);
            if (st != 1 && up >= RSQP_INFTY && upN < RSQP_INFTY) up = fmax(upN, val + RSQP_BOUND_RELAXATION);
        }
        if (lower) Sb[my_sl] = st;
        return true;
    }
};

template <int RV, int RC, int CV, int CC>
__global__ void __launch_bounds__(256, 1) small_qpg_kernel(QPPools P, int nq, int mode, int maxWSR) {
    typedef EngineG<RV, RC, CV, CC> ENG;
    // STATIC LDS (the image does not depend on the problem's sizes): every vector's address is a link-time constant that
    // folds into the ds instructions
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
        if (rcode != RET_BAIL) obj = E.finish(rcode == RET_OK);
    }
    const int tid = (int)threadIdx.x;
    if (rcode == RET_BAIL) {
        if (tid == 0) { P.ret[q] = RET_BAIL; P.nflips[q] = E.bail_reason; P.nwsr[q] = 1000 + E.bail_reason; }    // the null-space kernel takes this member over
        //                                  (nwsr: overwritten by it; read by tools/bail_hist.py under RSQP_SMALL_KKT_ONLY=1)
        return;
    }
    E.write_results(P, d);
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
