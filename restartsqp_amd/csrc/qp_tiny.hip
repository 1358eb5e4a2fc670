// qp_tiny.hip -- the hs071-scale QP engine of round 4: problems of at most 8 variables and MC <= 8 constraints, 8 lanes per
// problem (8 problems per wave), the TABLEAU formulation of qp_small_g.h with everything in REGISTERS.
//
// Replaces, for these sizes, the qpOASES 3.2.1 SQProblem::init / hotstart calls made at reference
// src/qpOASESInterface.cpp:155,180,184,191,197,204 (the batched headline workload of bench.py: the hs071 QP through the
// QPhandler formulation, 8 variables x 2 constraints; the single-QP boundary of the hs071 SQP run). Same homotopy, ratio
// tests, tie breaks and drift correction as the other engines and the CPU restatement; the linear algebra is ONE symmetric
// N x N matrix, N = 8 + MC, in fixed slots (variable v -> v, constraint i -> 8 + i):
//      G = - SWEEP_S(K),  K = [H A'; A 0],  S = free variables + active constraints        (derivation: qp_small_g.h)
// so that a step direction is ONE product out = G in, a working-set change ONE principal pivot (an exchange one 2 x 2 block
// pivot, a flip none), and ONE step of iterative refinement with residuals from the data ends a solve.
//
// MI355X mapping: lane l of a group of 8 holds row l of the variable part of G and row l of the constraint part (l < MC),
// i.e. 2 N doubles, plus the state of those two rows (x | A x, limits, targets, multiplier, status, gradient data).
// What crosses lanes goes through the wave's permute network, never through memory:
//   * the input vector of the product: every lane fetches the N entries with ds_bpermute (no LDS allocation behind it);
//   * the pivot row: fetched from the lane that holds it, N entries -- afterwards EVERY lane has the whole row, so the pivot,
//     the curvature and independence tests and the update of its own rows are lane-local arithmetic;
//   * ratio-test argmin and the few sums: 3 DPP steps over the 8 lanes.
// No barrier, no LDS image of the solver state: LDS only holds a dense copy of K per problem (N^2 doubles), read when the
// data are needed again (set-up, |a_FR|, exact products every 8 changes and at the end). The Givens / TQ kernel this
// replaces for the headline workload kept Q, R, T and 20 vectors per problem in LDS (2.4 KB) and spent ~5.5 k wave
// instructions per 8 problems at two waves per SIMD; this one spends ~1.2 k.
//
// All four call shapes: cold start, hot start on new vectors (state = G + row state, kept in the problem's state block), hot
// start with new matrices and warm re-initialisation from (x0, y0, guessed bounds) -- the latter two build G for the guessed
// working set by a sequence of pivots from -K (warm_setup), as setup_aux of the CPU restatement builds its factors.
#include <cstdlib>

#include "rsqp_internal.h"
#include "rsqp_kkt.h"

#define LDS __attribute__((address_space(3)))
typedef LDS double ldouble;
typedef LDS char lchar;

#define TSYNC()                                                  \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)

// diagnostic build only (-DRSQP_STAMPS, tools/stamp_tiny_kernel.py): cycles per phase of wave 0 of block 0
#ifdef RSQP_STAMPS
__device__ unsigned long long g_tiny_stamps[16];
#define TSTAMP(k)                                                                                        \
    do {                                                                                                 \
        long long t_ = clock64();                                                                        \
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_tiny_stamps[k], (unsigned long long)(t_ - tlast)); \
        tlast = t_;                                                                                      \
    } while (0)
extern "C" void rsqp_debug_tiny_stamps(unsigned long long *out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tiny_stamps), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tiny_stamps), z, sizeof(z));
    }
}
#else
#define TSTAMP(k) do { } while (0)
#endif

namespace {

// ---- exchanges inside a group of 8 lanes by DPP permutations (xor 1, xor 2, i <-> 7 - i): see qp_small.hip
template <int S> __device__ __forceinline__ int xchg_i32(int x) {
    if constexpr (S == 0) return __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false);       // quad_perm [1,0,3,2]
    else if constexpr (S == 1) return __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
    else return __builtin_amdgcn_update_dpp(x, x, 0x141, 0xf, 0xf, false);                       // row_half_mirror
}
template <int S> __device__ __forceinline__ double xchg_f64(double x) {
    return __hiloint2double(xchg_i32<S>(__double2hiint(x)), xchg_i32<S>(__double2loint(x)));
}
__device__ __forceinline__ double sum8(double v) { v += xchg_f64<0>(v); v += xchg_f64<1>(v); v += xchg_f64<2>(v); return v; }
__device__ __forceinline__ double max8(double v) { v = fmax(v, xchg_f64<0>(v)); v = fmax(v, xchg_f64<1>(v)); v = fmax(v, xchg_f64<2>(v)); return v; }
__device__ __forceinline__ int or8(int v) { v |= xchg_i32<0>(v); v |= xchg_i32<1>(v); v |= xchg_i32<2>(v); return v; }
template <int S> __device__ __forceinline__ void argmin_step(double &t, int &id) {
    const double t2 = xchg_f64<S>(t);
    const int id2 = xchg_i32<S>(id);
    if (t2 < t || (t2 == t && id2 < id)) { t = t2; id = id2; }
}
__device__ __forceinline__ void argmin8(double &t, int &id) { argmin_step<0>(t, id); argmin_step<1>(t, id); argmin_step<2>(t, id); }
// value of lane `src` (0..7) of my group of 8
__device__ __forceinline__ double fetch8(double v, int src) { return __shfl(v, src, 8); }
__device__ __forceinline__ int fetch8i(int v, int src) { return __shfl(v, src, 8); }
// a value the compiler must not look through (one-hot weights would otherwise become indexed loads from scratch memory)
__device__ __forceinline__ double opaque(double v) { asm volatile("" : "+v"(v)); return v; }

__device__ __forceinline__ double clampinf(double v) { return v > RSQP_INFTY ? RSQP_INFTY : (v < -RSQP_INFTY ? -RSQP_INFTY : v); }
__device__ __forceinline__ double recip(double x) {      // v_rcp_f64 + two Newton steps: ~2^-52 relative, the same bits in every lane
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0); y = fma(y, e, y);
    e = fma(-x, y, 1.0); y = fma(y, e, y);
    return y;
}

constexpr int MV = 8;       // variable slots = lanes of a group
#ifndef TINY_BLOCK
#define TINY_BLOCK 256       // threads per workgroup = 32 problems (no barrier, no cross-group traffic: the size only sets the LDS granule)
#endif
constexpr int TB = TINY_BLOCK, TG = TINY_BLOCK / 8;

// GL = false: the tableau in registers (lane l holds rows l of both parts: 2 N doubles). GL = true: the tableau in LDS (row major,
// beside the dense K), every lane reads / updates its own two rows there and reads any other row directly -- no permutes for the
// pivot rows, no select chains for "entry q of my rows", and ~80 VGPRs less: a third wave per SIMD. A wave of this kernel takes
// its ~45 k cycles whatever runs beside it (the same count for 1 QP and for 65 536), so throughput is resident waves / that latency.
template <int MC, bool GL = false>
struct EngineT {
    static constexpr int N = MV + MC, REFRESH = 8;
    static constexpr int NS = N;          // row stride of the LDS tableau
    // ---- registers: my two rows of the tableau (GL: unused)
    double GV[GL ? 1 : N], GC[GL ? 1 : N];
    ldouble *Gl;                          // GL: the tableau of my problem in LDS
    __device__ __forceinline__ double gv(int k) const { if constexpr (GL) return Gl[l * NS + k]; else return GV[k]; }
    __device__ __forceinline__ double gc(int k) const { if constexpr (GL) return l < MC ? Gl[(MV + l) * NS + k] : 0.0; else return GC[k]; }
    __device__ __forceinline__ void set_gv(int k, double v) { if constexpr (GL) Gl[l * NS + k] = v; else GV[k] = v; }
    __device__ __forceinline__ void set_gc(int k, double v) { if constexpr (GL) { if (l < MC) Gl[(MV + l) * NS + k] = v; } else GC[k] = v; }
    // ---- state of my variable row and (l < nC) my constraint row
    double xv, lo, up, loN, upN, yv, g, gN, gy, inV;
    double ax, loA, upA, cloN, cupN, yc, inC;
    int sv, sc;
    double dV, dC;                      // the diagonal entries of my two rows (G_ll, G_{8+l,8+l}): the pivot of a change is one of them
    // ---- group-uniform
    int nV, nC, l, fmask, amask;        // free variables / active constraints as bit masks (replicated in every lane)
    int status, infeasible, unbounded, nflips, since_refresh;
    double hscale, hreg;
    ldouble *Kd;                        // dense K of my problem in LDS, row major N x N (H + hreg I | A' ; A | 0)
    long long tlast;                    // (-DRSQP_STAMPS builds)

    __device__ __forceinline__ bool vV() const { return l < nV; }
    __device__ __forceinline__ bool vC() const { return l < nC; }
    __device__ __forceinline__ int nFR() const { return __popc(fmask); }
    __device__ __forceinline__ int nAC() const { return __popc(amask); }
    // entry `slot` of a vector every lane holds completely (slot: lane-varying or uniform, never a compile-time constant)
    __device__ __forceinline__ static double pick(const double (&u)[N], int slot) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < N; k++) v = fma(u[k], opaque(k == slot ? 1.0 : 0.0), v);
        return v;
    }
    // entry q (uniform over the group, not a compile-time constant) of my two rows = G_lq, G_{8+l,q}: by symmetry the entries of
    // ROW q that belong to my slots. A chain of selects on wave masks (k == q): no register-array indexing, no extra registers
    __device__ __forceinline__ void my_entries(int q, double &ev, double &ec) const {
        if constexpr (GL) { ev = Gl[l * NS + q]; ec = l < MC ? Gl[(MV + l) * NS + q] : 0.0; return; }
        ev = 0.0; ec = 0.0;
#pragma unroll
        for (int k = 0; k < N; k++) { const bool h = k == q; ev = h ? GV[k] : ev; ec = h ? GC[k] : ec; }
    }
    // G_pq for uniform slots p, q: the lane that owns slot p looks up entry q of its row, the group fetches it
    __device__ __forceinline__ double entry_pq(int p, int q) const {
        double ev, ec;
        my_entries(q, ev, ec);
        const bool pc = p >= MV;
        return fetch8(pc ? ec : ev, pc ? p - MV : p);
    }
    // the whole slot vector (variable part from the lanes' `a`, constraint part from their `b`) into every lane
    __device__ __forceinline__ void gather(double a, double b, double (&all)[N]) const {
#pragma unroll
        for (int k = 0; k < MV; k++) all[k] = fetch8(a, k);
#pragma unroll
        for (int k = 0; k < MC; k++) all[MV + k] = fetch8(b, k);
    }
    // row `src` of the variable (isc = false) or constraint part of G into every lane (= column of that slot: G is symmetric)
    __device__ __forceinline__ void fetch_row(bool isc, int src, double (&u)[N]) const {
#pragma unroll
        for (int k = 0; k < N; k++) {
            if constexpr (GL) u[k] = Gl[((isc ? MV : 0) + src) * NS + k];        // (every lane of the group reads the same row)
            else u[k] = fetch8(isc ? GC[k] : GV[k], src);
        }
    }

    // ------------------------------------------------------------------ staging
    // (patA / patH: where the row-index arrays of this problem's PATTERN start -- its own entries, or member 0's in a batch of one pattern)
    __device__ __forceinline__ void stage(const QPPools &P, const QPDesc &d, int patA, int patH) {
        const int *gAjc = P.Ajc + d.offAjc, *gAir = P.Air + patA, *gHjc = P.Hjc + d.offHjc, *gHir = P.Hir + patH;
        const double *gAval = P.Aval + d.offAnz, *gHval = P.Hval + d.offHnz;
        // every load that does not depend on another one first: the vectors and the column pointers travel together
        const bool v = vV(), c = vC();
        const int hb = (v && d.haveH) ? gHjc[l] : 0, he = (v && d.haveH) ? gHjc[l + 1] : 0, ab = v ? gAjc[l] : 0, ae = v ? gAjc[l + 1] : 0;
        const double g_ = v ? P.g[d.offV + l] : 0.0, lb_ = v ? P.lb[d.offV + l] : 0.0, ub_ = v ? P.ub[d.offV + l] : 0.0;
        const double la_ = c ? P.lbA[d.offC + l] : -RSQP_INFTY, ua_ = c ? P.ubA[d.offC + l] : RSQP_INFTY;
        for (int k = l; k < N * N; k += 8) Kd[k] = 0.0;
        TSYNC();
        // column l of H and of A (CSC), four entries per trip: their index / value loads are in flight together (a loop of one entry
        // per trip pays a full memory round trip per entry). K is symmetric, so the A entries go to both triangles
        for (int k0 = hb; k0 < he; k0 += 4) {
            int r[4]; double w[4];
#pragma unroll
            for (int t = 0; t < 4; t++) { const bool in = k0 + t < he; r[t] = in ? gHir[k0 + t] : -1; w[t] = in ? gHval[k0 + t] : 0.0; }
#pragma unroll
            for (int t = 0; t < 4; t++) if (r[t] >= 0) Kd[r[t] * N + l] = w[t];
        }
        for (int k0 = ab; k0 < ae; k0 += 4) {
            int r[4]; double w[4];
#pragma unroll
            for (int t = 0; t < 4; t++) { const bool in = k0 + t < ae; r[t] = in ? gAir[k0 + t] : -1; w[t] = in ? gAval[k0 + t] : 0.0; }
#pragma unroll
            for (int t = 0; t < 4; t++) if (r[t] >= 0) { Kd[(MV + r[t]) * N + l] = w[t]; Kd[l * N + MV + r[t]] = w[t]; }
        }
        TSYNC();
        if (v && hreg != 0.0) Kd[l * N + l] += hreg;
        TSYNC();
        hscale = max8(v ? fabs(Kd[l * N + l]) : 0.0);
        gN = g_;
        loN = v ? clampinf(lb_) : 0.0; upN = v ? clampinf(ub_) : 0.0;
        cloN = c ? clampinf(la_) : -RSQP_INFTY; cupN = c ? clampinf(ua_) : RSQP_INFTY;
    }
    __device__ __forceinline__ void g_from_K() {     // S empty: G = -K
#pragma unroll
        for (int k = 0; k < N; k++) { set_gv(k, -Kd[l * N + k]); set_gc(k, l < MC ? -Kd[(MV + (l < MC ? l : 0)) * N + k] : 0.0); }
        dV = -Kd[l * N + l]; dC = 0.0;
        if constexpr (GL) TSYNC();
    }
    __device__ __forceinline__ bool bounds_inconsistent() const {
        return or8(((vV() && loN > upN + RSQP_EPS) || (vC() && cloN > cupN + RSQP_EPS)) ? 1 : 0) != 0;
    }

    // ------------------------------------------------------------------ products with the data (LDS copy of K)
    // gyx = (A'y_C - H x) of my variable row, axx = (A x) of my constraint row, hxx = (H x) of my variable row
    __device__ __forceinline__ void exact_products(double &gyx, double &axx, double &hxx) const {
        double all[N];
        gather(xv, yc, all);
        double h = 0.0, aty = 0.0, a = 0.0;
#pragma unroll
        for (int k = 0; k < MV; k++) { h = fma(Kd[l * N + k], all[k], h); a = fma(Kd[(MV + (l < MC ? l : 0)) * N + k], all[k], a); }
#pragma unroll
        for (int k = 0; k < MC; k++) aty = fma(Kd[l * N + MV + k], all[MV + k], aty);
        gyx = aty - h; axx = a; hxx = h;
    }

    // ------------------------------------------------------------------ pivots (every lane holds the pivot column(s) u / u2)
    // principal pivot on slot q: G <- G0 - (1 / pi) u~ u~', G0 = G with row and column q zeroed, u~ = u except u~_q = sgn
    __device__ __forceinline__ void pivot1(const double (&u)[N], int q, double sgn, double pi) {
        const double c = -recip(pi);
        const bool rv = l == q, rc = MV + l == q;                                  // my row IS row q: it starts from zero
        double ev, ec;
        my_entries(q, ev, ec);                                                     // = u at my slots (G is symmetric)
        const double uv = rv ? sgn : ev, uc = rc ? sgn : ec;
        const double tv = c * uv, tc = c * uc;
        const double kv = rv ? 0.0 : 1.0, kc = rc ? 0.0 : 1.0;
#pragma unroll
        for (int k = 0; k < N; k++) {
            const bool h = k == q;
            const double ut = h ? sgn : u[k];
            set_gv(k, fma(tv, ut, gv(k) * (h ? 0.0 : kv)));
            set_gc(k, fma(tc, ut, gc(k) * (h ? 0.0 : kc)));
        }
        dV = fma(tv, uv, dV * kv); dC = fma(tc, uc, dC * kc);
        if constexpr (GL) TSYNC();
    }
    // 2 x 2 block pivot on (p, q) with W = [G_pp G_pq; G_pq G_qq]^-1: G <- G00 - U~ W U~', U~ = [u_p u_q], rows p, q = diag(sp, sq)
    __device__ __forceinline__ void pivot2(const double (&up_)[N], const double (&uq)[N], int p, double sp, int q, double sq,
                                           double w11, double w12, double w22) {
        const bool vp = l == p, vq = l == q, cp_ = MV + l == p, cq_ = MV + l == q;
        double epv, epc, eqv, eqc;
        my_entries(p, epv, epc); my_entries(q, eqv, eqc);
        const double av = vp ? sp : (vq ? 0.0 : epv), bv = vq ? sq : (vp ? 0.0 : eqv);
        const double ac = cp_ ? sp : (cq_ ? 0.0 : epc), bc = cq_ ? sq : (cp_ ? 0.0 : eqc);
        const double kv = (vp || vq) ? 0.0 : 1.0, kc = (cp_ || cq_) ? 0.0 : 1.0;
#pragma unroll
        for (int k = 0; k < N; k++) {
            const bool hp = k == p, hq = k == q;
            const double a = hp ? sp : (hq ? 0.0 : up_[k]), b = hq ? sq : (hp ? 0.0 : uq[k]), kk = (hp || hq) ? 0.0 : 1.0;
            const double cp = fma(w11, a, w12 * b), cq = fma(w12, a, w22 * b);
            set_gv(k, fma(-av, cp, fma(-bv, cq, gv(k) * (kk * kv))));
            set_gc(k, fma(-ac, cp, fma(-bc, cq, gc(k) * (kk * kc))));
        }
        if constexpr (GL) TSYNC();
        dV = fma(-av, fma(w11, av, w12 * bv), fma(-bv, fma(w12, av, w22 * bv), dV * kv));
        dC = fma(-ac, fma(w11, ac, w12 * bc), fma(-bc, fma(w12, ac, w22 * bc), dC * kc));
    }
    // sum of u_v^2 over the free variables; sum of a_v^2 over them for the row `arow` of A (>= 0), e_v (arow = -2 - v), nothing (-1)
    __device__ __forceinline__ void free_norms(const double (&u)[N], int arow, double &pn2, double &na2) const {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < MV; k++) {
            const bool fr = (fmask >> k) & 1;
            const double a = arow >= 0 ? Kd[(MV + arow) * N + k] : (k == -arow - 2 ? 1.0 : 0.0);
            s1 = fr ? fma(u[k], u[k], s1) : s1; s2 = fr ? fma(a, a, s2) : s2;
        }
        pn2 = s1; na2 = s2;
    }

    // G for a guessed working set, from G = -K: the pending free variables (mask pf) and guessed active constraints (mask pa)
    // enter S by principal pivots. The CPU restatement only asks that the FINAL reduced Hessian be positive definite and takes the
    // constraints in index order, skipping dependent ones; the order of the pivots is free as long as each keeps the inertia of a
    // KKT matrix: single pivots of the right sign first (a variable with curvature, a constraint independent of S), then 2 x 2
    // blocks (variable, constraint) with an indefinite block -- hs071's Hessian has zero diagonal entries, its free variables only
    // enter together with a constraint. Variables left over: no positive definite reduced Hessian (false). Constraints left
    // over: dependent on the others, they stay out (pa returns what was skipped).
    __device__ __forceinline__ bool warm_pivots(int pf, int &pa) {
        for (int round = 0; round < 2 * N; round++) {
            bool progress = false;
            for (int v = 0; v < MV; v++)
                if ((pf >> v) & 1) {
                    double u[N];
                    fetch_row(false, v, u);
                    const double pi = fetch8(dV, v);
                    if (-pi > 1e-8 * hscale) { pivot1(u, v, 1.0, pi); fmask |= 1 << v; pf &= ~(1 << v); progress = true; }
                }
            for (int i = 0; i < MC; i++)
                if ((pa >> i) & 1) {
                    double u[N], pn2, na2;
                    fetch_row(true, i, u);
                    const double pi = fetch8(dC, i);
                    free_norms(u, i, pn2, na2);
                    if (nFR() - nAC() > 0 && na2 > 0.0 && hscale * hscale * pn2 > 1e-18 * na2 && pi * hscale > 1e-10 * na2) {
                        pivot1(u, MV + i, 1.0, pi); amask |= 1 << i; pa &= ~(1 << i); progress = true;
                    }
                }
            if (!progress && pf != 0 && pa != 0) {
                for (int v = 0; v < MV && !progress; v++)
                    if ((pf >> v) & 1)
                        for (int i = 0; i < MC && !progress; i++)
                            if ((pa >> i) & 1) {
                                double uv[N], ui[N];
                                fetch_row(false, v, uv); fetch_row(true, i, ui);
                                const double pp = fetch8(dV, v), qq = fetch8(dC, i), pq = entry_pq(v, MV + i);
                                const double det = pp * qq - pq * pq;
                                if (det < 0.0 && -det > 1e-10 * fmax(fabs(pp * qq), pq * pq) && pq * pq > 1e-16 * hscale * hscale) {
                                    const double rd = recip(det);
                                    pivot2(uv, ui, v, 1.0, MV + i, 1.0, qq * rd, -pq * rd, pp * rd);
                                    fmask |= 1 << v; amask |= 1 << i; pf &= ~(1 << v); pa &= ~(1 << i);
                                    progress = true;
                                }
                            }
            }
            if (!progress) break;
        }
        return pf == 0;
    }

    // ------------------------------------------------------------------ auxiliary QP (setup_aux of the CPU restatement)
    // cold: x = 0, y = 0, every variable on a finite bound (lower first). Warm (x0 / y0 / guessed bounds gb / guessed
    // constraints gc per lane, any of them absent): the working set of the guess, G built for it by pivots from -K.
    // false: the guess does not give a positive definite reduced Hessian (the caller falls back to a cold start)
    __device__ __forceinline__ bool setup(bool have_x0, bool have_y0, bool have_gb, bool have_gc, bool c_from_y0, double x0, double y0v,
                                          double y0c, int gb, int gc) {
        status = QPS_PREPARINGAUXILIARYQP;
        infeasible = unbounded = 0;
        xv = (have_x0 && vV()) ? x0 : 0.0;
        yv = (have_y0 && vV()) ? y0v : 0.0; yc = (have_y0 && vC()) ? y0c : 0.0;
        int s;
        if (have_gb) s = gb;
        else if (have_x0) s = xv <= loN + RSQP_BOUND_TOLERANCE ? -1 : (xv >= upN - RSQP_BOUND_TOLERANCE ? 1 : 0);
        else if (have_y0) s = yv > RSQP_EPS ? -1 : (yv < -RSQP_EPS ? 1 : 0);
        else s = -1;
        if (s == -1 && loN <= -RSQP_INFTY) s = (upN < RSQP_INFTY && !have_x0 && !have_gb) ? 1 : 0;
        if (s == 1 && upN >= RSQP_INFTY) s = 0;
        sv = vV() ? s : -1;
        g_from_K();
        fmask = amask = 0;
        // A x of the guess and the constraints' sides (qpOASES: from the guess, else from y0, else from A x0)
        const bool cold = !have_x0 && !have_y0;          // x = 0, y = 0: every product with the data is zero
        double gyx = 0.0, axx = 0.0, hxx = 0.0;
        if (!cold) exact_products(gyx, axx, hxx);
        ax = vC() ? axx : 0.0;
        int sgc = 0;
        if (have_gc) sgc = gc;
        else if (have_y0 && (!have_x0 || c_from_y0)) sgc = yc > RSQP_EPS ? -1 : (yc < -RSQP_EPS ? 1 : 0);
        else if (have_x0) sgc = ax <= cloN + RSQP_BOUND_TOLERANCE ? -1 : (ax >= cupN - RSQP_BOUND_TOLERANCE ? 1 : 0);
        if (sgc == -1 && cloN <= -RSQP_INFTY) sgc = 0;
        if (sgc == 1 && cupN >= RSQP_INFTY) sgc = 0;
        if (!vC()) sgc = 0;
        int pa = or8(sgc != 0 ? 1 << l : 0);
        const int pf = or8((vV() && sv == 0) ? 1 << l : 0);
        if (!warm_pivots(pf, pa)) return false;
        sc = ((pa >> l) & 1) ? 0 : sgc;          // (what is left in pa was dependent: it stays out)
        // multipliers: zero when inactive, clipped to the sign their side requires
        yv = sv == 0 ? 0.0 : ((sv == -1 && yv < 0.0) || (sv == 1 && yv > 0.0) ? 0.0 : yv);
        yc = sc == 0 ? 0.0 : ((sc == -1 && yc < 0.0) || (sc == 1 && yc > 0.0) ? 0.0 : yc);
        // gradient of the auxiliary QP from stationarity, its limits around the iterate
        if (!cold) exact_products(gyx, axx, hxx);
        gy = gyx; g = gyx + yv;
        lo = sv == -1 ? xv : fmin(loN, xv - RSQP_BOUND_RELAXATION);
        up = sv == 1 ? xv : fmax(upN, xv + RSQP_BOUND_RELAXATION);
        loA = sc == -1 ? ax : fmin(cloN, ax - RSQP_BOUND_RELAXATION);
        upA = sc == 1 ? ax : fmax(cupN, ax + RSQP_BOUND_RELAXATION);
        if (!vV()) { lo = up = 0.0; }
        status = QPS_AUXILIARYQPSOLVED;
        return true;
    }

    // ------------------------------------------------------------------ one working-set change
    // kind 1 constraint idx leaves | 2 bound of idx leaves | 3 constraint idx enters at `side` | 4 variable idx gets fixed at `side`
    __device__ __forceinline__ int change(int kind, int idx, int side, double tau, bool &treat_done) {
        const bool isc = kind == 1 || kind == 3;
        const int q = isc ? MV + idx : idx;
        double u[N];
        fetch_row(isc, idx, u);
        const double pi = fetch8(isc ? dC : dV, idx);
        const bool myV = !isc && l == idx, myC = isc && l == idx;
        bool flip = false;
        if (kind == 1) {
            double d2, dummy;
            free_norms(u, -1, d2, dummy);
            flip = !(d2 > 0.0 && -pi > 1e-8 * hscale * d2);
        } else if (kind == 2) flip = !(-pi > 1e-8 * hscale);
        if (flip) {
            // the released direction has no curvature: the constraint / bound goes to its OPPOSITE side, G is unchanged
            const int cant = fetch8i(isc ? ((sc == -1 ? cupN : cloN) >= RSQP_INFTY || (sc == -1 ? cupN : cloN) <= -RSQP_INFTY ? 1 : 0)
                                         : ((sv == -1 ? upN : loN) >= RSQP_INFTY || (sv == -1 ? upN : loN) <= -RSQP_INFTY ? 1 : 0), idx);
            if (cant) return RET_UNBOUNDED;
            upA = (myC && sc == -1) ? ax : upA; loA = (myC && sc == 1) ? ax : loA; yc = myC ? 0.0 : yc; sc = myC ? -sc : sc;
            up = (myV && sv == -1) ? xv : up; lo = (myV && sv == 1) ? xv : lo; yv = myV ? 0.0 : yv; sv = myV ? -sv : sv;
            nflips++;
            since_refresh = REFRESH;
            return RET_OK;
        }
        int pk = 0, pidx = -1;
        double ynew = 0.0;
        if (kind >= 3) {
            const double sg = kind == 3 ? -1.0 : 1.0;
            double pn2, na2;
            free_norms(u, kind == 3 ? idx : -2 - idx, pn2, na2);
            int li;
            if (nFR() - nAC() <= 0 || !(na2 > 0.0)) li = 0;
            else {
                // rel = hscale |P a| / |a_FR| against 1e-6 / 1e-12, squared: no square root, no division
                const double p2 = hscale * hscale * pn2;
                li = p2 > 1e-12 * na2 ? 1 : (p2 < 1e-24 * na2 ? 0 : -1);
            }
            if (li < 0) {
                // the band: the residual of the row's representation by the active rows decides (qp_small_g.h)
                double r = vV() ? (kind == 3 ? Kd[(MV + idx) * N + l] : (l == idx ? 1.0 : 0.0)) : 0.0;
#pragma unroll
                for (int i = 0; i < MC; i++) r = ((amask >> i) & 1) ? fma(-Kd[(MV + i) * N + l], sg * u[MV + i], r) : r;
                const double rn2 = sum8((vV() && sv == 0) ? r * r : 0.0);
                li = rn2 > 9e-16 * na2 ? 1 : 0;                 // |r| / |a_FR| > 3e-8
            }
            if (li == 0) {
                // ---- exchange: shift the multipliers along the dependency until one of them reaches zero; that one leaves
                const double sgn = side == 1 ? -1.0 : 1.0;
                double ev, ec;
                my_entries(q, ev, ec);
                const double xiv = (vV() && sv != 0) ? sgn * sg * ev : 0.0, xic = (vC() && sc != 0) ? sgn * sg * ec : 0.0;
                double bt = RSQP_INFTY;
                int bid = 0x7fffffff;
                if (vC() && sc != 0) {
                    const double num = sc == -1 ? yc : -yc, den = sc == -1 ? xic : -xic;
                    if (den > RSQP_EPS_DEN) { bt = (num > 0.0 ? num : 0.0) / den; bid = l; }
                }
                if (vV() && sv != 0) {
                    const double num = sv == -1 ? yv : -yv, den = sv == -1 ? xiv : -xiv;
                    if (den > RSQP_EPS_DEN) { const double t = (num > 0.0 ? num : 0.0) / den; if (t < bt) { bt = t; bid = nC + l; } }
                }
                argmin8(bt, bid);
                if (bid == 0x7fffffff) {
                    // no partner: the QP is infeasible beyond this point of the homotopy -- unless that point IS its end to rounding
                    // (the blocking row was met at tau = 1 - O(eps): a degenerate vertex), then the solve is complete
                    if (tau >= 1.0 - 1e-9) { treat_done = true; return RET_OK; }
                    return RET_INFEASIBLE;
                }
                if (bid < nC) { pk = 1; pidx = bid; } else { pk = 2; pidx = bid - nC; }
                yv -= bt * xiv; yc -= bt * xic;
                ynew = sgn * bt;
                const int p = pk == 1 ? MV + pidx : pidx;
                double u2[N];
                fetch_row(pk == 1, pidx, u2);
                const double pp = fetch8(pk == 1 ? dC : dV, pidx), qq = pi, pq = entry_pq(p, q);
                const double det = pp * qq - pq * pq;
                if (!(det < 0.0) || !(-det > 1e-10 * fmax(fabs(pp * qq), pq * pq))) return RET_SETUP_FAILED;
                const double rd = recip(det);
                pivot2(u2, u, p, pk == 1 ? -1.0 : 1.0, q, kind == 3 ? 1.0 : -1.0, qq * rd, -pq * rd, pp * rd);
                since_refresh = REFRESH;
            } else {
                if (kind == 3) { if (!(pi * hscale > 1e-10 * na2)) return RET_SETUP_FAILED; }
                else if (!(pi * hscale > 1e-10)) return RET_SETUP_FAILED;
            }
        }
        if (pk == 0) pivot1(u, q, (kind == 2 || kind == 3) ? 1.0 : -1.0, pi);
        // ---- the working set
        const bool leaves = kind == 1 || kind == 2;
        const bool pV = pk == 2 && l == pidx, pC = pk == 1 && l == pidx;
        sv = myV ? (leaves ? 0 : side) : (pV ? 0 : sv);
        yv = myV ? (leaves ? 0.0 : ynew) : (pV ? 0.0 : yv);
        sc = myC ? (leaves ? 0 : side) : (pC ? 0 : sc);
        yc = myC ? (leaves ? 0.0 : ynew) : (pC ? 0.0 : yc);
        if (kind == 1) amask &= ~(1 << idx); else if (kind == 2) fmask |= 1 << idx; else if (kind == 3) amask |= 1 << idx; else fmask &= ~(1 << idx);
        if (pk == 1) amask &= ~(1 << pidx); else if (pk == 2) fmask |= 1 << pidx;
        return RET_OK;
    }

    __device__ __forceinline__ int homotopy(int maxit, int &nWSR, bool hot) {
        int iter = 0, rcode = RET_OK;
        status = QPS_PERFORMINGHOMOTOPY;
        since_refresh = REFRESH;
        // (as the homotopy of the other engines begins) an inactive side that was infinite and now has a finite target only has
        // to stay clear of the iterate
        if (sv != -1 && lo <= -RSQP_INFTY && loN > -RSQP_INFTY) lo = fmin(loN, xv - RSQP_BOUND_RELAXATION);
        if (sv != 1 && up >= RSQP_INFTY && upN < RSQP_INFTY) up = fmax(upN, xv + RSQP_BOUND_RELAXATION);
        if (sc != -1 && loA <= -RSQP_INFTY && cloN > -RSQP_INFTY) loA = fmin(cloN, ax - RSQP_BOUND_RELAXATION);
        if (sc != 1 && upA >= RSQP_INFTY && cupN < RSQP_INFTY) upA = fmax(cupN, ax + RSQP_BOUND_RELAXATION);
        for (;;) {
            // ---- x exactly on its active bounds; (exact products); drift correction + input of the product
            if (sv != 0) xv = sv == -1 ? lo : up;
            const bool keep = hot && iter == 0;
            if (since_refresh >= REFRESH) {
                if (!keep) { double gyx, axx, hxx; exact_products(gyx, axx, hxx); gy = gyx; ax = vC() ? axx : 0.0; }
                since_refresh = 0;
            }
            {
                const double gv = keep ? g : gy + yv;
                if (keep) gy = gv - yv;
                g = gv;
                inV = vV() ? (sv == 0 ? -(gN - gv) : (sv == -1 ? loN - lo : upN - up)) : 0.0;
                loA = (!keep && sc == -1) ? ax : loA; upA = (!keep && sc == 1) ? ax : upA;
                inC = sc == 0 ? 0.0 : (sc == -1 ? cloN - loA : cupN - upA);
            }
            TSTAMP(5);
            // ---- out = G in; dx / dy / A dx and the ratio-test candidates of my rows
            double dxv, dyv, hd, dax, dyc;
            double bt = 1.0;
            int bid = 0x7fffffff;
            {
                double all[N];
                gather(inV, inC, all);
                double ov = 0.0, oc = 0.0;
#pragma unroll
                for (int k = 0; k < N; k++) { ov = fma(gv(k), all[k], ov); oc = fma(gc(k), all[k], oc); }
                const double dg = gN - g;
                dxv = sv == 0 ? ov : inV; dyv = sv == 0 ? 0.0 : dg - ov; hd = sv == 0 ? -dg : -ov;
                dax = sc != 0 ? inC : -oc; dyc = sc != 0 ? -oc : 0.0;
            }
            TSTAMP(6);
            {
                // candidates in the order of their ids (ties go to the lowest id)
                auto cand = [&](double num, double den, int id, bool ok) {
                    if (ok && den >= RSQP_EPS_DEN) {
                        const double t = (num > 0.0 ? num : 0.0) / den;
                        if (t < bt || (t == bt && id < bid)) { bt = t; bid = id; }
                    }
                };
                if (vC()) {
                    cand(sc != 0 ? (sc == -1 ? yc : -yc) : ax - loA, sc != 0 ? (sc == -1 ? -dyc : dyc) : (cloN - loA) - dax,
                         sc != 0 ? l : nC + nV + l, sc != 0 || cloN > -RSQP_INFTY);
                    cand(upA - ax, dax - (cupN - upA), 2 * nC + nV + l, sc == 0 && cupN < RSQP_INFTY);
                }
                if (vV()) {
                    cand(sv != 0 ? (sv == -1 ? yv : -yv) : xv - lo, sv != 0 ? (sv == -1 ? -dyv : dyv) : (loN - lo) - dxv,
                         sv != 0 ? nC + l : 3 * nC + nV + l, sv != 0 || loN > -RSQP_INFTY);
                    cand(up - xv, dxv - (upN - up), 3 * nC + 2 * nV + l, sv == 0 && upN < RSQP_INFTY);
                }
            }
            if (!(bt < 1.0)) { bt = 1.0; bid = 0x7fffffff; }
            TSTAMP(7);
            argmin8(bt, bid);
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
            bool done = kind == 0;
            const bool cap = iter >= maxit;
            // ---- homotopy step on my rows
            yv += tau * dyv; yc += tau * dyc;
            {
                const double xn = xv + tau * dxv, an = ax + tau * dax;
                const double l1 = lo + tau * (loN - lo), u1 = up + tau * (upN - up), l2 = loA + tau * (cloN - loA), u2 = upA + tau * (cupN - upA);
                const bool hitV = !done && !cap && kind == 4 && l == idx, hitC = !done && !cap && kind == 3 && l == idx;
                xv = done ? (sv == -1 ? loN : (sv == 1 ? upN : xn)) : xn;
                ax = done ? ax : an;
                g = done ? gN : g + tau * (gN - g);
                gy -= tau * hd;
                lo = done ? loN : ((hitV && side == -1) ? xn : l1); up = done ? upN : ((hitV && side == 1) ? xn : u1);
                loA = done ? cloN : ((hitC && side == -1) ? an : l2); upA = done ? cupN : ((hitC && side == 1) ? an : u2);
            }
            TSTAMP(8);
            if (done || cap) {
                if (done) status = QPS_SOLVED; else rcode = RET_MAX_NWSR;
                break;
            }
            bool treat_done = false;
            rcode = change(kind, idx, side, tau, treat_done);
            TSTAMP(9);
            if (treat_done) {      // (see change: no exchange partner at the very end of the homotopy)
                g = gN; lo = loN; up = upN; loA = cloN; upA = cupN;
                if (sv != 0) xv = sv == -1 ? loN : upN;
                status = QPS_SOLVED;
                break;
            }
            if (rcode == RET_INFEASIBLE) { infeasible = 1; break; }
            if (rcode == RET_UNBOUNDED) { unbounded = 1; break; }
            if (rcode != RET_OK) break;
            iter++;
            since_refresh++;
        }
        nWSR = iter;
        return rcode;
    }

    // solved: ONE step of iterative refinement on the final KKT system with residuals from the data, multipliers of the fixed
    // variables from stationarity, A x of the final iterate; returns the objective 0.5 x'Hx + gN'x (refine = false: A x, objective)
    __device__ __forceinline__ double finish(bool refine) {
        double gyx, axx, hxx;
        if (refine) {
            exact_products(gyx, axx, hxx);
            inV = (vV() && sv == 0) ? -(gN - gyx) : 0.0;
            inC = sc != 0 ? (sc == 1 ? cupN : cloN) - axx : 0.0;
            double all[N];
            gather(inV, inC, all);
            double ov = 0.0, oc = 0.0;
#pragma unroll
            for (int k = 0; k < N; k++) { ov = fma(gv(k), all[k], ov); oc = fma(gc(k), all[k], oc); }
            if (vV() && sv == 0) xv += ov;
            if (sc != 0) yc -= oc;
        }
        exact_products(gyx, axx, hxx);
        if (refine && vV()) yv = sv != 0 ? gN - gyx : 0.0;
        ax = vC() ? axx : 0.0;
        return sum8(vV() ? xv * fma(0.5, hxx - hreg * xv, gN) : 0.0);      // (the objective excludes the LP regularisation)
    }
};

// persistent state of one problem (hot starts), in the problem's state block, every piece LANE-CONTIGUOUS (the 8 lanes of a problem
// store 8 consecutive doubles per instruction; the first layout -- a row of the tableau and the fields of a slot consecutive PER LANE --
// made every store of a wave 64 separate 8-byte pieces): [N x N tableau, entry k of every slot's row together: position k N + slot; the
// matrix is symmetric, so this is its transpose up to rounding][6 fields x 8 variable slots: x g lo up gy y][4 fields x 8 constraint
// slots: A x, loA, upA, y][G_ll of the 8 variable slots][G_{8+l,8+l} of the 8 constraint slots][ints: sv (8), sc (8), status, masks, magic,
// pivots]
template <int MC> __device__ __forceinline__ long long tiny_state_doubles() { return (long long)(MV + MC) * (MV + MC) + 8LL * 6 + 8LL * 4 + 8LL * 2; }
constexpr int TINY_MAGIC = 0x7a11e;

template <int MC, int W, bool GL = false>
__global__ void __launch_bounds__(TB, W) tiny_qp_kernel(QPPools P, int nq, int mode_in, int maxWSR) {
    typedef EngineT<MC, GL> ENG;
    constexpr int N = ENG::N;
    __shared__ __attribute__((aligned(16))) double kd_all[TG * N * N];
    __shared__ __attribute__((aligned(16))) double gl_all[GL ? TG * N * ENG::NS : 1];
    const int grp = (int)threadIdx.x >> 3;
    const int q = (int)blockIdx.x * TG + grp;
    if (q >= nq) return;        // (no workgroup barrier anywhere: idle groups may leave)
    // a batch of ONE shape and ONE sparsity pattern (QPPools::uni_pat): offsets by arithmetic, pattern arrays of member 0
    QPDesc d;
    int patA, patH;
    if (P.uni_pat) {
        d.nV = P.uniV; d.nC = P.uniC; d.offV = q * P.uniV; d.offC = q * P.uniC; d.offAjc = 0; d.offHjc = 0; d.offArp = 0;
        d.offAnz = q * P.uni_annz; d.offHnz = q * P.uni_hnnz; d.haveH = P.uni_haveH; d.annz = P.uni_annz; d.hnnz = P.uni_hnnz;
        d.hreg = P.uni_hreg; d.offState = (long long)q * P.uni_state;
        patA = patH = 0;
    } else { d = P.desc[q]; patA = d.offAnz; patH = d.offHnz; }
    ENG E;
    E.l = (int)threadIdx.x & 7; E.nV = d.nV; E.nC = d.nC; E.hreg = d.hreg;
    E.Kd = (ldouble *)kd_all + grp * N * N;
    E.Gl = (ldouble *)gl_all + (GL ? grp * N * ENG::NS : 0);
    E.nflips = 0; E.infeasible = E.unbounded = 0; E.status = QPS_NOTINITIALISED; E.fmask = E.amask = 0; E.since_refresh = 0;
    const int l = E.l;
#ifdef RSQP_STAMPS
    E.tlast = clock64();
    long long &tlast = E.tlast;
#endif
    E.stage(P, d, patA, patH);
    TSTAMP(0);
    int mode = mode_in;
    double *sd = P.state + d.offState;
    int *si = reinterpret_cast<int *>(sd + tiny_state_doubles<MC>());
    // stored state of the previous solve (hot starts; x / y / working set as the guess of a hot start with new matrices)
    double px = 0.0, pyv = 0.0, pyc = 0.0; int psv = -1, psc = 0;
    if (mode == 1 || mode == 2) {
        if (si[18] != TINY_MAGIC || si[16] == QPS_NOTINITIALISED) mode = 0;
        else {
            double *pr = sd + N * N;
            px = pr[0 * 8 + l]; pyv = pr[5 * 8 + l]; pyc = pr[48 + 3 * 8 + l]; psv = si[l]; psc = si[8 + l];
            if (mode == 1) {
#pragma unroll
                for (int k = 0; k < N; k++) { E.set_gv(k, sd[k * N + l]); E.set_gc(k, l < MC ? sd[k * N + MV + (l < MC ? l : 0)] : 0.0); }
                if constexpr (GL) TSYNC();
                E.xv = px; E.g = pr[1 * 8 + l]; E.lo = pr[2 * 8 + l]; E.up = pr[3 * 8 + l]; E.gy = pr[4 * 8 + l]; E.yv = pyv;
                E.ax = pr[48 + 0 * 8 + l]; E.loA = pr[48 + 1 * 8 + l]; E.upA = pr[48 + 2 * 8 + l]; E.yc = pyc;
                E.sv = psv; E.sc = psc; E.status = si[16]; E.fmask = si[17] & 0xff; E.amask = (si[17] >> 8) & 0xff;
                E.dV = pr[80 + l]; E.dC = pr[88 + l];
            }
        }
    }
    int rcode = RET_OK, nWSR = 0, setup_pivots = 0;
    if (E.bounds_inconsistent()) {
        // qpOASES areBoundsConsistent: infeasible before any change (a hot start keeps the stored iterate)
        E.infeasible = 1;
        rcode = RET_INFEASIBLE;
        if (mode != 1) { E.xv = px; E.yv = pyv; E.yc = pyc; E.sv = psv; E.sc = psc; E.ax = 0.0; E.g = E.gy = 0.0; E.lo = E.up = E.loA = E.upA = 0.0; E.g_from_K(); }
    } else {
        // ONE call site of the set-up (inlined three times -- cold, new matrices, warm re-init -- it tripled the kernel): the guess of
        // the call shape first, the cold start as the fallback when the guess has no positive definite reduced Hessian
        bool ok = mode == 1;
        bool hx = false, hy = false, hg = false, hc = false;
        double x0 = 0.0, y0v = 0.0, y0c = 0.0;
        int gb = 0, gc = 0;
        if (mode == 2) { hx = hy = hg = hc = true; x0 = px; y0v = pyv; y0c = pyc; gb = psv; gc = psc; }
        else if (mode == 3) {
            hx = P.x0 != nullptr; hy = P.y0 != nullptr; hg = P.guess_b != nullptr;
            x0 = (hx && l < d.nV) ? P.x0[d.offV + l] : 0.0;
            y0v = (hy && l < d.nV) ? P.y0[d.offV + d.offC + l] : 0.0; y0c = (hy && l < d.nC) ? P.y0[d.offV + d.offC + d.nV + l] : 0.0;
            gb = (hg && l < d.nV) ? P.guess_b[d.offV + l] : 0;
        }
        for (int attempt = 0; attempt < 2 && !ok; attempt++) {
            ok = E.setup(hx, hy, hg, hc, P.reinit_from_y0 != 0, x0, y0v, y0c, gb, gc);
            hx = hy = hg = hc = false;
        }
        TSTAMP(1);
        if (mode != 1) setup_pivots = E.nFR() + E.nAC();
        if (!ok) rcode = RET_SETUP_FAILED;
        else rcode = E.homotopy(maxWSR, nWSR, mode == 1);
        TSTAMP(2);
    }
    // the refinement step repairs what the update-only tableau accumulates: with at most 4 pivots since G was built from the data
    // (the two changes of the headline hs071 QP) there is nothing to repair yet -- |dx| ~ 1e-15 either way
    const int pivots = (mode == 1 ? si[19] : setup_pivots) + nWSR;
    const double obj = E.finish(rcode == RET_OK && pivots > 4);
    TSTAMP(3);
    // ---- results (x, y = [bounds; constraints], working set, status / nWSR / objective)
    if (l < d.nV) { P.x[d.offV + l] = E.xv; P.ws_b[d.offV + l] = E.sv; P.y[d.offV + d.offC + l] = E.yv; }
    if (l < d.nC) { P.y[d.offV + d.offC + d.nV + l] = E.yc; P.ws_c[d.offC + l] = E.sc; }
    if (l == 0) {
        const int st = E.status;
        P.status[q] = E.infeasible ? 100 + st : (E.unbounded ? 200 + st : st);
        P.ret[q] = rcode; P.nwsr[q] = nWSR; P.nflips[q] = E.nflips; P.obj[q] = obj;
    }
    if (P.keep_state) {
        // (a hot start on new vectors that changed nothing in the working set found the tableau in the block and leaves it there:
        //  half of the block's bytes -- the late iterations of an SQP run)
        if (!(mode == 1 && nWSR == 0)) {
#pragma unroll
            for (int k = 0; k < N; k++) { sd[k * N + l] = E.gv(k); if (l < MC) sd[k * N + MV + l] = E.gc(k); }
        }
        double *pr = sd + N * N;
        pr[0 * 8 + l] = E.xv; pr[1 * 8 + l] = E.g; pr[2 * 8 + l] = E.lo; pr[3 * 8 + l] = E.up; pr[4 * 8 + l] = E.gy; pr[5 * 8 + l] = E.yv;
        pr[48 + 0 * 8 + l] = E.ax; pr[48 + 1 * 8 + l] = E.loA; pr[48 + 2 * 8 + l] = E.upA; pr[48 + 3 * 8 + l] = E.yc;
        pr[80 + l] = E.dV; pr[88 + l] = E.dC;
        si[l] = E.sv; si[8 + l] = E.sc;
        if (l == 0) { si[16] = E.status; si[17] = (E.fmask & 0xff) | ((E.amask & 0xff) << 8); si[18] = TINY_MAGIC; si[19] = pivots; }
    } else if (l == 0 && !P.skip_mark) { si[16] = QPS_NOTINITIALISED; si[18] = TINY_MAGIC; }
    TSTAMP(4);
    if (P.cert_out) {
        // the reference's certificate (qpOASESInterface::test_optimality, src/qpOASESInterface.cpp:498-684) on the answer just
        // written, from the rows this lane holds: A x, A'y_C - H x of the final iterate are exact products of finish()
        const double lraw = l < d.nV ? P.lb[d.offV + l] : 0.0, uraw = l < d.nV ? P.ub[d.offV + l] : 0.0;
        const double laraw = l < d.nC ? P.lbA[d.offC + l] : 0.0, uaraw = l < d.nC ? P.ubA[d.offC + l] : 0.0;
        double gyx, axx, hxx;
        E.exact_products(gyx, axx, hxx);
        double primal = 0.0, dual = 0.0, compl_ = 0.0, stat = 0.0;
        int bad = 0;
        if (l < d.nV) {
            const double lo_ = fmax(lraw, -RSQP_K_INFTY), up_ = fmin(uraw, RSQP_K_INFTY);
            const int Wm = map_bound(E.sv, E.xv, lo_, up_);
            P.cert_Wb[d.offV + l] = Wm;
            primal += fmax(0.0, lo_ - E.xv) + -fmin(0.0, up_ - E.xv);
            kkt_terms(Wm, E.yv, E.xv, lo_, up_, dual, compl_, bad);
            stat += fabs(gyx + E.yv - E.gN);                         // A'y_C + y_B - g - H x
        }
        if (l < d.nC) {
            const double lo_ = fmax(laraw, -RSQP_K_INFTY), up_ = fmin(uaraw, RSQP_K_INFTY);
            const int Wm = map_constr(E.sc, axx, lo_, up_);
            P.cert_Wc[d.offC + l] = Wm;
            primal += fmax(0.0, lo_ - axx) + -fmin(0.0, up_ - axx);
            kkt_terms(Wm, E.yc, axx, lo_, up_, dual, compl_, bad);
        }
        primal = sum8(primal); dual = sum8(dual); compl_ = sum8(compl_); stat = sum8(stat);
        bad = or8(bad);
        if (l == 0) {
            double *o = P.cert_out + 6LL * q;
            o[0] = primal; o[1] = dual; o[2] = compl_; o[3] = stat; o[4] = compl_ + stat + dual + primal; o[5] = (double)bad;
        }
    }
    if (P.done_flag) {
        __threadfence_system();      // the results above are in host-mapped memory: visible before the flag
        if (q == 0 && l == 0) *reinterpret_cast<volatile int *>(P.done_flag) = P.done_val;
    }
}

}  // namespace

// bytes of the state block the kernel needs (the caller's blocks are sized by rsqp_state_bytes: checked by the launcher)
long long rsqp_tiny_state_bytes(int nCmax) {
    const long long N = MV + (nCmax <= 2 ? 2 : (nCmax <= 4 ? 4 : 8));
    return 8 * (N * N + 48 + 32 + 16) + 4 * 24;
}
// 1 if the batch shape is served by this engine
int rsqp_tiny_fits(const SmallKnobs &kn, int nVmax, int nCmax) {
    return !kn.no_tiny && nVmax <= MV && nCmax <= 8 && nVmax >= 1;
}
hipError_t rsqp_launch_tiny_qp(const SmallKnobs &kn, const QPPools &p, int nq, int nVmax, int nCmax, int mode, int maxWSR, hipStream_t stream) {
    if (nq <= 0) return hipSuccess;
    if (!rsqp_tiny_fits(kn, nVmax, nCmax)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((nq + TG - 1) / TG)), block(TB);
    // RSQP_TINY_LDS=1: the tableau in LDS, three waves per SIMD (tuning: default decided by measurement)
    const int glds = kn.tiny_lds;
    if (nCmax <= 2 && glds) hipLaunchKernelGGL((tiny_qp_kernel<2, 3, true>), grid, block, 0, stream, p, nq, mode, maxWSR);
    // (launches of at most one workgroup per CU -- the single QP of an SQP iteration above all -- get the builds for ONE wave per SIMD:
    //  268 instead of 256 registers, none spilled to scratch, whose round trips sit in the chain of a lone wave: cold solve of the hs071 QP 30.5 ->
    //  28.8 us through the Python loop, the solveQP replay 23.0 -> 22.6 us through the C++ boundary, batches of up to 8 192 QPs 3 % faster)
    else if (nCmax <= 2 && nq <= 32 * 256) hipLaunchKernelGGL((tiny_qp_kernel<2, 1>), grid, block, 0, stream, p, nq, mode, maxWSR);
    else if (nCmax <= 2) hipLaunchKernelGGL((tiny_qp_kernel<2, 2>), grid, block, 0, stream, p, nq, mode, maxWSR);
    else if (nCmax <= 4 && nq <= 32 * 256) hipLaunchKernelGGL((tiny_qp_kernel<4, 1>), grid, block, 0, stream, p, nq, mode, maxWSR);
    else if (nCmax <= 4) hipLaunchKernelGGL((tiny_qp_kernel<4, 2>), grid, block, 0, stream, p, nq, mode, maxWSR);
    else hipLaunchKernelGGL((tiny_qp_kernel<8, 1>), grid, block, 0, stream, p, nq, mode, maxWSR);
    return hipGetLastError();
}
