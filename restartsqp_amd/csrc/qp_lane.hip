// qp_lane.hip -- hs071-scale QPs of a ONE-SHAPE batch, cold start, ONE LANE PER PROBLEM (round 5).
//
// Replaces, for large batches of one shape (one sparsity pattern -- parameter scans, perturbations of one QP: bench.py's
// headline workload, 65 536 x the hs071 QP through the QPhandler formulation -- or patterns of their own), the qpOASES 3.2.1 SQProblem::init call made
// at reference src/qpOASESInterface.cpp:155,180. Same algorithm as qp_tiny.hip -- the symmetric tableau
//      G = - SWEEP_S(K),  K = [H A'; A 0],  S = free variables + active constraints,  N = 8 + MC fixed slots
// with one product per step direction, one principal pivot per working-set change (2 x 2 block for an exchange), exact
// products with the data after an exchange / every 8 changes and at the end, the same ratio tests, tie breaks and tolerances --
// but mapped the other way round: qp_tiny.hip gives a problem 8 lanes and keeps the tableau in their registers, so every
// pivot row, product input and argmin crosses lanes (ds_bpermute / DPP) and 3 of 4 constraint rows are idle at MC = 2; a wave
// (8 problems) takes ~45 k cycles, two waves per SIMD: 16 384 problems in flight, 4 rounds for 65 536.
// Here a LANE owns a problem:
//   * the tableau (upper triangle, N (N + 1) / 2 doubles) and the dense A of the data live in LDS (H's upper triangle, only ever
//     read at compile-time positions, in registers), entry e of
//     lane i at (e * 64 + i) * 8: a wave's access to one entry is 512 consecutive bytes -- conflict-free for static AND for
//     lane-varying entries (the pivot row of each lane's own problem), no index arithmetic for static ones (immediate offsets);
//   * every vector of the solver state (x, limits, targets, multipliers, gradients: 9 per variable, 6 per constraint) is a
//     register array indexed by compile-time slots; "slot idx of my problem" is a predicated pass over the slots;
//   * nothing crosses lanes: no permutes, no barriers, no reductions -- a wave is 64 independent scalar programs in lockstep
//     (identical paths for perturbations of one QP; diverging members cost the union of their paths, as in any SIMT code).
// One wave per workgroup, LDS 36.4 KB (MC = 2): four waves per CU, one per SIMD with the whole register file (512 per lane) --
// 65 536 problems in flight on the 256 CUs.
#include <type_traits>

#include "rsqp_internal.h"

#define LDS __attribute__((address_space(3)))
typedef LDS double ldouble;

// diagnostic build only (-DRSQP_STAMPS, tools/stamp_lane_kernel.py): cycles per phase of wave 0
#ifdef RSQP_STAMPS
__device__ unsigned long long g_lane_stamps[16];
#define LSTAMP(k)                                                                                        \
    do {                                                                                                 \
        long long t_ = clock64();                                                                        \
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_lane_stamps[k], (unsigned long long)(t_ - tlast)); \
        tlast = t_;                                                                                      \
    } while (0)
extern "C" void rsqp_debug_lane_stamps(unsigned long long *out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lane_stamps), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lane_stamps), z, sizeof(z));
    }
}
#else
// the phase boundaries stay fences for the instruction scheduler in the product build: without them it hoists the loads and
// row fetches of a phase far up into the one before, and the 234 registers of persistent state (of the 256 the vector ALU can
// name) leave no room for that -- 0.0566 ms per launch of the headline batch without, 0.0433 ms with them (the build with the
// time stamps had been the faster one). Finer fences (per slot, inside the pivots) measured no better (0.0442 ms)
#define LSTAMP(k) __builtin_amdgcn_sched_barrier(0)
#endif

namespace {

constexpr int MV = 8, WL = 64;
// loops over slots with COMPILE-TIME indices: the register arrays of the solver state are split into scalars before any other
// transformation sees them (a run-time loop, even one that is fully unrolled later, leaves selects between member addresses
// behind, and the whole state goes to scratch memory)
template <int I, int E, class F> __device__ __forceinline__ void sfor_(F &&f) {
    if constexpr (I < E) { f(std::integral_constant<int, I>{}); sfor_<I + 1, E>(f); }
}
#define SFOR(var, n, ...) sfor_<0, (n)>([&](auto var##_c) { constexpr int var = decltype(var##_c)::value; (void)var; __VA_ARGS__ })
#define SFOR1(var, n, ...) sfor_<1, (n) + 1>([&](auto var##_c) { constexpr int var = decltype(var##_c)::value; (void)var; __VA_ARGS__ })

// a status word the compiler must not look through: the slot passes of homotopy() each start from a fresh copy, so the lane masks
// derived from it (free / at lower / at upper) are re-formed per pass (one compare) instead of kept in scalar registers across the
// whole iteration, from where they were spilled to lanes of vector registers and read back (2-3 % of the kernel)
__device__ __forceinline__ int opq(int v) { asm volatile("" : "+v"(v)); return v; }
__host__ __device__ constexpr int tri(int j, int k) { return j <= k ? k * (k + 1) / 2 + j : j * (j + 1) / 2 + k; }

__device__ __forceinline__ double clampinf(double v) { return v > RSQP_INFTY ? RSQP_INFTY : (v < -RSQP_INFTY ? -RSQP_INFTY : v); }
__device__ __forceinline__ double recip(double x) {      // v_rcp_f64 + two Newton steps (as qp_tiny.hip)
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0); y = fma(y, e, y);
    e = fma(-x, y, 1.0); y = fma(y, e, y);
    return y;
}

template <int MC>
struct LaneT {
    static constexpr int N = MV + MC, NT = N * (N + 1) / 2, NH = MV * (MV + 1) / 2, NA = MC * MV, REFRESH = 8;
    ldouble *G, *K;                      // my problem's tableau / the dense A of the data: entry e at [e * WL]
    double Hr[NH];                       // the upper triangle of H: registers (compile-time positions only; staged through LDS)
    double xv[MV], lo[MV], up[MV], loN[MV], upN[MV], yv[MV], g[MV], gN[MV], gy[MV];
    double ax[MC], loA[MC], upA[MC], cloN[MC], cupN[MC], yc[MC];
    int sv[MV], sc[MC];
    int nV, nC, fmask, amask, status, infeasible, unbounded, nflips, since_refresh;
    double hscale, hreg;
    long long tlast;                     // (-DRSQP_STAMPS builds)

    __device__ __forceinline__ double Hs(int i, int k) const { return Hr[tri(i, k)]; }                     // static i, k < MV
    __device__ __forceinline__ double As(int i, int k) const { return K[(i * MV + k) * WL]; }             // i may vary by lane
    __device__ __forceinline__ int nFR() const { return __popc(fmask); }
    __device__ __forceinline__ int nAC() const { return __popc(amask); }
    // row q of G (q varies by lane): by symmetry from the stored triangle
    __device__ __forceinline__ void fetch_row(int q, double (&u)[N]) const {
        const int tq = (q * (q + 1)) >> 1;
        SFOR(k, N, const int e = k <= q ? tq + k : k * (k + 1) / 2 + q; u[k] = G[e * WL];);
    }
    __device__ __forceinline__ void store_row(int q, const double (&w)[N]) {
        const int tq = (q * (q + 1)) >> 1;
        SFOR(k, N, const int e = k <= q ? tq + k : k * (k + 1) / 2 + q; G[e * WL] = w[k];);
    }
    __device__ __forceinline__ double Gd(int p, int q) const {
        const int a = p < q ? p : q, b = p < q ? q : p;
        return G[(((b * (b + 1)) >> 1) + a) * WL];
    }

    // ------------------------------------------------------------------ products
    // gyx = A'y_C - H x, axx = A x, hxx = H x (sums in ascending column / row order, as qp_tiny.hip forms them)
    __device__ __forceinline__ void exact_products(double (&gyx)[MV], double (&axx)[MC], double (&hxx)[MV]) const {
        double h[MV], aty[MV], a[MC];
        SFOR(k, MV, h[k] = 0.0; aty[k] = 0.0;);
        SFOR(i, MC, a[i] = 0.0;);
        SFOR(k, MV, SFOR(j, (k) + 1, const double w = Hs(j, k);
                h[j] = fma(w, xv[k], h[j]);
                if (j != k) h[k] = fma(w, xv[j], h[k]);););
        // (h[k] above: the terms j < k arrive in its own pass, the diagonal last of that pass, the terms beyond in later passes)
        SFOR(k, MV, SFOR(i, MC, const double w = As(i, k);
                a[i] = fma(w, xv[k], a[i]);
                aty[k] = fma(w, yc[i], aty[k]);););
                SFOR(k, MV, gyx[k] = aty[k] - h[k]; hxx[k] = h[k];);
                SFOR(i, MC, axx[i] = a[i];);
    }
    // out = G in (in = [inV; inC]), every row's sum in ascending column order
    __device__ __forceinline__ void g_times(const double (&in)[N], double (&o)[N]) const {
        SFOR(k, N, o[k] = 0.0;);
        SFOR(k, N, SFOR(j, (k) + 1, const double w = G[tri(j, k) * WL];
                o[j] = fma(w, in[k], o[j]);
                if (j != k) o[k] = fma(w, in[j], o[k]);););
    }

    // ------------------------------------------------------------------ pivots
    // principal pivot on slot q: G <- G0 - (1 / pi) u~ u~', G0 = G with row and column q zeroed, u~ = u except u~_q = sgn.
    // Every stored pair gets the update; row q (whose old entries do not enter) is then written from scratch.
    __device__ __forceinline__ void pivot1(const double (&u)[N], int q, double sgn, double pi) {
        const double c = -recip(pi);
        double ut[N], t[N];
        SFOR(k, N, ut[k] = k == q ? sgn : u[k]; t[k] = c * ut[k];);
        SFOR(k, N, SFOR(j, (k) + 1, G[tri(j, k) * WL] = fma(t[j], ut[k], G[tri(j, k) * WL]);););
        double w[N];
        const double tq = c * sgn;
        SFOR(k, N, w[k] = tq * ut[k];);
        store_row(q, w);
    }
    // 2 x 2 block pivot on (p, q) with W = [G_pp G_pq; G_pq G_qq]^-1: G <- G00 - U~ W U~', U~ = [u_p u_q], rows p, q = diag(sp, sq)
    __device__ __forceinline__ void pivot2(const double (&up_)[N], const double (&uq)[N], int p, double sp, int q, double sq, double w11,
                                           double w12, double w22) {
        double a[N], b[N], cp[N], cq[N];
        SFOR(k, N, const bool hp = k == p, hq = k == q;
            a[k] = hp ? sp : (hq ? 0.0 : up_[k]);
            b[k] = hq ? sq : (hp ? 0.0 : uq[k]);
            cp[k] = fma(w11, a[k], w12 * b[k]);
            cq[k] = fma(w12, a[k], w22 * b[k]););
            // (one pass; as two passes of one rank-1 term each -- 20 instead of 40 doubles of vectors live -- 0.0449 against 0.0433 ms)
        SFOR(k, N, SFOR(j, (k) + 1, G[tri(j, k) * WL] = fma(-a[j], cp[k], fma(-b[j], cq[k], G[tri(j, k) * WL]));););
        double w[N];
        SFOR(k, N, w[k] = -sp * cp[k];);
        store_row(p, w);
        SFOR(k, N, w[k] = -sq * cq[k];);
        store_row(q, w);
    }

    // ------------------------------------------------------------------ staging: the wave's 64 problems, HBM -> LDS -> registers
    // In HBM a problem's values are consecutive (member-major pools, the layout every kernel of the library shares), so the block
    // of the wave's 64 problems is one contiguous run per array: lane j loads elements j, j + 64, ... (512 consecutive bytes per
    // instruction) and drops each one into the LDS slot of the lane that owns its problem, [entry][owner]; a lane then finds its
    // own values at compile-time offsets. (One lane gathering its own problem touches 64 cache lines per instruction to use 8
    // bytes of each: 4 waves per CU staging that way spent 21 k cycles here, a fifth of the kernel.)
    // T: the wave's LDS block from lane 0's point of view (G = T + lane); nqw: problems of this wave (64, less in the last one)
    template <int CH> __device__ __forceinline__ void load_block(const double *src, int n, int t0, int cnt, int lane, double (&w)[CH]) const {
        // elements m = (t0 + t) * 64 + lane of the block, t < CH (beyond the block: its last element, dropped by drop_block)
        SFOR(t, CH, const int m = (t0 + t) * WL + lane; w[t] = src[m < cnt ? m : cnt - 1];);
    }
    template <int CH> __device__ __forceinline__ void drop_block(ldouble *T, int s0, int n, int t0, int cnt, int lane, const double (&w)[CH]) const {
        const float rn = 1.0f / (float)n;
        SFOR(t, CH, if (t0 + t < n) {
                 const int m = (t0 + t) * WL + lane;
                 const int qq = (int)(((float)m + 0.5f) * rn), e = m - qq * n;      // (m < 64 * 64: the quotient is exact)
                 if (m < cnt) T[(s0 + e) * WL + qq] = w[t];
             });
    }
    // ---- the state blocks of the wave's problems (QPPools::keep_state: what the next hot start continues from), LDS -> HBM. A
    // problem's block is N N + 96 doubles + 20 ints, consecutive, blocks `stride` doubles apart (the layout of qp_tiny.hip, see
    // LANE_TINY_MAGIC). A lane storing its own block piece by piece touches 64 lines per instruction for 8 bytes of each -- 0.17 ms
    // instead of 0.04 for the headline launch with its state kept; instead lane j handles pairs j, j + 64, ... of the wave's run
    // of blocks (two doubles of one problem per instruction, consecutive lanes consecutive pairs): 0.075 ms.
    // The tableau part: both triangles in HBM (what the 8-lane kernel holds), from the upper one in the LDS slots
    __device__ __forceinline__ void state_put_tableau(double *sbase, long long stride, int nqw, int lane, const ldouble *T) const {
        static_assert((N & 1) == 0, "pairs of a row");
        constexpr int PAIRS = N * N / 2;
        // (a run-time loop: nothing here indexes a register array, and 50 unrolled trips were hoisted in front of each other until
        //  1 493 registers spilled)
#pragma unroll 5
        for (int t = 0; t < PAIRS; t++) {
            const int m = t * WL + lane;
            const int qq = m / PAIRS, e = (m - qq * PAIRS) * 2, r = e / N, k = e - r * N;            // entries (r, k), (r, k + 1)
            const int hi0 = r > k ? r : k, lo0 = r > k ? k : r, hi1 = r > k + 1 ? r : k + 1, lo1 = r > k + 1 ? k + 1 : r;
            double2 v;
            v.x = T[(((hi0 * (hi0 + 1)) >> 1) + lo0) * WL + qq]; v.y = T[(((hi1 * (hi1 + 1)) >> 1) + lo1) * WL + qq];
            if (qq < nqw) *reinterpret_cast<double2 *>(sbase + (long long)qq * stride + e) = v;
        }
    }
    // the slot state + status words behind it: TAILD doubles (the ints as pairs), staged in LDS slots 0 .. in two halves
    static constexpr int TAILD = 96 + 10, THALF = TAILD / 2;
    static_assert(THALF <= NT, "the halves of the slot state pass through the tableau's slots");
    __device__ __forceinline__ void state_put_tail_half(double *sbase, long long stride, int nqw, int lane, const ldouble *T, int half) const {
#pragma unroll 4
        for (int t = 0; t < THALF; t++) {
            const int m = t * WL + lane;
            const int qq = m / THALF, f = m - qq * THALF;
            if (qq < nqw) sbase[(long long)qq * stride + N * N + half * THALF + f] = T[f * WL + qq];
        }
    }
    // my slot state as the TAILD doubles behind the tableau: f < 48 field f / 8 (x g lo up gy y) of variable slot f % 8; f < 80 field
    // (f - 48) / 8 (A x, loA, upA, y) of constraint slot (f - 48) % 8 (slots beyond MC: zero); f < 96 the tableau's diagonal (G_ll of the
    // 8 variable slots, then G_{8+l,8+l}: the 8-lane kernel tracks them apart); then the status words two by two: sv (8), sc (8), status, masks, magic, pivots since G was built from the data
    template <int W> __device__ __forceinline__ int state_word(int pivots) const {
        if constexpr (W < 8) return sv[W];
        else if constexpr (W < 16) { if constexpr (W - 8 < MC) return sc[W - 8]; else return 0; }
        else if constexpr (W == 16) return status;
        else if constexpr (W == 17) return (fmask & 0xff) | ((amask & 0xff) << 8);
        else if constexpr (W == 18) return 0x7a11e;
        else return pivots;
    }
    template <int F> __device__ __forceinline__ double tail_value(const double (&dg)[N], int pivots) const {
        if constexpr (F < 48) {
            constexpr int c = F / 8, l = F % 8;
            if constexpr (c == 0) return xv[l]; else if constexpr (c == 1) return g[l]; else if constexpr (c == 2) return lo[l];
            else if constexpr (c == 3) return up[l]; else if constexpr (c == 4) return gy[l]; else return yv[l];
        } else if constexpr (F < 80) {
            constexpr int c = (F - 48) / 8, l = (F - 48) % 8;
            if constexpr (l >= MC) return 0.0;
            else if constexpr (c == 0) return ax[l]; else if constexpr (c == 1) return loA[l]; else if constexpr (c == 2) return upA[l]; else return yc[l];
        } else if constexpr (F < 96) {
            constexpr int c = (F - 80) / 8, l = (F - 80) % 8;
            if constexpr (c == 0) return dg[l]; else if constexpr (l < MC) return dg[MV + l]; else return 0.0;
        } else {
            constexpr int w = (F - 96) * 2;
            return __hiloint2double(state_word<w + 1>(pivots), state_word<w>(pivots));
        }
    }
    template <int HALF> __device__ __forceinline__ void tail_put_half(ldouble *Gm, const double (&dg)[N], int pivots) const {
        SFOR(fl, THALF, Gm[fl * WL] = tail_value<HALF * THALF + fl>(dg, pivots););
    }
    // the reverse for results: every lane has put its n values into slots s0 .. s0 + n - 1 of its own column; lane j stores
    // elements j, j + 64, ... of the wave's contiguous block (64-bit values, or 32-bit ones kept in the low halves of the slots)
    template <int CH, class TV> __device__ __forceinline__ void put_block(TV *dst, int n, int cnt, int lane, const ldouble *T, int s0) const {
        const float rn = 1.0f / (float)n;
        SFOR(t, CH, if (t < n) {
                 const int m = t * WL + lane;
                 const int qq = (int)(((float)m + 0.5f) * rn), e = m - qq * n;
                 if (m < cnt) {
                     if constexpr (sizeof(TV) == 8) dst[m] = T[(s0 + e) * WL + qq];
                     else dst[m] = ((const __attribute__((address_space(3))) int *)(T + (s0 + e) * WL + qq))[0];
                 }
             });
    }
    __device__ __forceinline__ void wave_sync() const {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // drop_block for a matrix: entry e of a problem goes to the slot the wave's table names for it (tab[e]: a slot, or -1 = skip)
    template <int CH> __device__ __forceinline__ void drop_entries(ldouble *T, const __attribute__((address_space(3))) int *tab, int n, int cnt,
                                                                   int lane, const double (&w)[CH]) const {
        const float rn = 1.0f / (float)n;
        SFOR(t, CH, if (t < n) {
                 const int m = t * WL + lane;
                 const int qq = (int)(((float)m + 0.5f) * rn), e = m - qq * n;
                 const int d = tab[e < CH ? e : 0];
                 if (m < cnt && d >= 0) T[d * WL + qq] = w[t];
             });
    }
    // the entries of MY problem's A (into the dense copy) or H (upper triangle into slots 0 ..) from its own CSC arrays: column
    // pointers first, then the entries four at a time (index and value loads of a round in flight together), a column by counting
    // the pointers an entry has passed. Lanes differ in their counts: the loop runs to the longest of the wave
    template <bool ISH> __device__ __forceinline__ void own_entries(const int *jc, const int *ir_pool, const double *val_pool, int off) {
        int cp[MV + 1];
        SFOR(j, (MV) + 1, cp[j] = jc[j <= nV ? j : nV];);
        const int cnt = cp[MV] - cp[0];                       // (cp[j] = cp[nV] beyond nV)
        const int *ir = ir_pool + off; const double *val = val_pool + off;
        for (int e0 = 0; e0 < cnt; e0 += 4) {
            int r[4]; double w[4];
            SFOR(t, 4, const int e = e0 + t < cnt ? e0 + t : cnt - 1; r[t] = ir[e]; w[t] = val[e];);
            SFOR(t, 4, if (e0 + t < cnt) {
                     const int e = cp[0] + e0 + t;
                     int c = 0;
                     SFOR1(j, MV, c += (j <= nV && e >= cp[j]) ? 1 : 0;);
                     if constexpr (ISH) { if (r[t] <= c) G[(((c * (c + 1)) >> 1) + r[t]) * WL] = w[t]; }
                     else K[(r[t] * MV + c) * WL] = w[t];
                 });
        }
    }
    // UNI: the batch has ONE sparsity pattern (QPPools::uni_pat). Otherwise -- one SHAPE, patterns of their own -- the vectors still
    // come through the wave's block, but every lane walks the CSC arrays of its own problem (own_entries)
    template <bool UNI> __device__ __forceinline__ void stage(const QPPools &P, long long q0, int lane, int nqw, ldouble *T) {
        typedef __attribute__((address_space(3))) int lint;
        const int annz = UNI ? P.uni_annz : 0, hnnz = (UNI && P.uni_haveH) ? P.uni_hnnz : 0;       // (entries that travel through the block)
        constexpr int SV = 0, SC = 3 * MV, HC = 16;          // slots of the vectors' pass; H passes through the table when it has <= 16 entries
        static_assert(3 * MV + 2 * MC <= NT && NA <= HC, "staging slots");
        const int cV = nqw * nV, cC = nqw * nC, cA = nqw * annz, cH = nqw * hnnz;
        const bool hfast = hnnz > 0 && hnnz <= HC;
        // lanes beyond the batch (last wave) take the LAST problem's values: they run the same path beside it, store nothing, and
        // stay for the joint work of the wave (staging here, the stores of the results at the end)
        const ldouble *S = T + (lane < nqw ? lane : nqw - 1);
        lint *tab = (lint *)(T + (NT + NA) * WL);            // 2 x 16 destination slots (A, H), behind the tableau and A
        // ---- every load before anything waits: vectors, the entries of A, up to 16 entries of H, and the batch's one pattern
        // (member 0's arrays): lane e < 16 takes entry e of A, lane 16 + e entry e of H -- its row index, its column by counting
        // the column pointers it has passed (wave-uniform scalars)
        double wg[MV], wl[MV], wu[MV], wla[MC], wua[MC], wa[NA], wh[HC];
        load_block<MV>(P.g + q0 * nV, nV, 0, cV, lane, wg);
        load_block<MV>(P.lb + q0 * nV, nV, 0, cV, lane, wl);
        load_block<MV>(P.ub + q0 * nV, nV, 0, cV, lane, wu);
        if (nC > 0) { load_block<MC>(P.lbA + q0 * nC, nC, 0, cC, lane, wla); load_block<MC>(P.ubA + q0 * nC, nC, 0, cC, lane, wua); }
        if (annz > 0) load_block<NA>(P.Aval + q0 * annz, annz, 0, cA, lane, wa);
        if (hfast) load_block<HC>(P.Hval + q0 * hnnz, hnnz, 0, cH, lane, wh);
        if constexpr (UNI) {
        const bool forH = lane >= HC;
        const int pe = forH ? lane - HC : lane, pn = forH ? (hfast ? hnnz : 0) : annz;
        int prow = 0;
        if (pe < pn) prow = forH ? P.Hir[pe] : P.Air[pe];
        int pcol = 0;
        SFOR1(j, MV, const int ja = P.Ajc[j <= nV ? j : nV], jh = hfast ? P.Hjc[j <= nV ? j : nV] : 0;
              const int jj = forH ? jh : ja; pcol += (j <= nV && pe >= jj) ? 1 : 0;);
        LSTAMP(10);
        if (lane < 2 * HC) tab[lane] = pe < pn ? (forH ? (prow <= pcol ? ((pcol * (pcol + 1)) >> 1) + prow : -1) : NT + prow * MV + pcol) : -1;
        }
        SFOR(e, NA, K[e * WL] = 0.0;);
        wave_sync();
        drop_block<MV>(T, SV, nV, 0, cV, lane, wg);
        drop_block<MV>(T, SV + MV, nV, 0, cV, lane, wl);
        drop_block<MV>(T, SV + 2 * MV, nV, 0, cV, lane, wu);
        if (nC > 0) { drop_block<MC>(T, SC, nC, 0, cC, lane, wla); drop_block<MC>(T, SC + MC, nC, 0, cC, lane, wua); }
        if (annz > 0) drop_entries<NA>(T, tab, annz, cA, lane, wa);          // (straight to their places in the dense copy)
        wave_sync();
        LSTAMP(11);
        // ---- my vectors (slots beyond the sizes: neutral values)
        double lb_[MV], ub_[MV], la_[MC], ua_[MC];
        SFOR(l, MV, const bool v = l < nV;
             const double a0 = S[(SV + l) * WL], a1 = S[(SV + MV + l) * WL], a2 = S[(SV + 2 * MV + l) * WL];
             gN[l] = v ? a0 : 0.0; lb_[l] = v ? a1 : 0.0; ub_[l] = v ? a2 : 0.0;);
        SFOR(i, MC, const bool c = i < nC;
             const double a0 = S[(SC + i) * WL], a1 = S[(SC + MC + i) * WL];
             la_[i] = c ? a0 : -RSQP_INFTY; ua_[i] = c ? a1 : RSQP_INFTY;);
        if (UNI && lane >= nqw) {        // (a lane beyond the batch: its own column of A is the last problem's as well)
            SFOR(e, NA, K[e * WL] = S[(NT + e) * WL];);
        }
        const long long qown = q0 + (lane < nqw ? lane : nqw - 1);
        if constexpr (!UNI) own_entries<false>(P.Ajc + qown * (nV + 1), P.Air, P.Aval, P.desc[qown].offAnz);
        wave_sync();           // (every lane has read its slots: the space is the tableau's / H's from here on)
        LSTAMP(12);
        // ---- H: the upper triangle is collected in slots 0 .. NH - 1 (H arrives with both triangles: the table skips the lower one).
        // Up to 16 entries per problem (the headline's H has 11) come through the table like A; a fuller H is gathered by its owner
        // (a round of 16 load instructions covers ALL entries of 1024 / hnnz problems, not 16 entries of each)
        SFOR(e, NH, G[e * WL] = 0.0;);
        wave_sync();
        if (hfast) {
            drop_entries<HC>(T, tab + HC, hnnz, cH, lane, wh);
            wave_sync();
            if (lane >= nqw) {
                SFOR(e, NH, G[e * WL] = S[e * WL];);
            }
        } else if (!UNI) {
            if (P.uni_haveH) own_entries<true>(P.Hjc + qown * (nV + 1), P.Hir, P.Hval, P.desc[qown].offHnz);
        } else if (hnnz > HC) {
            int hjc[MV + 1];
            SFOR(j, (MV) + 1, hjc[j] = P.Hjc[j <= nV ? j : nV]; if (j > nV) hjc[j] = 0x7fffffff;);
            const double *gH = P.Hval + (q0 + (lane < nqw ? lane : nqw - 1)) * hnnz;
            for (int e0 = 0; e0 < hnnz; e0 += 8) {
                double w[8];
                SFOR(t, 8, w[t] = gH[e0 + t < hnnz ? e0 + t : hnnz - 1];);
                SFOR(t, 8, if (e0 + t < hnnz) {
                         const int e = e0 + t, r = P.Hir[e];
                         int c = 0;
                         SFOR1(j, MV, c += e >= hjc[j] ? 1 : 0;);
                         if (r <= c) G[(((c * (c + 1)) >> 1) + r) * WL] = w[t];
                     });
            }
        }
        // (H + hreg I: the LP regularisation; hscale = the largest diagonal entry)
        hscale = 0.0;
        SFOR(k, MV, SFOR(j, k + 1, double w = G[tri(j, k) * WL];
                         if (j == k) { w = (k < nV) ? w + hreg : w; hscale = fmax(hscale, k < nV ? fabs(w) : 0.0); }
                         Hr[tri(j, k)] = w;););
        LSTAMP(13);
        SFOR(l, MV, const bool v = l < nV;
             loN[l] = v ? clampinf(lb_[l]) : 0.0; upN[l] = v ? clampinf(ub_[l]) : 0.0;);
        SFOR(i, MC, const bool c = i < nC;
             cloN[i] = c ? clampinf(la_[i]) : -RSQP_INFTY; cupN[i] = c ? clampinf(ua_[i]) : RSQP_INFTY;);
    }
    __device__ __forceinline__ void g_from_K() {     // S empty: G = -K
        SFOR(k, MV, SFOR(j, k + 1, G[tri(j, k) * WL] = -Hs(j, k);););
        SFOR(i, MC, SFOR(k, MV, G[tri(k, MV + i) * WL] = -As(i, k);); SFOR(j, i + 1, G[tri(MV + j, MV + i) * WL] = 0.0;););
    }
    __device__ __forceinline__ bool bounds_inconsistent() const {
        bool bad = false;
        SFOR(l, MV, bad = bad || (l < nV && loN[l] > upN[l] + RSQP_EPS););
        SFOR(i, MC, bad = bad || (i < nC && cloN[i] > cupN[i] + RSQP_EPS););
        return bad;
    }

    // ------------------------------------------------------------------ auxiliary QP of a cold start (setup_aux of the CPU restatement)
    // x = 0, y = 0, every variable on a finite bound (lower first); variables without one are free and enter S by principal pivots
    // (those with curvature, repeatedly: a pivot may give the next one its curvature). false: free variables are left over
    __device__ __forceinline__ bool setup_cold() {
        status = QPS_PREPARINGAUXILIARYQP;
        infeasible = unbounded = 0;
        int pf = 0;
        SFOR(l, MV, xv[l] = 0.0; yv[l] = 0.0;
            int s = -1;
            if (loN[l] <= -RSQP_INFTY) s = upN[l] < RSQP_INFTY ? 1 : 0;
            sv[l] = l < nV ? s : -1;
            pf |= (l < nV && s == 0) ? 1 << l : 0;);
            SFOR(i, MC, yc[i] = 0.0; ax[i] = 0.0; sc[i] = 0;);
        g_from_K();
        fmask = amask = 0;
        for (int round = 0; round < 2 * N && pf != 0; round++) {
            bool progress = false;
#pragma unroll 1
            for (int v = 0; v < MV; v++)
                if ((pf >> v) & 1) {
                    const double pi = Gd(v, v);
                    if (-pi > 1e-8 * hscale) {
                        double u[N];
                        fetch_row(v, u);
                        pivot1(u, v, 1.0, pi);
                        fmask |= 1 << v; pf &= ~(1 << v); progress = true;
                    }
                }
            if (!progress) break;
        }
        if (pf != 0) return false;
        SFOR(l, MV, gy[l] = 0.0; g[l] = 0.0;
            lo[l] = sv[l] == -1 ? xv[l] : fmin(loN[l], xv[l] - RSQP_BOUND_RELAXATION);
            up[l] = sv[l] == 1 ? xv[l] : fmax(upN[l], xv[l] + RSQP_BOUND_RELAXATION);
            if (l >= nV) { lo[l] = 0.0; up[l] = 0.0; });
            SFOR(i, MC, loA[i] = fmin(cloN[i], ax[i] - RSQP_BOUND_RELAXATION);
            upA[i] = fmax(cupN[i], ax[i] + RSQP_BOUND_RELAXATION););
        status = QPS_AUXILIARYQPSOLVED;
        return true;
    }

    // ------------------------------------------------------------------ one working-set change
    // kind 1 constraint idx leaves | 2 bound of idx leaves | 3 constraint idx enters at `side` | 4 variable idx gets fixed at `side`
    // (style of the slot passes here and in homotopy(): every arm of a choice is computed first, then selected -- plain selects
    //  between values; a ternary with arithmetic in its arms becomes a tree of divergent branches per slot)
    __device__ __forceinline__ int change(int kind, int idx, int side, double tau, bool &treat_done) {
        const bool isc = (kind == 1) | (kind == 3);
        const int q = isc ? MV + idx : idx;
        double u[N];
        fetch_row(q, u);
        const double pi = Gd(q, q);
        bool flip = false;
        if (kind == 1) {
            double d2 = 0.0;
            SFOR(k, MV, const double d2n = fma(u[k], u[k], d2); d2 = ((fmask >> k) & 1) ? d2n : d2;);
            flip = !(d2 > 0.0 && -pi > 1e-8 * hscale * d2);
        } else if (kind == 2) flip = !(-pi > 1e-8 * hscale);
        if (flip) {
            // the released direction has no curvature: the constraint / bound goes to its OPPOSITE side, G is unchanged
            double opp = 0.0;
            SFOR(i, MC, const double o1 = sc[i] == -1 ? cupN[i] : cloN[i]; opp = (isc & (i == idx)) ? o1 : opp;);
            SFOR(l, MV, const double o1 = sv[l] == -1 ? upN[l] : loN[l]; opp = (!isc & (l == idx)) ? o1 : opp;);
            if (opp >= RSQP_INFTY || opp <= -RSQP_INFTY) return RET_UNBOUNDED;
            SFOR(i, MC, const bool my = isc & (i == idx); const bool wl = sc[i] == -1, wu = sc[i] == 1;
                 upA[i] = (my & wl) ? ax[i] : upA[i]; loA[i] = (my & wu) ? ax[i] : loA[i];
                 yc[i] = my ? 0.0 : yc[i]; sc[i] = my ? -sc[i] : sc[i];);
            SFOR(l, MV, const bool my = !isc & (l == idx); const bool wl = sv[l] == -1, wu = sv[l] == 1;
                 up[l] = (my & wl) ? xv[l] : up[l]; lo[l] = (my & wu) ? xv[l] : lo[l];
                 yv[l] = my ? 0.0 : yv[l]; sv[l] = my ? -sv[l] : sv[l];);
            nflips++;
            since_refresh = REFRESH;
            return RET_OK;
        }
        int pk = 0, pidx = -1;
        double ynew = 0.0;
        if (kind >= 3) {
            const double sg = kind == 3 ? -1.0 : 1.0;
            // sum of u_v^2 over the free variables; sum of a_v^2 over them for the incoming row (row idx of A, or e_idx)
            double arow[MV];
            double pn2 = 0.0, na2 = 0.0;
            const int ia = kind == 3 ? idx : 0;
            SFOR(k, MV, const bool fr = (fmask >> k) & 1;
                 const double aA = As(ia, k), aE = k == idx ? 1.0 : 0.0;
                 arow[k] = kind == 3 ? aA : aE;
                 const double p1 = fma(u[k], u[k], pn2), n1 = fma(arow[k], arow[k], na2);
                 pn2 = fr ? p1 : pn2; na2 = fr ? n1 : na2;);
            int li;
            if (nFR() - nAC() <= 0 || !(na2 > 0.0)) li = 0;
            else {
                const double p2 = hscale * hscale * pn2;
                li = p2 > 1e-12 * na2 ? 1 : (p2 < 1e-24 * na2 ? 0 : -1);
            }
            if (li < 0) {
                // the band: the residual of the row's representation by the active rows decides (qp_small_g.h)
                double rn2 = 0.0;
                SFOR(l, MV, double r = l < nV ? arow[l] : 0.0;
                     SFOR(i, MC, const double r1 = fma(-As(i, l), sg * u[MV + i], r); r = ((amask >> i) & 1) ? r1 : r;);
                     rn2 += ((l < nV) & (sv[l] == 0)) ? r * r : 0.0;);
                li = rn2 > 9e-16 * na2 ? 1 : 0;                 // |r| / |a_FR| > 3e-8
            }
            if (li == 0) {
                // ---- exchange: shift the multipliers along the dependency until one of them reaches zero; that one leaves
                const double sgn = side == 1 ? -1.0 : 1.0, ss = sgn * sg;
                double xiv[MV], xic[MC];
                double bt = RSQP_INFTY;
                int bid = 0x7fffffff;
                SFOR(i, MC, const bool on = (i < nC) & (sc[i] != 0), wl = sc[i] == -1;
                     const double xu = ss * u[MV + i];
                     xic[i] = on ? xu : 0.0;
                     const double num = wl ? yc[i] : -yc[i], den = wl ? xic[i] : -xic[i];
                     if (on & (den > RSQP_EPS_DEN)) {
                         const double t = (num > 0.0 ? num : 0.0) / den;
                         const bool better = (t < bt) | ((t == bt) & (i < bid));
                         bt = better ? t : bt; bid = better ? i : bid;
                     });
                SFOR(l, MV, const bool on = (l < nV) & (sv[l] != 0), wl = sv[l] == -1;
                     const double xu = ss * u[l];
                     xiv[l] = on ? xu : 0.0;
                     const double num = wl ? yv[l] : -yv[l], den = wl ? xiv[l] : -xiv[l];
                     if (on & (den > RSQP_EPS_DEN)) {
                         const double t = (num > 0.0 ? num : 0.0) / den;
                         const bool better = (t < bt) | ((t == bt) & (nC + l < bid));
                         bt = better ? t : bt; bid = better ? nC + l : bid;
                     });
                if (bid == 0x7fffffff) {
                    // no partner: infeasible beyond this point of the homotopy -- unless that point IS its end to rounding
                    if (tau >= 1.0 - 1e-9) { treat_done = true; return RET_OK; }
                    return RET_INFEASIBLE;
                }
                if (bid < nC) { pk = 1; pidx = bid; } else { pk = 2; pidx = bid - nC; }
                SFOR(l, MV, yv[l] -= bt * xiv[l];);
                SFOR(i, MC, yc[i] -= bt * xic[i];);
                ynew = sgn * bt;
                const int p = pk == 1 ? MV + pidx : pidx;
                double u2[N];
                fetch_row(p, u2);
                const double pp = Gd(p, p), qq = pi, pq = Gd(p, q);
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
        const bool leaves = (kind == 1) | (kind == 2);
        const int snew = leaves ? 0 : side;
        const double ynw = leaves ? 0.0 : ynew;
        SFOR(l, MV, const bool my = !isc & (l == idx), pr = (pk == 2) & (l == pidx);
             const int s1 = pr ? 0 : sv[l]; const double y1 = pr ? 0.0 : yv[l];
             sv[l] = my ? snew : s1; yv[l] = my ? ynw : y1;);
        SFOR(i, MC, const bool my = isc & (i == idx), pr = (pk == 1) & (i == pidx);
             const int s1 = pr ? 0 : sc[i]; const double y1 = pr ? 0.0 : yc[i];
             sc[i] = my ? snew : s1; yc[i] = my ? ynw : y1;);
        if (kind == 1) amask &= ~(1 << idx); else if (kind == 2) fmask |= 1 << idx; else if (kind == 3) amask |= 1 << idx; else fmask &= ~(1 << idx);
        if (pk == 1) amask &= ~(1 << pidx); else if (pk == 2) fmask |= 1 << pidx;
        return RET_OK;
    }

    __device__ __forceinline__ int homotopy(int maxit, int &nWSR) {
        int iter = 0, rcode = RET_OK;
        status = QPS_PERFORMINGHOMOTOPY;
        since_refresh = REFRESH;
        SFOR(l, MV, const double l1 = fmin(loN[l], xv[l] - RSQP_BOUND_RELAXATION), u1 = fmax(upN[l], xv[l] + RSQP_BOUND_RELAXATION);
             const bool c1 = (sv[l] != -1) & (lo[l] <= -RSQP_INFTY) & (loN[l] > -RSQP_INFTY);
             const bool c2 = (sv[l] != 1) & (up[l] >= RSQP_INFTY) & (upN[l] < RSQP_INFTY);
             lo[l] = c1 ? l1 : lo[l]; up[l] = c2 ? u1 : up[l];);
        SFOR(i, MC, const double l1 = fmin(cloN[i], ax[i] - RSQP_BOUND_RELAXATION), u1 = fmax(cupN[i], ax[i] + RSQP_BOUND_RELAXATION);
             const bool c1 = (sc[i] != -1) & (loA[i] <= -RSQP_INFTY) & (cloN[i] > -RSQP_INFTY);
             const bool c2 = (sc[i] != 1) & (upA[i] >= RSQP_INFTY) & (cupN[i] < RSQP_INFTY);
             loA[i] = c1 ? l1 : loA[i]; upA[i] = c2 ? u1 : upA[i];);
        for (;;) {
            // ---- x exactly on its active bounds; (exact products); drift correction + input of the product
            SFOR(l, MV, const int s_ = opq(sv[l]); const double xb = s_ == -1 ? lo[l] : up[l]; xv[l] = s_ != 0 ? xb : xv[l];);
            if (since_refresh >= REFRESH) {
                double gyx[MV], axx[MC], hxx[MV];
                exact_products(gyx, axx, hxx);
                SFOR(l, MV, gy[l] = gyx[l];);
                SFOR(i, MC, ax[i] = i < nC ? axx[i] : 0.0;);
                since_refresh = 0;
            }
            double in[N], o[N];
            SFOR(l, MV, const int s_ = opq(sv[l]); const double gl = gy[l] + yv[l]; g[l] = gl;
                 const double a0 = -(gN[l] - gl), a1 = loN[l] - lo[l], a2 = upN[l] - up[l];
                 const double a12 = s_ == -1 ? a1 : a2;
                 in[l] = s_ == 0 ? a0 : a12;);              // (slots beyond nV: fixed at 0 = lo = loN)
            SFOR(i, MC, const int s_ = opq(sc[i]); const bool wl = s_ == -1, wu = s_ == 1;
                 loA[i] = wl ? ax[i] : loA[i]; upA[i] = wu ? ax[i] : upA[i];
                 const double a1 = cloN[i] - loA[i], a2 = cupN[i] - upA[i];
                 const double a12 = wl ? a1 : a2;
                 in[MV + i] = s_ == 0 ? 0.0 : a12;);
            LSTAMP(5);
            // ---- out = G in; dx / dy / A dx and the ratio test over my slots (ties go to the lowest id)
            g_times(in, o);
            LSTAMP(6);
            double dxv[MV], dyv[MV], hd[MV], dax[MC], dyc[MC];
            double bt = 1.0;
            int bid = 0x7fffffff;
            // (one branch per candidate, around its division: perturbations of one QP agree on which candidates are dead -- half
            //  of them on the headline batch --, and the wave skips those)
            auto cand = [&](double num, double den, int id, bool ok) {
                const bool live = ok & (den >= RSQP_EPS_DEN);
                if (live) {
                    const double t = (num > 0.0 ? num : 0.0) / den;
                    const bool better = (t < bt) | ((t == bt) & (id < bid));
                    bt = better ? t : bt; bid = better ? id : bid;
                }
            };
            SFOR(i, MC, const int s_ = opq(sc[i]); const double oc = o[MV + i], noc = -oc;
                 const bool act = s_ != 0, wl = s_ == -1;
                 dax[i] = act ? in[MV + i] : noc; dyc[i] = act ? noc : 0.0;
                 const double nA = wl ? yc[i] : -yc[i], nI = ax[i] - loA[i];
                 const double dA = wl ? -dyc[i] : dyc[i], dI = (cloN[i] - loA[i]) - dax[i];
                 const double num1 = act ? nA : nI, den1 = act ? dA : dI;
                 const int id1 = act ? i : nC + nV + i;
                 cand(num1, den1, id1, (i < nC) & (act | (cloN[i] > -RSQP_INFTY)));
                 cand(upA[i] - ax[i], dax[i] - (cupN[i] - upA[i]), 2 * nC + nV + i, (i < nC) & !act & (cupN[i] < RSQP_INFTY)););
            SFOR(l, MV, const int s_ = opq(sv[l]); const double ov = o[l], dg = gN[l] - g[l], nov = -ov, dgo = dg - ov, ndg = -dg;
                 const bool fr = s_ == 0, wl = s_ == -1;
                 dxv[l] = fr ? ov : in[l]; dyv[l] = fr ? 0.0 : dgo; hd[l] = fr ? ndg : nov;
                 const double nA = wl ? yv[l] : -yv[l], nI = xv[l] - lo[l];
                 const double dA = wl ? -dyv[l] : dyv[l], dI = (loN[l] - lo[l]) - dxv[l];
                 const double num1 = fr ? nI : nA, den1 = fr ? dI : dA;
                 const int id1 = fr ? 3 * nC + nV + l : nC + l;
                 cand(num1, den1, id1, (l < nV) & (!fr | (loN[l] > -RSQP_INFTY)));
                 cand(up[l] - xv[l], dxv[l] - (upN[l] - up[l]), 3 * nC + 2 * nV + l, (l < nV) & fr & (upN[l] < RSQP_INFTY)););
            if (!(bt < 1.0)) { bt = 1.0; bid = 0x7fffffff; }
            LSTAMP(7);
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
            const bool done = kind == 0;
            const bool cap = iter >= maxit;
            const bool go = !done & !cap;
            // ---- homotopy step on my slots
            SFOR(l, MV, const int s_ = opq(sv[l]); yv[l] += tau * dyv[l];
                 const double xn = xv[l] + tau * dxv[l];
                 const double l1 = lo[l] + tau * (loN[l] - lo[l]), u1 = up[l] + tau * (upN[l] - up[l]);
                 const double gn = g[l] + tau * (gN[l] - g[l]);
                 const bool hit = go & (kind == 4) & (l == idx);
                 const double xe1 = s_ == 1 ? upN[l] : xn, xe = s_ == -1 ? loN[l] : xe1;
                 xv[l] = done ? xe : xn;
                 g[l] = done ? gN[l] : gn;
                 gy[l] -= tau * hd[l];
                 const double l2 = (hit & (side == -1)) ? xn : l1, u2 = (hit & (side == 1)) ? xn : u1;
                 lo[l] = done ? loN[l] : l2; up[l] = done ? upN[l] : u2;);
            SFOR(i, MC, yc[i] += tau * dyc[i];
                 const double an = ax[i] + tau * dax[i];
                 const double l1 = loA[i] + tau * (cloN[i] - loA[i]), u1 = upA[i] + tau * (cupN[i] - upA[i]);
                 const bool hit = go & (kind == 3) & (i == idx);
                 ax[i] = done ? ax[i] : an;
                 const double l2 = (hit & (side == -1)) ? an : l1, u2 = (hit & (side == 1)) ? an : u1;
                 loA[i] = done ? cloN[i] : l2; upA[i] = done ? cupN[i] : u2;);
            LSTAMP(8);
            if (done || cap) {
                if (done) status = QPS_SOLVED; else rcode = RET_MAX_NWSR;
                break;
            }
            bool treat_done = false;
            rcode = change(kind, idx, side, tau, treat_done);
            LSTAMP(9);
            if (treat_done) {      // (see change: no exchange partner at the very end of the homotopy)
                SFOR(l, MV, g[l] = gN[l]; lo[l] = loN[l]; up[l] = upN[l];
                     const double xb = sv[l] == -1 ? loN[l] : upN[l]; xv[l] = sv[l] != 0 ? xb : xv[l];);
                SFOR(i, MC, loA[i] = cloN[i]; upA[i] = cupN[i];);
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
        double gyx[MV], axx[MC], hxx[MV];
        if (refine) {
            exact_products(gyx, axx, hxx);
            double in[N], o[N];
            SFOR(l, MV, in[l] = (l < nV && sv[l] == 0) ? -(gN[l] - gyx[l]) : 0.0;);
            SFOR(i, MC, in[MV + i] = sc[i] != 0 ? (sc[i] == 1 ? cupN[i] : cloN[i]) - axx[i] : 0.0;);
            g_times(in, o);
            SFOR(l, MV, if (l < nV && sv[l] == 0) xv[l] += o[l];);
            SFOR(i, MC, if (sc[i] != 0) yc[i] -= o[MV + i];);
        }
        exact_products(gyx, axx, hxx);
        double t[MV];
        // (the objective excludes the LP regularisation)
        SFOR(l, MV, if (refine && l < nV) yv[l] = sv[l] != 0 ? gN[l] - gyx[l] : 0.0;
             t[l] = l < nV ? xv[l] * fma(0.5, hxx[l] - hreg * xv[l], gN[l]) : 0.0;);
        SFOR(i, MC, ax[i] = i < nC ? axx[i] : 0.0;);
        return ((t[0] + t[1]) + (t[2] + t[3])) + ((t[7] + t[6]) + (t[5] + t[4]));      // (the order of qp_tiny.hip's lane tree)
    }
};

// persistent state of a problem between solves (hot starts): the layout qp_tiny.hip keeps in the problem's state block. This
// kernel only WRITES it (KEEP): every hot start of a batch -- new vectors, new matrices, warm re-initialisation -- runs on the 8-lane
// kernel, which continues from what either kernel left:
// [N x N tableau, both triangles][6 fields x 8 variable slots: x g lo up gy y][4 fields x 8 constraint slots: A x, loA, upA, y][G_ll (8)]
// [G_{8+l,8+l} (8)][ints: sv (8), sc (8), status, masks, magic, pivots since the tableau was built from the data]
// (Hot starts were built for this mapping as well -- state block -> LDS -> registers, the guess of a hot start with new matrices
// turned into a tableau by single and 2 x 2 pivots -- and matched the CPU restatement's hot-start sequences in every test; they were 2-3 x
// SLOWER than the 8-lane kernel (0.23 / 0.25 ms against 0.097 / 0.13 ms for 65 536 members): 1.6 KB of state per problem each way
// with one wave per SIMD to hide the round trips, 100-144 KB of code, 340-960 registers spilled to scratch. Removed.)
constexpr int LANE_TINY_MAGIC = 0x7a11e;

template <int MC, bool KEEP, bool UNI>
__global__ void __launch_bounds__(WL) lane_qp_kernel(QPPools P, int nq, int maxWSR) {
    typedef LaneT<MC> ENG;
    constexpr int N = ENG::N;
    __shared__ __attribute__((aligned(16))) double lds[(ENG::NT + ENG::NA) * WL + 16];      // (+ the staging table: 32 ints)
    ldouble *T = (ldouble *)lds;
    const int lane = (int)threadIdx.x;
    const int q0 = (int)blockIdx.x * WL, q = q0 + lane;
    const int nqw = nq - q0 < WL ? nq - q0 : WL;
    ENG E;
    E.G = T + lane; E.K = E.G + ENG::NT * WL;
    E.nV = P.uniV; E.nC = P.uniC; E.hreg = P.uni_hreg;
    E.nflips = 0; E.infeasible = E.unbounded = 0; E.status = QPS_NOTINITIALISED; E.fmask = E.amask = 0; E.since_refresh = 0;
    const int nV = E.nV, nC = E.nC;
#ifdef RSQP_STAMPS
    E.tlast = clock64();
    long long &tlast = E.tlast;
#endif
    E.template stage<UNI>(P, q0, lane, nqw, T);
    LSTAMP(0);
    int rcode = RET_OK, nWSR = 0, setup_pivots = 0;
    if (E.bounds_inconsistent()) {
        // qpOASES areBoundsConsistent: infeasible before any change
        E.infeasible = 1;
        rcode = RET_INFEASIBLE;
        SFOR(l, MV, E.xv[l] = 0.0; E.yv[l] = 0.0; E.sv[l] = -1; E.g[l] = E.gy[l] = 0.0; E.lo[l] = E.up[l] = 0.0;);
        SFOR(i, MC, E.yc[i] = 0.0; E.sc[i] = 0; E.ax[i] = 0.0; E.loA[i] = E.upA[i] = 0.0;);
        E.g_from_K();
    } else {
        const bool ok = E.setup_cold();
        LSTAMP(1);
        setup_pivots = E.nFR() + E.nAC();
        if (!ok) rcode = RET_SETUP_FAILED;
        else rcode = E.homotopy(maxWSR, nWSR);
        LSTAMP(2);
    }
    // (the refinement step repairs what the update-only tableau accumulates: with at most 4 pivots there is nothing to repair yet)
    const int pivots = setup_pivots + nWSR;
    const double obj = E.finish(rcode == RET_OK && pivots > 4);
    LSTAMP(3);
    if constexpr (KEEP) {
        // ---- the state first (its tableau part IS the LDS slots the rest is about to pass through)
        double *sbase = P.state + (long long)q0 * P.uni_state;
        double dg[N];
        SFOR(k, N, dg[k] = E.G[tri(k, k) * WL];);
        E.wave_sync();
        E.state_put_tableau(sbase, P.uni_state, nqw, lane, T);
        E.wave_sync();
        E.template tail_put_half<0>(E.G, dg, pivots);
        E.wave_sync();
        E.state_put_tail_half(sbase, P.uni_state, nqw, lane, T, 0);
        E.wave_sync();
        E.template tail_put_half<1>(E.G, dg, pivots);
        E.wave_sync();
        E.state_put_tail_half(sbase, P.uni_state, nqw, lane, T, 1);
        LSTAMP(15);
    }
    // ---- results (x, y = [bounds; constraints], working set, status / nWSR / objective): the member-major vectors leave through the
    // wave's LDS block like the inputs came (a lane storing its own problem's 8-byte pieces touched 64 lines per instruction)
    {
        typedef __attribute__((address_space(3))) int lint;
        constexpr int SX = 0, SY = MV, SB = 2 * MV + MC, SW = 3 * MV + MC;          // slots: x | y | ws_b | ws_c (<= 3 MV + 2 MC <= NT)
        E.wave_sync();
        SFOR(l, MV, E.G[(SX + l) * WL] = E.xv[l]; ((lint *)(E.G + (SB + l) * WL))[0] = E.sv[l];);
        // y of a problem = [nV bound multipliers; nC constraint multipliers]: its slots follow the problem's own nV
        SFOR(l, MV, if (l < nV) E.G[(SY + l) * WL] = E.yv[l];);
        SFOR(i, MC, if (i < nC) E.G[(SY + nV + i) * WL] = E.yc[i]; ((lint *)(E.G + (SW + i) * WL))[0] = E.sc[i];);
        E.wave_sync();
        E.template put_block<MV>(P.x + (long long)q0 * nV, nV, nqw * nV, lane, T, SX);
        E.template put_block<MV + MC>(P.y + (long long)q0 * (nV + nC), nV + nC, nqw * (nV + nC), lane, T, SY);
        E.template put_block<MV>(P.ws_b + (long long)q0 * nV, nV, nqw * nV, lane, T, SB);
        if (nC > 0) E.template put_block<MC>(P.ws_c + (long long)q0 * nC, nC, nqw * nC, lane, T, SW);
    }
    if (q < nq) {
        const int st = E.status;
        P.status[q] = E.infeasible ? 100 + st : (E.unbounded ? 200 + st : st);
        P.ret[q] = rcode; P.nwsr[q] = nWSR; P.nflips[q] = E.nflips; P.obj[q] = obj;
    }
    LSTAMP(4);
}

}  // namespace

// 1 if this launch is served by the lane-per-problem kernel: a cold start of a one-shape batch of at most 8 x 2 with more members
// (16 384) than 8 lanes per problem hold at a time; no certificate / doorbell of a single-QP handle, no warm re-initialisation
// inputs. A batch that keeps its state gets it written in the 8-lane kernel's layout; one that does not leaves no mark either
// (the handle remembers: QPPools::skip_mark)
int rsqp_lane_fits(const SmallKnobs &kn, const QPPools &p, int nq, int nVmax, int nCmax, int mode) {
    if (kn.lane == 0) return 0;
    // one shape (every member nV x nC); one sparsity pattern (member 0's arrays serve all) or patterns of their own (each lane walks its own)
    if (!(p.uniV >= 1 && p.uniV <= MV && p.uniC >= 0 && p.uniC <= 2 && nVmax <= MV && nCmax <= 2 && (p.uni_pat || p.desc))) return 0;
    if (mode != 0 || (!p.keep_state && !p.skip_mark) || p.cert_out || p.done_flag || !p.tiny_ok || p.x0 || p.y0 || p.guess_b) return 0;
    // (measured, tools/lane_vs_tiny_sweep.py: a launch of this kernel takes 36 us up to 16 384 problems and 43 us at 65 536 -- one
    //  round of waves either way; the 8-lane kernel holds 16 384 problems at a time: 21 us up to 8 192, 26 us at 16 384, 40 us at
    //  20 480 (second round), 47 us at 32 768, 90 us at 65 536. With the state kept, 65 536 members: 0.075 against 0.132 ms)
    return nq >= (kn.lane > 0 ? kn.lane : 16385) ? 1 : 0;
}
hipError_t rsqp_launch_lane_qp(const QPPools &p, int nq, int maxWSR, hipStream_t stream) {
    if (nq <= 0) return hipSuccess;
    const dim3 grid((unsigned)((nq + WL - 1) / WL)), block(WL);
    if (p.uni_pat) {
        if (p.keep_state) hipLaunchKernelGGL((lane_qp_kernel<2, true, true>), grid, block, 0, stream, p, nq, maxWSR);
        else hipLaunchKernelGGL((lane_qp_kernel<2, false, true>), grid, block, 0, stream, p, nq, maxWSR);
    } else {
        if (p.keep_state) hipLaunchKernelGGL((lane_qp_kernel<2, true, false>), grid, block, 0, stream, p, nq, maxWSR);
        else hipLaunchKernelGGL((lane_qp_kernel<2, false, false>), grid, block, 0, stream, p, nq, maxWSR);
    }
    return hipGetLastError();
}
