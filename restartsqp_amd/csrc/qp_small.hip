// qp_small.hip -- batched online active-set QP engine for gfx950: the whole solver state of a
// problem (Q, T, R, iterate, working set) resident in LDS, L = 16 / 32 / 64 lanes of a wave per
// problem (64 / L problems share a one-wave workgroup).
//
// Replaces, for hs0xx-scale problems, the qpOASES 3.2.1 SQProblem::init / hotstart calls
// made at reference src/qpOASESInterface.cpp:155,180,184,191,197,204. Algorithm = dense
// null-space online active-set strategy (see DESIGN.md section "Algorithm"):
//   A_AC,FR * Q = [0 T]  (T reverse triangular),  R'R = Z'HZ,  Givens up/down-dates,
//   primal + dual ratio tests with lowest-candidate-id tie break, exchange on linear
//   dependence, bound flipping when Z'HZ would lose definiteness.
//
// MI355X mapping:
//   * workgroup = one wave = 64 / L problems, grid = ceil(nq * L / 64) -- a batch fills the 256
//     CUs with independent problems; no inter-workgroup communication and no s_barrier: the
//     problems of a wave follow their own control flow under exec masking, and a wave's LDS
//     instructions execute in program order, so a compiler fence is all the sync it needs.
//     hs0xx-scale problems (nV <= 16) use L = 16: their vectors never filled 64 lanes.
//   * LDS image per problem (rsqp_image_bytes): Q and R column-major with an ODD leading
//     dimension so that the lane<->row and lane<->column access patterns below are both
//     bank-conflict free for ds_read_b64; T row-major with the same stride.
//   * sparse H / A stay in global memory in CSC (+ a CSR copy of A): they are read-only
//     and L2-resident; lane-per-row / lane-per-column products, no atomics.
//   * reductions are butterflies over the L lanes of a problem (DPP permutations inside a row, identical result
//     in each of them) so control flow stays uniform per problem; argmin carries the candidate
//     id for the deterministic tie break. With nV <= L every lane owns at most one entry of a
//     vector, so the sums are bit-identical for every L (tools/small_pack_check.py).
//   * the image is written back to HBM at the end of a solve and reloaded by the next
//     hot start (qpOASES keeps the same data inside the SQProblem object).
#include <cstdlib>
#include <type_traits>

#include "rsqp_internal.h"

// A workgroup is ONE wave: LDS instructions of a wave execute in program order, so making a
// write visible to the other lanes only needs the compiler to keep the order (no s_barrier,
// which would also be illegal inside the per-problem divergent control flow of packed waves).
// (L > 64: a problem owns several waves of its workgroup -- a real barrier.)
#define SYNC()                                                   \
    do {                                                         \
        if constexpr (L > 64) {                                  \
            __syncthreads();                                     \
        } else {                                                 \
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
            __builtin_amdgcn_wave_barrier();                     \
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
        }                                                        \
    } while (0)
#define PFOR(i, n) for (int i = lane; i < (n); i += L)
// packed upper-Hessenberg R of the Givens / TQ engine: element (row r, column c), r <= c + 1
#define RIX(c, r) ((c) * ((c) + 3) / 2 + (r))
// LDS-qualified pointer types: guarantees ds_read / ds_write (a generic pointer would be
// lowered to flat_load, which is several times slower and costs two registers)
#define LDS __attribute__((address_space(3)))
typedef LDS double ldouble;
typedef LDS short lint;   // working-set arrays in LDS: statuses in {-1,0,1}, indices < 32768 (HBM copies stay int)
typedef LDS unsigned short lidx;   // staged matrix indices: every LDS-resident problem has < 65536 rows / entries
typedef LDS char lchar;

// diagnostic build only (-DRSQP_STAMPS, tools/stamp_small_kernel.py): cycles per phase of block 0
#ifdef RSQP_STAMPS
__device__ unsigned long long g_stamps[48];
#define STAMP(k)                                                                                    \
    do {                                                                                            \
        long long t_ = clock64();                                                                   \
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_stamps[k], (unsigned long long)(t_ - tlast)); \
        tlast = t_;                                                                                 \
    } while (0)
extern "C" void rsqp_debug_stamps(unsigned long long *out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 48);
    if (reset) {
        unsigned long long z[48] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z));
    }
}
#else
#define STAMP(k) do { } while (0)
#endif

namespace {

struct Blocking {
    double tau;
    int kind;  // 0 none, 1 remove constraint, 2 remove bound, 3 add constraint, 4 add bound
    int idx, side;
};

// The target vectors gN / lbN / ubN are only ever read by the lane that owns the entry (and by two uniform-index
// reads in the flipping guard). Builds whose shape is a compile-time constant (nV <= L: one entry per lane) keep
// them in a REGISTER per lane instead of 3 nV doubles of LDS -- which is what brings the hs071-scale image to
// 2368 B = 64 (mod 256): the four problems of a 32-lane LDS access group then sit on disjoint banks.
struct LdsVec {
    ldouble *p;
    __device__ __forceinline__ ldouble &operator[](int i) const { return p[i]; }
    template <int L> __device__ __forceinline__ double bcast(int i) const { return p[i]; }
};
// ---- partner exchange of an all-reduce WITHOUT the LDS crossbar. __shfl_xor compiles to ds_bpermute_b32 (two per
// double, ~100 cycles each and a slot of the LDS pipe the kernel's data also goes through); inside a row of 16 lanes
// a DPP permutation does the same in the VALU. Step S pairs every lane with one that holds the sum of the OTHER
// 2^S-lane block of its 2^(S+1)-lane block: S = 0, 1 quad_perm (xor 1, xor 2), S = 2 row_half_mirror (i <-> 7 - i),
// S = 3 row_mirror (i <-> 15 - i); S = 4, 5 (xor 16, 32) cross rows and stay on ds_bpermute. Steps must run in
// ASCENDING order (the mirrors rely on the blocks below being reduced already); every lane of a block ends with the
// same bits because floating-point addition is commutative.
template <int S> __device__ __forceinline__ int xchg_i32(int x) {
    if constexpr (S == 0) return __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false);       // quad_perm [1,0,3,2]
    else if constexpr (S == 1) return __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
    else if constexpr (S == 2) return __builtin_amdgcn_update_dpp(x, x, 0x141, 0xf, 0xf, false); // row_half_mirror
    else if constexpr (S == 3) return __builtin_amdgcn_update_dpp(x, x, 0x140, 0xf, 0xf, false); // row_mirror
    else return __shfl_xor(x, 1 << S);
}
template <int S> __device__ __forceinline__ double xchg_f64(double x) {
    if constexpr (S >= 4) return __shfl_xor(x, 1 << S);
    else return __hiloint2double(xchg_i32<S>(__double2hiint(x)), xchg_i32<S>(__double2loint(x)));
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// A whole wave (NSTEP == 6): after the four DPP steps every lane holds the sum of its row of 16; the four row sums are
// read through scalar registers (v_readlane) and added as (R0 + R1) + (R2 + R3) -- the value the xor-16 / xor-32
// exchanges produce in every lane (addition is commutative), without their four ds_bpermute round trips.
template <int NSTEP, int S = 0> __device__ __forceinline__ double allreduce_sum(double v) {
    if constexpr (NSTEP == 6 && S == 4) {
        const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
        return (r0 + r1) + (r2 + r3);
    } else if constexpr (S < NSTEP) { v += xchg_f64<S>(v); return allreduce_sum<NSTEP, S + 1>(v); }
    else return v;
}
template <int NSTEP, int S = 0> __device__ __forceinline__ void allreduce_argmin(double &t, int &id) {   // lexicographic min of (t, id)
    if constexpr (NSTEP == 6 && S == 4) {
        double bt = readlane_f64(t, 0);
        int bi = __builtin_amdgcn_readlane(id, 0);
#pragma unroll
        for (int r = 16; r < 64; r += 16) {
            const double t2 = readlane_f64(t, r);
            const int id2 = __builtin_amdgcn_readlane(id, r);
            if (t2 < bt || (t2 == bt && id2 < bi)) { bt = t2; bi = id2; }
        }
        t = bt; id = bi;
    } else if constexpr (S < NSTEP) {
        const double t2 = xchg_f64<S>(t);
        const int id2 = xchg_i32<S>(id);
        if (t2 < t || (t2 == t && id2 < id)) { t = t2; id = id2; }
        allreduce_argmin<NSTEP, S + 1>(t, id);
    }
}
constexpr int ilog2c(int n) { return n <= 1 ? 0 : 1 + ilog2c(n / 2); }

struct RegVec {
    double r;
    __device__ __forceinline__ double &operator[](int) { return r; }                 // index == owning lane by construction
    __device__ __forceinline__ const double &operator[](int) const { return r; }
    template <int L> __device__ __forceinline__ double bcast(int i) const { return __shfl(r, i, L); }   // i uniform over the problem
};
template <bool B> struct MatPtr { typedef const lidx *I; typedef const ldouble *D; };
template <> struct MatPtr<false> { typedef const int *I; typedef const double *D; };

// MAT_LDS: the sparse matrices were staged into LDS behind the image (they fit for every
// hs0xx-scale problem); otherwise they are read from global memory (L2).
template <int L, bool MAT_LDS, bool REGV = false>
struct Engine {
    typedef typename MatPtr<MAT_LDS>::I MI;
    typedef typename MatPtr<MAT_LDS>::D MD;
    // problem
    int nV, nC, ld, sizeT, haveH;
    double hreg;
    MI Ajc, Air; MD Aval;
    MI Arp, Aci; MD Arv;
    MI Hjc, Hir; MD Hval;
    // LDS image
    ldouble *Q, *R, *T;
    ldouble *x, *g, *lb, *ub, *dx, *wq, *wv1, *wv2, *wv3, *wv4, *rc, *rs;
    typedef typename std::conditional<REGV, RegVec, LdsVec>::type TV;
    TV gN, lbN, ubN;
    ldouble *Ax, *lbA, *ubA, *lbAN, *ubAN, *dAx, *wc1, *wc2;
    ldouble *y, *dy;
    lint *Sb, *Sc, *AC, *posAC;
    lint *iscal;    // 8 ints
    static constexpr bool DENSE_MATS = false;
    // uniform over the L lanes of the problem
    int lane;
    int nFR, nAC, status, infeasible, unbounded, nflips;
    long long tlast;

    // ------------------------------------------------------------------ carve
    static constexpr bool K_IMAGE = false;      // (only batches of the explicit-inverse engine are shared with qp_small_g.h)
    // doubles of this formulation's image (<= rsqp_image_doubles, the size of the persistent copy)
    __host__ __device__ static long long image_doubles(int nV, int nC) {
        const long long ld = rsqp_ld(nV), sT = nV < nC ? nV : nC;
        return ld * nV + (long long)nV * (nV + 3) / 2 + sT * ld + (REGV ? 9LL : 12LL) * nV + 8LL * nC + 2LL * (nV + nC);
    }
    __host__ __device__ static long long image_ints(int nV, int nC) { return nV + 3LL * nC + 4; }
    __host__ __device__ static long long factor_doubles(int nV, int nC) {   // Q, R, T: (re)initialised by setup_aux
        const long long ld = rsqp_ld(nV), sT = nV < nC ? nV : nC;
        return ld * nV + (long long)nV * (nV + 3) / 2 + sT * ld;
    }
    // leading part of the image that survives a solve (factors, iterate, auxiliary data, multipliers)
    __host__ __device__ static long long persist_doubles(int nV, int nC) {
        const long long ld = rsqp_ld(nV), sT = nV < nC ? nV : nC;
        return ld * nV + (long long)nV * (nV + 3) / 2 + sT * ld + 4LL * nV + 3LL * nC + (nV + nC);
    }
    __device__ __forceinline__ void carve(lchar *base, int nV_, int nC_) {
        nV = nV_; nC = nC_; ld = rsqp_ld(nV); sizeT = nV < nC ? nV : nC;
        ldouble *p = (ldouble *)base;
        Q = p; p += ld * nV;
        R = p; p += RIX(nV, 0);   // packed upper Hessenberg: column c holds rows 0 .. c + 1 (the sweeps create one sub-diagonal)
        T = p; p += sizeT * ld;
#define CARVE_V(name) name = p; p += nV
#define CARVE_C(name) name = p; p += nC
        // what a hot start needs (persist_doubles, written back to HBM) ...
        CARVE_V(x); CARVE_V(g); CARVE_V(lb); CARVE_V(ub);
        CARVE_C(Ax); CARVE_C(lbA); CARVE_C(ubA);
        y = p; p += nV + nC;
        // ... and the per-solve scratch
        if constexpr (!REGV) { CARVE_V(gN.p); CARVE_V(lbN.p); CARVE_V(ubN.p); }
        CARVE_V(dx); CARVE_V(wq); CARVE_V(wv1); CARVE_V(wv2); CARVE_V(wv3);
        wv4 = wv3;   // the incoming row of an exchange / the staged x0: never alive together with wv3 (Cholesky work vector)
        CARVE_C(lbAN); CARVE_C(ubAN); CARVE_C(dAx); CARVE_C(wc1); CARVE_C(wc2);
#undef CARVE_V
#undef CARVE_C
        dy = p; p += nV + nC;
        // Givens coefficients of a sweep live in dx / dy: the step direction is dead from the homotopy
        // step to the next step_direction(), which is when the working set changes (and in setup_aux,
        // after y0 has been taken out of dy)
        rc = dx; rs = dy;
        lint *ip = (lint *)p;
        Sb = ip; ip += nV;
        Sc = ip; ip += nC;
        AC = ip; ip += nC;
        posAC = ip; ip += nC;
        iscal = ip; ip += 4;
    }

    // ------------------------------------------------------------------ reductions
    // butterflies over the L lanes of this problem (xor offsets < L never leave the group);
    // every lane of the group ends with the same value, so control flow stays group-uniform
    __device__ __forceinline__ double block_sum(double v) { return allreduce_sum<ilog2c(L)>(v); }
    // lexicographic min of (t, id)
    __device__ __forceinline__ void block_argmin(double &t, int &id) { allreduce_argmin<ilog2c(L)>(t, id); }
    __device__ __forceinline__ double dot(const ldouble *a, const ldouble *b, int n) {
        double s = 0.0;
        PFOR(i, n) s += a[i] * b[i];
        return block_sum(s);
    }

    // ------------------------------------------------------------------ sparse products
    // sum_k val[k] * v[idx[k]] over [k0, k1), accumulated in entry order. Four entries per trip:
    // their index / value / gather loads are independent, so the LDS latencies overlap instead of
    // chaining two round trips per entry.
    template <class IP, class DP>
    __device__ __forceinline__ static double sparse_dot(IP idx, DP val, const ldouble *v, int k0, int k1) {
        double s = 0.0;
        int k = k0;
        if constexpr (!MAT_LDS) {
            // matrices in global memory (the image alone nearly fills the LDS): 8 entries per trip, so
            // that 16 L2 round trips are in flight at once instead of 2
            for (; k + 8 <= k1; k += 8) {
                int c[8]; double a[8], w[8];
#pragma unroll
                for (int u = 0; u < 8; u++) { c[u] = idx[k + u]; a[u] = val[k + u]; }
#pragma unroll
                for (int u = 0; u < 8; u++) w[u] = v[c[u]];
#pragma unroll
                for (int u = 0; u < 8; u++) s += a[u] * w[u];
            }
        }
        for (; k + 4 <= k1; k += 4) {
            const int c0 = idx[k], c1 = idx[k + 1], c2 = idx[k + 2], c3 = idx[k + 3];
            const double a0 = val[k], a1 = val[k + 1], a2 = val[k + 2], a3 = val[k + 3];
            const double v0 = v[c0], v1 = v[c1], v2 = v[c2], v3 = v[c3];
            s += a0 * v0; s += a1 * v1; s += a2 * v2; s += a3 * v3;
        }
        if (k + 2 <= k1) {
            const int c0 = idx[k], c1 = idx[k + 1];
            const double a0 = val[k], a1 = val[k + 1];
            const double v0 = v[c0], v1 = v[c1];
            s += a0 * v0; s += a1 * v1;
            k += 2;
        }
        if (k < k1) s += val[k] * v[idx[k]];
        return s;
    }
    __device__ __forceinline__ void A_times(const ldouble *v, ldouble *out) {
        PFOR(r, nC) out[r] = sparse_dot(Aci, Arv, v, Arp[r], Arp[r + 1]);
        SYNC();
    }
    __device__ __forceinline__ void AT_times(const ldouble *yc, ldouble *out) {
        PFOR(c, nV) out[c] = sparse_dot(Air, Aval, yc, Ajc[c], Ajc[c + 1]);
        SYNC();
    }
    __device__ __forceinline__ void H_times(const ldouble *v, ldouble *out) {
        PFOR(c, nV) {
            const double s = haveH ? sparse_dot(Hir, Hval, v, Hjc[c], Hjc[c + 1]) : 0.0;
            out[c] = s + hreg * v[c];
        }
        SYNC();
    }
    // a[v] = A[i][v] for free v, 0 otherwise (all==true: every variable)
    __device__ __forceinline__ void row_of_A(int i, ldouble *a, bool all) {
        PFOR(v, nV) a[v] = 0.0;
        SYNC();
        for (int k = Arp[i] + lane; k < Arp[i + 1]; k += L) {
            int c = Aci[k];
            if (all || Sb[c] == 0) a[c] = Arv[k];
        }
        SYNC();
    }
    // w[c] = Q[:,c] . a   for c < nFR
    __device__ __forceinline__ void QT_times(const ldouble *a, ldouble *w) {
        PFOR(c, nFR) {
            const ldouble *qc = Q + c * ld;
            double s = 0.0;
            for (int v = 0; v < nV; v++) s += qc[v] * a[v];
            w[c] = s;
        }
        SYNC();
    }

    // ------------------------------------------------------------------ Givens helpers
    __device__ __forceinline__ static void givens(double a_elim, double b_keep, double &c, double &s) {
        if (a_elim == 0.0) { c = 1.0; s = 0.0; return; }
        double r = hypot(a_elim, b_keep);
        c = b_keep / r;
        s = a_elim / r;
    }
    // rotations (j, j+1), j = j0 .. j1-1, left to right, coefficients rc/rs[j], applied
    // to the vector w (thread 0 computes them: the chain is sequential)
    __device__ __forceinline__ void plan_sweep(ldouble *w, int j0, int j1, int jskip_below) {
        if (lane == 0) {
            for (int j = j0; j < j1; j++) {
                double c = 1.0, s = 0.0;
                if (j >= jskip_below) givens(w[j], w[j + 1], c, s);
                double a = w[j], b = w[j + 1];
                w[j] = c * a - s * b;
                w[j + 1] = s * a + c * b;
                rc[j] = c; rs[j] = s;
            }
        }
        SYNC();
    }
    // apply the planned sweep to the columns of Q (lane per variable, value carried)
    __device__ __forceinline__ void sweep_Q(int j0, int j1) {
        if (j1 <= j0) return;
        PFOR(v, nV) {
            double carry = Q[j0 * ld + v];
            for (int j = j0; j < j1; j++) {
                double b = Q[(j + 1) * ld + v], c = rc[j], s = rs[j];
                Q[j * ld + v] = c * carry - s * b;
                carry = s * carry + c * b;
            }
            Q[j1 * ld + v] = carry;
        }
        SYNC();
    }
    // same sweep on the columns of T (lane per active row)
    __device__ __forceinline__ void sweep_T(int j0, int j1) {
        if (j1 <= j0) return;
        PFOR(i, nAC) {
            ldouble *row = T + i * ld;
            double carry = row[j0];
            for (int j = j0; j < j1; j++) {
                double b = row[j + 1], c = rc[j], s = rs[j];
                row[j] = c * carry - s * b;
                carry = s * carry + c * b;
            }
            row[j1] = carry;
        }
        SYNC();
    }
    // column sweep (j, j+1), j = 0..nZ-2, on R followed by the row rotations that make
    // it upper triangular again
    __device__ __forceinline__ void sweep_R(int nZ) {
        if (nZ < 2) return;
        PFOR(r, nZ) {
            int j0 = r > 0 ? r - 1 : 0;
            double carry = R[RIX(j0, r)];
            for (int j = j0; j + 1 < nZ; j++) {
                double b = R[RIX(j + 1, r)], c = rc[j], s = rs[j];
                R[RIX(j, r)] = c * carry - s * b;
                carry = s * carry + c * b;
            }
            R[RIX(nZ - 1, r)] = carry;
        }
        SYNC();
        for (int j = 0; j + 1 < nZ; j++) {
            double diag = R[RIX(j, j)], sub = R[RIX(j, j + 1)];
            if (sub != 0.0) {  // uniform: every lane read the same LDS words
                double r = hypot(diag, sub), cc = diag / r, ss = sub / r;
                SYNC();
                for (int col = j + lane; col < nZ; col += L) {
                    ldouble *pc = R + RIX(col, 0);
                    double a = pc[j], b = pc[j + 1];
                    pc[j] = cc * a + ss * b;
                    pc[j + 1] = col == j ? 0.0 : -ss * a + cc * b;
                }
                SYNC();
            }
        }
    }

    // ------------------------------------------------------------------ independence tests
    __device__ __forceinline__ bool constraint_is_LI(int i) {
        int nZ = nFR - nAC;
        if (nZ <= 0) return false;
        row_of_A(i, wv1, false);
        double na2 = dot(wv1, wv1, nV);
        if (na2 == 0.0) return false;
        double s = 0.0;
        PFOR(c, nZ) {
            const ldouble *qc = Q + c * ld;
            double d = 0.0;
            for (int v = 0; v < nV; v++) d += qc[v] * wv1[v];
            s += d * d;
        }
        s = block_sum(s);
        return sqrt(s) > RSQP_EPS_LI * sqrt(na2);
    }
    __device__ __forceinline__ bool bound_is_LI(int v) {
        int nZ = nFR - nAC;
        if (nZ <= 0) return false;
        double s = 0.0;
        PFOR(c, nZ) { double d = Q[c * ld + v]; s += d * d; }
        s = block_sum(s);
        return sqrt(s) > RSQP_EPS_LI;
    }

    // ------------------------------------------------------------------ working-set updates
    __device__ __forceinline__ void add_constraint(int i, int st, bool upd_chol, bool skipZ) {
        int nZ = nFR - nAC;
        row_of_A(i, wv1, false);
        QT_times(wv1, wq);
        int j1 = nZ - 1 > 0 ? nZ - 1 : 0;
        plan_sweep(wq, 0, j1, skipZ ? j1 : 0);
        if (!skipZ) {
            sweep_Q(0, j1);
            if (upd_chol) sweep_R(nZ);
        }
        ldouble *row = T + nAC * ld;
        PFOR(c, nV) row[c] = (c >= nZ - 1 && c < nFR) ? wq[c] : 0.0;
        if (lane == 0) { AC[nAC] = i; posAC[i] = nAC; Sc[i] = st; }
        nAC++;
        SYNC();
    }

    __device__ __forceinline__ void add_bound(int v, int st, bool upd_chol, bool skipZ) {
        int nZ = nFR - nAC;
        PFOR(c, nFR) wq[c] = Q[c * ld + v];
        SYNC();
        int jz = nZ - 1 > 0 ? nZ - 1 : 0;
        plan_sweep(wq, 0, nFR - 1, skipZ ? jz : 0);
        sweep_Q(skipZ ? jz : 0, nFR - 1);
        if (!skipZ && upd_chol) sweep_R(nZ);
        sweep_T(jz, nFR - 1);
        // row v is now +-e_{nFR-1}: drop that row and column
        PFOR(c, nFR) Q[c * ld + v] = 0.0;
        PFOR(u, nV) Q[(nFR - 1) * ld + u] = 0.0;
        PFOR(i, nAC) T[i * ld + nFR - 1] = 0.0;
        if (lane == 0) Sb[v] = st;
        nFR--;
        SYNC();
    }

    // append the Cholesky column of the new null-space column zc; false = not pos. def.
    __device__ __forceinline__ bool chol_append(int zc) {
        const ldouble *z = Q + zc * ld;
        H_times(z, wv2);
        double zHz = dot(z, wv2, nV);
        PFOR(j, zc) {
            const ldouble *qj = Q + j * ld;
            double s = 0.0;
            for (int v = 0; v < nV; v++) s += qj[v] * wv2[v];
            wv3[j] = s;
        }
        SYNC();
        // R' r = rhs, column oriented
        for (int j = 0; j < zc; j++) {
            double rj = wv3[j] / R[RIX(j, j)];
            SYNC();
            if (lane == 0) wv3[j] = rj;
            for (int k = j + 1 + lane; k < zc; k += L) wv3[k] -= R[RIX(k, j)] * rj;
            SYNC();
        }
        double rr = dot(wv3, wv3, zc);
        double rho2 = zHz - rr;
        if (!(rho2 > RSQP_EPS_PD_REL * (fabs(zHz) + rr) + RSQP_EPS_PD_ABS)) return false;
        PFOR(j, zc + 2) R[RIX(zc, j)] = j < zc ? wv3[j] : (j == zc ? sqrt(rho2) : 0.0);   // rows 0 .. zc + 1 of the packed column
        SYNC();
        return true;
    }

    // right-to-left sweep used by the two removals: rotation t acts on columns
    // (cfirst - t, cfirst - t + 1) and is fixed by row (row0 + t) of T
    __device__ __forceinline__ void removal_sweep(int row0, int nrot, int cfirst) {
        for (int t = 0; t < nrot; t++) {
            int i = row0 + t, c0 = cfirst - t;
            ldouble *ri = T + i * ld;
            double c, s;
            givens(ri[c0], ri[c0 + 1], c, s);  // uniform
            SYNC();
            if (lane == 0) { rc[t] = c; rs[t] = s; }
            if (s != 0.0) {
                for (int ii = i + lane; ii < nAC; ii += L) {
                    ldouble *row = T + ii * ld;
                    double a = row[c0], b = row[c0 + 1];
                    row[c0] = ii == i ? 0.0 : c * a - s * b;
                    row[c0 + 1] = s * a + c * b;
                }
            }
            SYNC();
        }
        if (nrot <= 0) return;
        PFOR(v, nV) {
            double keep = Q[(cfirst + 1) * ld + v];
            for (int t = 0; t < nrot; t++) {
                int c0 = cfirst - t;
                double a = Q[c0 * ld + v], c = rc[t], s = rs[t];
                Q[(c0 + 1) * ld + v] = s * a + c * keep;
                keep = c * a - s * keep;
            }
            Q[(cfirst - nrot + 1) * ld + v] = keep;
        }
        SYNC();
    }

    __device__ __forceinline__ int remove_constraint_tq(int k) {
        int cons = AC[k];
        SYNC();
        PFOR(c, nV)
            for (int i = k; i + 1 < nAC; i++) T[i * ld + c] = T[(i + 1) * ld + c];
        if (lane == 0) {
            for (int i = k; i + 1 < nAC; i++) { AC[i] = AC[i + 1]; posAC[AC[i]] = i; }
            posAC[cons] = -1;
            Sc[cons] = 0;
        }
        nAC--;
        SYNC();
        PFOR(c, nV) T[nAC * ld + c] = 0.0;
        SYNC();
        removal_sweep(k, nAC - k, nFR - 2 - k);
        return nFR - nAC - 1;
    }

    __device__ __forceinline__ int remove_bound_tq(int v) {
        int cn = nFR;
        nFR++;
        PFOR(u, nV) Q[cn * ld + u] = u == v ? 1.0 : 0.0;
        PFOR(c, cn) Q[c * ld + v] = 0.0;
        PFOR(i, nAC) T[i * ld + cn] = 0.0;
        if (lane == 0) Sb[v] = 0;
        SYNC();
        for (int k = Ajc[v] + lane; k < Ajc[v + 1]; k += L) {
            int r = Air[k];
            if (Sc[r] != 0) T[posAC[r] * ld + cn] = Aval[k];
        }
        SYNC();
        removal_sweep(0, nAC, cn - 1);
        return nFR - nAC - 1;
    }

    __device__ __forceinline__ bool chol_setup() {
        int nZ = nFR - nAC;
        for (int c = 0; c < nZ; c++)
            if (!chol_append(c)) return false;
        return true;
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

    // warm-start inputs are staged by the caller: x0 in wv4, y0 in dy, guessed bound status in
    // wq and guessed constraint status in wc1 (as doubles); the flags say which are present
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
        for (int k = lane; k < ld * nV; k += L) Q[k] = 0.0;
        for (int k = lane; k < RIX(nV, 0); k += L) R[k] = 0.0;
        for (int k = lane; k < sizeT * ld; k += L) T[k] = 0.0;
        PFOR(i, nC) { Sc[i] = 0; posAC[i] = -1; }
        SYNC();
        if (lane == 0) {
            int n = 0;
            for (int v = 0; v < nV; v++)
                if (Sb[v] == 0) Q[(n++) * ld + v] = 1.0;
            iscal[0] = n;
        }
        SYNC();
        nFR = iscal[0];
        nAC = 0;
        // products with x = 0 / y = 0 (a cold start) are zero vectors: no need to walk the matrices
        if (x0) A_times(x, Ax); else { PFOR(i, nC) Ax[i] = 0.0; SYNC(); }
        for (int i = 0; i < nC; i++) {
            int s = 0;
            if (guess_c) s = (int)wc1[i];
            else if (y0 && (!x0 || cy0)) s = y[nV + i] > RSQP_EPS ? -1 : (y[nV + i] < -RSQP_EPS ? 1 : 0);
            else if (x0) s = Ax[i] <= lbAN[i] + RSQP_BOUND_TOLERANCE ? -1 : (Ax[i] >= ubAN[i] - RSQP_BOUND_TOLERANCE ? 1 : 0);
            if (s == -1 && lbAN[i] <= -RSQP_INFTY) s = 0;
            if (s == 1 && ubAN[i] >= RSQP_INFTY) s = 0;
            if (s != 0 && constraint_is_LI(i)) add_constraint(i, s, false, false);
        }
        PFOR(v, nV) {
            double yv = y[v];
            if (Sb[v] == 0 || (Sb[v] == -1 && yv < 0.0) || (Sb[v] == 1 && yv > 0.0)) y[v] = 0.0;
        }
        PFOR(i, nC) {
            double yi = y[nV + i];
            if (Sc[i] == 0 || (Sc[i] == -1 && yi < 0.0) || (Sc[i] == 1 && yi > 0.0)) y[nV + i] = 0.0;
        }
        SYNC();
        if (y0) AT_times(y + nV, wv1); else { PFOR(v, nV) wv1[v] = 0.0; }
        if (x0) H_times(x, wv2); else { PFOR(v, nV) wv2[v] = 0.0; }
        SYNC();
        PFOR(v, nV) {
            double xv = x[v];
            g[v] = wv1[v] + y[v] - wv2[v];
            lb[v] = Sb[v] == -1 ? xv : fmin(lbN[v], xv - RSQP_BOUND_RELAXATION);
            ub[v] = Sb[v] == 1 ? xv : fmax(ubN[v], xv + RSQP_BOUND_RELAXATION);
        }
        PFOR(i, nC) {
            double ax = Ax[i];
            lbA[i] = Sc[i] == -1 ? ax : fmin(lbAN[i], ax - RSQP_BOUND_RELAXATION);
            ubA[i] = Sc[i] == 1 ? ax : fmax(ubAN[i], ax + RSQP_BOUND_RELAXATION);
        }
        SYNC();
        if (!chol_setup()) return RET_SETUP_FAILED;
        status = QPS_AUXILIARYQPSOLVED;
        return RET_OK;
    }

    // ------------------------------------------------------------------ step direction
    __device__ __forceinline__ static double delta_of(double target, double cur) {
        return (fabs(target) >= RSQP_INFTY && fabs(cur) >= RSQP_INFTY) ? 0.0 : target - cur;
    }

    __device__ __forceinline__ void step_direction() {
        int nZ = nFR - nAC;
        PFOR(v, nV) dx[v] = Sb[v] == -1 ? delta_of(lbN[v], lb[v]) : (Sb[v] == 1 ? delta_of(ubN[v], ub[v]) : 0.0);
        PFOR(i, nV + nC) dy[i] = 0.0;
        SYNC();
        A_times(dx, wc2);
        // without a null space (nZ == 0: every free direction is pinned by an active constraint -- the
        // whole cold-start phase of hs0xx-scale problems) the projected-gradient part below is empty:
        // its two Hessian products are skipped, nothing else depends on them
        if (nZ > 0) H_times(dx, wv2);
        PFOR(i, nAC) {
            int r = AC[i];
            wc1[i] = (Sc[r] == -1 ? delta_of(lbAN[r], lbA[r]) : delta_of(ubAN[r], ubA[r])) - wc2[r];
        }
        if (nZ > 0) { PFOR(v, nV) wv1[v] = (gN[v] - g[v]) + wv2[v]; }  // tmpg
        PFOR(c, nFR) wq[c] = 0.0;
        SYNC();
        // range space: T wY = bA (column oriented)
        for (int i = 0; i < nAC; i++) {
            int c = nFR - 1 - i;
            double w = wc1[i] / T[i * ld + c];
            SYNC();
            if (lane == 0) wq[c] = w;
            for (int ii = i + 1 + lane; ii < nAC; ii += L) wc1[ii] -= T[ii * ld + c] * w;
            SYNC();
        }
        PFOR(v, nV) {
            double s = 0.0;
            for (int c = nZ; c < nFR; c++) s += Q[c * ld + v] * wq[c];
            wv3[v] = s;  // xY
        }
        SYNC();
        // null space: R'R wZ = -Z'(tmpg + H xY)
        if (nZ > 0) {
            H_times(wv3, wv2);
            PFOR(v, nV) wv2[v] += wv1[v];
            SYNC();
        }
        PFOR(j, nZ) {
            const ldouble *qj = Q + j * ld;
            double s = 0.0;
            for (int v = 0; v < nV; v++) s += qj[v] * wv2[v];
            wq[j] = -s;
        }
        SYNC();
        for (int j = 0; j < nZ; j++) {
            double u = wq[j] / R[RIX(j, j)];
            SYNC();
            if (lane == 0) wq[j] = u;
            for (int k = j + 1 + lane; k < nZ; k += L) wq[k] -= R[RIX(k, j)] * u;
            SYNC();
        }
        for (int j = nZ - 1; j >= 0; j--) {
            double w = wq[j] / R[RIX(j, j)];
            SYNC();
            if (lane == 0) wq[j] = w;
            for (int k = lane; k < j; k += L) wq[k] -= R[RIX(j, k)] * w;
            SYNC();
        }
        PFOR(v, nV) {
            if (Sb[v] == 0) {
                double s = wv3[v];
                for (int j = 0; j < nZ; j++) s += Q[j * ld + v] * wq[j];
                dx[v] = s;
            }
        }
        SYNC();
        // multipliers of the active constraints: T' dyAC = Y'(H dx + dg)
        H_times(dx, wv2);
        PFOR(v, nV) wv2[v] += gN[v] - g[v];
        SYNC();
        for (int c = nZ + lane; c < nFR; c += L) {
            const ldouble *qc = Q + c * ld;
            double s = 0.0;
            for (int v = 0; v < nV; v++) s += qc[v] * wv2[v];
            wq[c] = s;
        }
        SYNC();
        for (int m = 0; m < nAC; m++) {
            int i = nAC - 1 - m, c = nZ + m;
            const ldouble *ri = T + i * ld;
            double d = wq[c] / ri[c];
            SYNC();
            if (lane == 0) dy[nV + AC[i]] = d;
            for (int cc = c + 1 + lane; cc < nFR; cc += L) wq[cc] -= ri[cc] * d;
            SYNC();
        }
        if (nAC > 0) AT_times(dy + nV, wv3); else { PFOR(v, nV) wv3[v] = 0.0; SYNC(); }   // no active constraint: dy_C = 0
        PFOR(v, nV) dy[v] = Sb[v] != 0 ? wv2[v] - wv3[v] : 0.0;
        A_times(dx, dAx);
    }

    // ------------------------------------------------------------------ ratio tests
    __device__ __forceinline__ static void cand(double num, double den, int id, double &bt, int &bid) {
        // the quotient does not wait for the comparison with the running minimum (two candidates of a
        // lane divide back to back); a rejected denominator only wastes a division
        const double t = (num > 0.0 ? num : 0.0) / den;
        if (den >= RSQP_EPS_DEN && (t < bt || (t == bt && id < bid))) { bt = t; bid = id; }
    }
    // candidate ids: [0,nC) active constr. duals, [nC,nC+nV) fixed-variable duals,
    // then inactive constr. lower / upper, then free variables lower / upper
    __device__ __forceinline__ Blocking ratio_tests() {
        double bt = 1.0;
        int bid = 0x7fffffff;
        PFOR(i, nC) {
            double Axi = Ax[i], dA = dAx[i];
            if (Sc[i] != 0) {
                double yi = y[nV + i], d = dy[nV + i];
                if (Sc[i] == -1) cand(yi, -d, i, bt, bid); else cand(-yi, d, i, bt, bid);
            } else {
                if (lbAN[i] > -RSQP_INFTY) cand(Axi - lbA[i], delta_of(lbAN[i], lbA[i]) - dA, nC + nV + i, bt, bid);
                if (ubAN[i] < RSQP_INFTY) cand(ubA[i] - Axi, dA - delta_of(ubAN[i], ubA[i]), 2 * nC + nV + i, bt, bid);
            }
        }
        PFOR(v, nV) {
            if (Sb[v] != 0) {
                double yi = y[v], d = dy[v];
                if (Sb[v] == -1) cand(yi, -d, nC + v, bt, bid); else cand(-yi, d, nC + v, bt, bid);
            } else {
                if (lbN[v] > -RSQP_INFTY) cand(x[v] - lb[v], delta_of(lbN[v], lb[v]) - dx[v], 3 * nC + nV + v, bt, bid);
                if (ubN[v] < RSQP_INFTY) cand(ub[v] - x[v], dx[v] - delta_of(ubN[v], ub[v]), 3 * nC + 2 * nV + v, bt, bid);
            }
        }
        // a candidate only blocks if it is strictly inside the step (t < 1)
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

    // ------------------------------------------------------------------ removal with guard
    __device__ __forceinline__ int remove_with_guard(bool is_bound, int idx) {
        if (is_bound) {
            int old = Sb[idx];
            SYNC();
            int zc = remove_bound_tq(idx);
            if (lane == 0) y[idx] = 0.0;
            SYNC();
            if (chol_append(zc)) return RET_OK;
            if ((old == -1 && ubN.template bcast<L>(idx) >= RSQP_INFTY) || (old == 1 && lbN.template bcast<L>(idx) <= -RSQP_INFTY)) {
                add_bound(idx, old, false, true);
                return RET_UNBOUNDED;
            }
            add_bound(idx, -old, false, true);
            if (lane == 0) { if (old == -1) ub[idx] = x[idx]; else lb[idx] = x[idx]; }
            nflips++;
            SYNC();
            return RET_OK;
        } else {
            int old = Sc[idx], k = posAC[idx];
            SYNC();
            int zc = remove_constraint_tq(k);
            if (lane == 0) y[nV + idx] = 0.0;
            SYNC();
            if (chol_append(zc)) return RET_OK;
            if ((old == -1 && ubAN[idx] >= RSQP_INFTY) || (old == 1 && lbAN[idx] <= -RSQP_INFTY)) {
                add_constraint(idx, old, false, true);
                return RET_UNBOUNDED;
            }
            add_constraint(idx, -old, false, true);
            if (lane == 0) { if (old == -1) ubA[idx] = Ax[idx]; else lbA[idx] = Ax[idx]; }
            nflips++;
            SYNC();
            return RET_OK;
        }
    }

    // ------------------------------------------------------------------ exchange
    // a_full in wv4. Shifts the multipliers; returns partner in (pkind, pidx)
    __device__ __forceinline__ int ensure_LI(int side, double &y_new, int &pkind, int &pidx) {
        int nZ = nFR - nAC;
        PFOR(v, nV) wv1[v] = Sb[v] == 0 ? wv4[v] : 0.0;
        PFOR(i, nC) wc2[i] = 0.0;
        SYNC();
        QT_times(wv1, wq);
        for (int m = 0; m < nAC; m++) {
            int i = nAC - 1 - m, c = nZ + m;
            const ldouble *ri = T + i * ld;
            double d = wq[c] / ri[c];
            SYNC();
            if (lane == 0) wc2[AC[i]] = d;
            for (int cc = c + 1 + lane; cc < nFR; cc += L) wq[cc] -= ri[cc] * d;
            SYNC();
        }
        if (nAC > 0) AT_times(wc2, wv2); else { PFOR(v, nV) wv2[v] = 0.0; SYNC(); }   // xi_C = 0 without active constraints
        PFOR(v, nV) wv2[v] = Sb[v] != 0 ? wv4[v] - wv2[v] : 0.0;  // xiB
        SYNC();
        double sgn = side == 1 ? -1.0 : 1.0;
        double bt = RSQP_INFTY;
        int bid = 0x7fffffff;
        PFOR(i, nC) {
            if (Sc[i] != 0) {
                double xi = sgn * wc2[i], yi = y[nV + i];
                double num = Sc[i] == -1 ? yi : -yi, den = Sc[i] == -1 ? xi : -xi;
                if (den > RSQP_EPS_DEN) {
                    double t = (num > 0.0 ? num : 0.0) / den;
                    if (t < bt || (t == bt && i < bid)) { bt = t; bid = i; }
                }
            }
        }
        PFOR(v, nV) {
            if (Sb[v] != 0) {
                double xi = sgn * wv2[v], yi = y[v];
                double num = Sb[v] == -1 ? yi : -yi, den = Sb[v] == -1 ? xi : -xi;
                if (den > RSQP_EPS_DEN) {
                    double t = (num > 0.0 ? num : 0.0) / den;
                    if (t < bt || (t == bt && nC + v < bid)) { bt = t; bid = nC + v; }
                }
            }
        }
        block_argmin(bt, bid);
        if (bid == 0x7fffffff) return RET_INFEASIBLE;
        PFOR(i, nC) if (Sc[i] != 0) y[nV + i] -= bt * sgn * wc2[i];
        PFOR(v, nV) if (Sb[v] != 0) y[v] -= bt * sgn * wv2[v];
        SYNC();
        y_new = sgn * bt;
        pkind = bid < nC ? 1 : 2;
        pidx = bid < nC ? bid : bid - nC;
        return RET_OK;
    }

    __device__ __forceinline__ bool remove_partner(int pkind, int pidx) {
        int zc;
        if (pkind == 1) {
            int k = posAC[pidx];
            SYNC();
            zc = remove_constraint_tq(k);
            if (lane == 0) y[nV + pidx] = 0.0;
        } else {
            zc = remove_bound_tq(pidx);
            if (lane == 0) y[pidx] = 0.0;
        }
        SYNC();
        return chol_append(zc);
    }

    __device__ __forceinline__ int change_active_set(const Blocking &b) {
        if (b.kind == 1) return remove_with_guard(false, b.idx);
        if (b.kind == 2) return remove_with_guard(true, b.idx);
        if (b.kind == 3 || b.kind == 4) {
            double ynew = 0.0;
            bool full = true;
            bool li = b.kind == 3 ? constraint_is_LI(b.idx) : bound_is_LI(b.idx);
            if (!li) {
                int pkind = 0, pidx = -1;
                if (b.kind == 3) row_of_A(b.idx, wv4, true);
                else { PFOR(v, nV) wv4[v] = v == b.idx ? 1.0 : 0.0; SYNC(); }
                int rc_ = ensure_LI(b.side, ynew, pkind, pidx);
                if (rc_ != RET_OK) return rc_;
                full = remove_partner(pkind, pidx);
            }
            if (b.kind == 3) {
                add_constraint(b.idx, b.side, full, !full);
                if (lane == 0) y[nV + b.idx] = ynew;
            } else {
                add_bound(b.idx, b.side, full, !full);
                if (lane == 0) y[b.idx] = ynew;
            }
            SYNC();
        }
        return RET_OK;
    }

    // ------------------------------------------------------------------ homotopy
    __device__ __forceinline__ void drift_correction() {
        PFOR(v, nV) if (Sb[v] != 0) x[v] = Sb[v] == -1 ? lb[v] : ub[v];
        SYNC();
        A_times(x, Ax);
        PFOR(i, nC) { if (Sc[i] == -1) lbA[i] = Ax[i]; else if (Sc[i] == 1) ubA[i] = Ax[i]; }
        PFOR(v, nV) {   // A'y_C and H x of variable v in one pass, then the gradient from stationarity
            const double aty = sparse_dot(Air, Aval, y + nV, Ajc[v], Ajc[v + 1]);
            const double hx = (haveH ? sparse_dot(Hir, Hval, x, Hjc[v], Hjc[v + 1]) : 0.0) + hreg * x[v];
            g[v] = aty + y[v] - hx;
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
            PFOR(i, nC) {
                if (done) { lbA[i] = lbAN[i]; ubA[i] = ubAN[i]; }
                else {
                    lbA[i] += tau * delta_of(lbAN[i], lbA[i]); ubA[i] += tau * delta_of(ubAN[i], ubA[i]);
                    // A x follows the step by its increment (dAx, a by-product of the step direction). Until the exact
                    // product in drift_correction only Ax[blocking] / Ax[flipped] are read, and only into bounds that
                    // drift_correction overwrites with the exact product once the constraint is active
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

    __device__ __forceinline__ bool bounds_inconsistent() {
        double bad = 0.0;
        PFOR(v, nV) if (lbN[v] > ubN[v] + RSQP_EPS) bad += 1.0;
        PFOR(i, nC) if (lbAN[i] > ubAN[i] + RSQP_EPS) bad += 1.0;
        return block_sum(bad) > 0.0;
    }
    __device__ __forceinline__ void restore(int nFR_, int nAC_, int status_) { nFR = nFR_; nAC = nAC_; status = status_; }

    __device__ __forceinline__ double objective() {
        H_times(x, wv2);
        double bs = 0.0;
        PFOR(i, nV) bs += gN[i] * x[i];
        double a = dot(x, wv2, nV), b = block_sum(bs), c = dot(x, x, nV);
        return 0.5 * (a - hreg * c) + b;
    }
};

#include "qp_small_x.h"

// ------------------------------------------------------------------------------------
// bytes of LDS needed to stage the sparse matrices of one problem behind its image
__host__ __device__ inline long long mat_lds_bytes(int nV, int nC, int annz, int hnnz) {
    long long idx = 2LL * (nV + 1) + (nC + 1) + 2LL * annz + hnnz, dbl = 2LL * annz + hnnz;   // 16-bit indices
    return ((idx * 2 + 7) & ~7LL) + dbl * 8;
}

// L = lanes per problem (64 / L problems share one wave; each owns `stride` bytes of LDS),
// W = minimum waves per SIMD the register allocator has to leave room for
// SHAPE = NVC * 256 + NCC > 0: every problem of the batch has the shape NVC x NCC, known at COMPILE time
// (parameter scans / the hs071-scale batch: 8 x 2 through the QPhandler formulation). Sizes, loop bounds and
// the offsets of the LDS image are then constants: every vector of the engine sits at an immediate offset of
// ONE per-lane base address instead of in a register of its own, and every loop over a vector is straight-line.
// (Passing the uniform shape as kernel ARGUMENTS instead -- wave-uniform scalars -- measured 221 VGPRs instead
// of 256 but 214 vs 226 M solves/s on 65 536 hs071-scale QPs: not built.)
template <class ENG, int L, bool MAT_LDS, int W, int SHAPE>
__global__ void __launch_bounds__(L > 64 ? L : 64, W)
small_qp_kernel(QPPools P, int nq, int stride, int mode, int maxWSR) {
    constexpr int NVC = SHAPE >> 8, NCC = SHAPE & 255;
    extern __shared__ __attribute__((aligned(16))) char smem_generic[];
    const int lane = L >= 64 ? (int)threadIdx.x : (int)threadIdx.x & (L - 1);
    const int grp = L >= 64 ? 0 : (int)threadIdx.x / L;  // L >= 64: everything below stays workgroup-uniform
    const int q = L >= 64 ? (int)blockIdx.x : blockIdx.x * (64 / L) + grp;
    if (q >= nq) return;  // no workgroup barrier anywhere below: idle groups may leave
    if (P.only_bailed && P.ret[q] != RET_BAIL) return;   // second pass behind the tableau kernel (qp_small_g.h)
    lchar *smem = (lchar *)smem_generic + grp * stride;
    QPDesc d = P.desc[q];
    if constexpr (SHAPE > 0) { d.nV = NVC; d.nC = NCC; }
    if constexpr (L < 64) {
        // packed waves are only launched when every problem of the batch has nV, nC <= L: a loop over a
        // vector of the engine is then a single predicated trip (no back edge, no counter)
        __builtin_assume(d.nV <= L && d.nV >= 0);
        __builtin_assume(d.nC <= L && d.nC >= 0);
    }
    ENG E;
    E.lane = lane;
    if constexpr (L > 64) E.part = (ldouble *)(smem + stride) - L;   // the launcher reserves 8 L bytes at the end
#ifdef RSQP_STAMPS
    E.tlast = clock64();
    long long &tlast = E.tlast;
#endif
    E.carve(smem, d.nV, d.nC);
    const int nd = (int)ENG::image_doubles(d.nV, d.nC), ni = (int)ENG::image_ints(d.nV, d.nC);
    const int np = (int)ENG::persist_doubles(d.nV, d.nC);   // what goes to / comes from HBM: [np doubles][ni ints]
    const int img_bytes = (nd * 8 + ni * 2 + 7) & ~7;   // the staged matrices follow 8-byte aligned
    E.haveH = d.haveH;
    E.hreg = d.hreg;
    const int *gAjc = P.Ajc + d.offAjc, *gAir = P.Air + d.offAnz, *gArp = P.Arp + d.offArp, *gAci = P.Aci + d.offAnz;
    const int *gHjc = P.Hjc + d.offHjc, *gHir = P.Hir + d.offHnz;
    const double *gAval = P.Aval + d.offAnz, *gArv = P.Arv + d.offAnz, *gHval = P.Hval + d.offHnz;
    if constexpr (ENG::DENSE_MATS) {
        E.stage_dense(smem + img_bytes, gAjc, gAir, gAval, gHjc, gHir, gHval);
    } else if constexpr (MAT_LDS) {
        // stage CSC(A), CSR(A), CSC(H) behind the image
        const int annz = d.annz >= 0 ? d.annz : gAjc[d.nV], hnnz = !d.haveH ? 0 : (d.hnnz >= 0 ? d.hnnz : gHjc[d.nV]);
        LDS unsigned short *ip0 = (LDS unsigned short *)(smem + img_bytes), *ip = ip0;
        LDS unsigned short *lAjc = ip; ip += d.nV + 1;
        LDS unsigned short *lArp = ip; ip += d.nC + 1;
        LDS unsigned short *lHjc = ip; ip += d.nV + 1;
        LDS unsigned short *lAir = ip; ip += annz;
        LDS unsigned short *lAci = ip; ip += annz;
        LDS unsigned short *lHir = ip; ip += hnnz;
        ldouble *dp = (ldouble *)(smem + img_bytes + (((ip - ip0) * 2 + 7) & ~7));
        ldouble *lAval = dp; dp += annz;
        ldouble *lArv = dp; dp += annz;
        ldouble *lHval = dp;
        for (int k = lane; k <= d.nV; k += L) { lAjc[k] = gAjc[k]; lHjc[k] = d.haveH ? gHjc[k] : 0; }
        for (int k = lane; k <= d.nC; k += L) lArp[k] = gArp[k];
        for (int k = lane; k < annz; k += L) { lAir[k] = gAir[k]; lAci[k] = gAci[k]; lAval[k] = gAval[k]; lArv[k] = gArv[k]; }
        for (int k = lane; k < hnnz; k += L) { lHir[k] = gHir[k]; lHval[k] = gHval[k]; }
        E.Ajc = lAjc; E.Air = lAir; E.Aval = lAval; E.Arp = lArp; E.Aci = lAci; E.Arv = lArv;
        E.Hjc = lHjc; E.Hir = lHir; E.Hval = lHval;
    } else {
        E.Ajc = gAjc; E.Air = gAir; E.Aval = gAval; E.Arp = gArp; E.Aci = gAci; E.Arv = gArv;
        E.Hjc = gHjc; E.Hir = gHir; E.Hval = gHval;
    }
    E.nflips = 0; E.infeasible = E.unbounded = 0; E.status = QPS_NOTINITIALISED; E.nFR = E.nAC = 0;
    double *img = P.state + d.offState;
    int *iimg = reinterpret_cast<int *>(img + np);
    ldouble *simg = (ldouble *)smem;
    lint *siimg = (lint *)(simg + nd);

    int rcode = RET_OK, nWSR = 0;
    if (mode == 0) {
        // (the factor arrays at the head of the image are zeroed by setup_aux itself)
        for (int k = (int)ENG::factor_doubles(d.nV, d.nC) + lane; k < nd; k += L) simg[k] = 0.0;
        for (int k = lane; k < ni; k += L) siimg[k] = 0;
        SYNC();
    }
    if (mode != 0) {  // reload the image of the previous solve
        for (int k = lane; k < np; k += L) simg[k] = img[k];
        for (int k = np + lane; k < nd; k += L) simg[k] = 0.0;
        for (int k = lane; k < ni; k += L) siimg[k] = iimg[k];
        SYNC();
        E.restore(E.iscal[1], E.iscal[2], E.iscal[3]);
        SYNC();
        if (E.status == QPS_NOTINITIALISED) mode = 0;
    }
    STAMP(0);
    E.store_targets(P.g + d.offV, P.lb + d.offV, P.ub + d.offV, P.lbA + d.offC, P.ubA + d.offC);
    if (E.bounds_inconsistent()) {  // qpOASES areBoundsConsistent: infeasible before any change
        E.infeasible = 1; E.unbounded = 0;
        rcode = RET_INFEASIBLE;
    } else if (mode == 0) {
        rcode = E.setup_aux(false, false, false, false);
    } else if (mode == 2) {  // hot start with new matrices: keep x, y and the working set
        for (int v = lane; v < d.nV; v += L) { E.wv4[v] = E.x[v]; E.wq[v] = (double)E.Sb[v]; }
        for (int i = lane; i < d.nV + d.nC; i += L) E.dy[i] = E.y[i];
        for (int i = lane; i < d.nC; i += L) E.wc1[i] = (double)E.Sc[i];
        SYNC();
        rcode = E.setup_aux(true, true, true, true);
        if (rcode != RET_OK) rcode = E.setup_aux(false, false, false, false);
    } else if (mode == 3) {  // warm re-initialisation from (x0, y0, guessed bounds)
        if (P.x0) for (int v = lane; v < d.nV; v += L) E.wv4[v] = P.x0[d.offV + v];
        if (P.y0) for (int i = lane; i < d.nV + d.nC; i += L) E.dy[i] = P.y0[d.offV + d.offC + i];
        if (P.guess_b) for (int v = lane; v < d.nV; v += L) E.wq[v] = (double)P.guess_b[d.offV + v];
        SYNC();
        // no guessed constraints in this call shape (qpOASESInterface.cpp:204-206): their sides come from the signs of
        // A x0 as qpOASES does (the default) -- or, opt-in (P.reinit_from_y0), from the signs of y0
        rcode = E.setup_aux(P.x0 != nullptr, P.y0 != nullptr, P.guess_b != nullptr, false, P.reinit_from_y0 != 0);
        if (rcode != RET_OK) rcode = E.setup_aux(false, false, false, false);
    } else {
        E.infeasible = E.unbounded = 0;
        if constexpr (ENG::K_IMAGE) {
            // the state is one the KKT-tableau kernel wrote (and then bailed out of this hot start): its factors
            // are not this engine's -- rebuild them for the stored working set, keep the homotopy data
            if (E.iscal[4] != 0) {     // (1: round 3, M = K^-1; 2: the tableau of qp_small_g.h -- either way not this engine's factors)
                rcode = E.rebuild_factors();
                if (rcode != RET_OK) rcode = E.setup_aux(false, false, false, false);
            }
        }
    }
    STAMP(2);
    if (rcode == RET_OK) rcode = E.homotopy(maxWSR, nWSR);
    double obj = E.objective();
    STAMP(8);

    // results
    for (int v = lane; v < d.nV; v += L) { P.x[d.offV + v] = E.x[v]; P.ws_b[d.offV + v] = E.Sb[v]; }
    for (int i = lane; i < d.nV + d.nC; i += L) P.y[d.offV + d.offC + i] = E.y[i];
    for (int i = lane; i < d.nC; i += L) P.ws_c[d.offC + i] = E.Sc[i];
    if (lane == 0) {
        int st = E.status;
        P.status[q] = E.infeasible ? 100 + st : (E.unbounded ? 200 + st : st);
        P.ret[q] = rcode;
        P.nwsr[q] = nWSR;
        P.nflips[q] = E.nflips;
        P.obj[q] = obj;
        E.iscal[1] = E.nFR; E.iscal[2] = E.nAC; E.iscal[3] = E.status;
        if constexpr (ENG::K_IMAGE) E.iscal[4] = 0;      // this engine's factors
    }
    if (P.done_flag) __threadfence_system();      // the results above are in host-mapped memory: visible before the flag
    SYNC();
    if (P.done_flag && q == 0 && lane == 0) *reinterpret_cast<volatile int *>(P.done_flag) = P.done_val;
    if (P.keep_state) {
        for (int k = lane; k < np; k += L) img[k] = simg[k];
        for (int k = lane; k < ni; k += L) iimg[k] = siimg[k];
    } else if (lane == 0) {
        iimg[(int)(E.iscal - siimg) + 3] = QPS_NOTINITIALISED;
    }
    STAMP(9);
}

#include "qp_small_g.h"

}  // namespace

static const long long kMaxLds = 160 * 1024;

static long long align16(long long v) { return (v + 15) & ~15LL; }

int rsqp_small_qp_fits(int nVmax, int nCmax) {
    return align16(rsqp_image_bytes(nVmax, nCmax)) <= kMaxLds;
}

static int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

SmallKnobs rsqp_small_knobs_from_env() {
    SmallKnobs k;
    k.engine = env_int("RSQP_SMALL_ENGINE", -1); k.k_debug_bail = env_int("RSQP_K_DEBUG_BAIL", -1); k.noshape = env_int("RSQP_SMALL_NOSHAPE", 0);
    k.lanes = env_int("RSQP_SMALL_LANES", -1); k.waves = env_int("RSQP_SMALL_WAVES", -1); k.wide = env_int("RSQP_SMALL_WIDE", -1);
    k.wide_lanes = env_int("RSQP_SMALL_WIDE_LANES", 256) == 512 ? 512 : 256; k.nospread = env_int("RSQP_SMALL_NOSPREAD", 0);
    k.no_kkt = env_int("RSQP_SMALL_NO_KKT", 0); k.kkt_only = env_int("RSQP_SMALL_KKT_ONLY", 0); k.no_tiny = env_int("RSQP_SMALL_NO_TINY", 0);
    k.tiny_lds = env_int("RSQP_TINY_LDS", 0); k.lane = env_int("RSQP_LANE", -1); k.exp_matglobal = env_int("RSQP_EXP_MATGLOBAL", 0);
    k.arena_mapped = env_int("RSQP_ARENA_MAPPED", -1);
    k.no_spin = getenv("RSQP_NO_SPIN") != nullptr; k.no_spec_cert = getenv("RSQP_NO_SPEC_CERT") != nullptr;
    return k;
}
int rsqp_small_launch_is_tiny(const SmallKnobs &kn, const QPPools &p, int nVmax, int nCmax) {
    return (kn.engine < 0 && p.tiny_ok && rsqp_tiny_fits(kn, nVmax, nCmax)) ? 1 : 0;
}
hipError_t rsqp_launch_small_qp(const SmallKnobs &kn, const QPPools &p_in, int nq, int nVmax, int nCmax, long long mat_bytes_max, int mode,
                                int maxWSR, hipStream_t stream) {
    QPPools p = p_in;
    p.only_bailed = 0;
    p.k_debug_bail = kn.k_debug_bail;
    if (nq <= 0) return hipSuccess;
    if (align16(rsqp_image_bytes(nVmax, nCmax)) > kMaxLds) return hipErrorInvalidValue;
    // formulation: 0 = Givens / TQ (Engine), 1 = explicit inverses (EngineX, qp_small_x.h), which keeps
    // DENSE copies of A and H in LDS. Measured per shape on the 512-QP hs0xx batch (ms, TQ vs explicit):
    // 5x1 0.14 / 0.16, 8x2 0.045 / 0.051, 8x3 0.25 / 0.21, 12x4 0.49 / 0.42, 16x6 0.73 / 0.55,
    // 23x6 1.26 / 1.01, 37x14 2.57 / 1.51, 69x28 10.8 / 4.1 -- the chains of the TQ form grow with nZ.
    const int forcedE = kn.engine;
    // hs071-scale problems: the register-resident tableau kernel (qp_tiny.hip) serves every call shape
    if (rsqp_small_launch_is_tiny(kn, p, nVmax, nCmax) && rsqp_lane_fits(kn, p, nq, nVmax, nCmax, mode)) return rsqp_launch_lane_qp(p, nq, maxWSR, stream);
    if (rsqp_small_launch_is_tiny(kn, p, nVmax, nCmax)) return rsqp_launch_tiny_qp(kn, p, nq, nVmax, nCmax, mode, maxWSR, stream);
    const int eng = forcedE == 0 || forcedE == 1 ? forcedE : (nVmax > 8 ? 1 : 0);
    if (eng == 1 && mat_bytes_max >= 0) mat_bytes_max = 8LL * ((long long)nVmax * nVmax + (long long)nCmax * nVmax);
    // uniform hs071-scale batches (8 x 2 through the QPhandler formulation; parameter scans of one NLP iterate) run
    // the build with the shape as a compile-time constant and the target vectors in registers (see RegVec)
    const int noshape = kn.noshape, forcedL0 = kn.lanes;
    const bool shape82 = eng == 0 && p.uniV == 8 && p.uniC == 2 && !noshape && mat_bytes_max >= 0 &&
                         (forcedL0 < 0 || forcedL0 == 8);        // only the 8-lane build has the shape instantiation
    // LDS image of the chosen formulation (the persistent copy in HBM is sized for the larger one)
    const long long imgd = eng == 1 ? EngineX<64, true>::image_doubles(nVmax, nCmax)
                                    : (shape82 ? Engine<8, true, true>::image_doubles(nVmax, nCmax) : Engine<64, true>::image_doubles(nVmax, nCmax));
    const long long imgi = eng == 1 ? EngineX<64, true>::image_ints(nVmax, nCmax) : Engine<64, true>::image_ints(nVmax, nCmax);
    const long long img = (8 * imgd + 2 * imgi + 7) & ~7LL;
#if defined(RSQP_SMALL_EXPERIMENT) && RSQP_SMALL_EXPERIMENT == 1
    // tuning build only: the 8-lane shape kernel with the matrices left in global memory (L2) -- a smaller LDS image per problem,
    // more resident waves (RSQP_EXP_MATGLOBAL=1 with RSQP_SMALL_WAVES=3)
    const int exp_nomat = kn.exp_matglobal;
#else
    constexpr int exp_nomat = 0;
#endif
    const bool mat_lds = mat_bytes_max >= 0 && align16(img + mat_bytes_max) <= kMaxLds && !exp_nomat;
    // LDS of one problem: image, then its staged matrices, 16-byte granular.
    long long stride = align16(img + (mat_lds ? mat_bytes_max : 0));
    // (an odd number of 16-byte units would spread the problems of a wave over the banks, but the LDS is
    // allocated in 512-byte steps and the 8-lane build needs 8 x 2880 = 45 x 512 bytes for 7 workgroups per CU)
    // lanes per problem: the vectors of the engine have nV (+ nC) entries, a wave of 64 lanes is
    // mostly idle on hs0xx-scale problems, so 64 / L of them share a wave. Problems in one wave
    // follow their own control flow (exec masking); the LDS capacity bounds the problems in flight.
    const int forcedL = kn.lanes, forcedW = kn.waves;
    const int nmax = nVmax > nCmax ? nVmax : nCmax;
    int L = nmax <= 8 ? 8 : (nmax <= 16 ? 16 : (nmax <= 32 ? 32 : 64));
    if ((forcedL == 8 || forcedL == 16 || forcedL == 32 || forcedL == 64) && forcedL >= L) L = forcedL;   // never fewer lanes than entries
    if (eng == 1 && L < 16) L = 16;   // the explicit-inverse build has no 8-lane instantiation
    if (!mat_lds && !exp_nomat) L = 64;
    while (L < 64 && (64 / L) * stride > kMaxLds) L *= 2;
    if (L == 64 && stride > kMaxLds) stride = align16(mat_lds ? img + mat_bytes_max : img);
    const int forcedWide0 = kn.wide;
    const bool wide0 = eng == 1 && mat_lds && L == 64 && (forcedWide0 >= 0 ? forcedWide0 != 0 : nVmax > 32);
    bool wide = false;
    // several waves per problem: four (256 lanes, one wave per SIMD). The kernel keeps ~430 values live per lane
    // (256 VGPRs + AGPRs), so an eight-wave build (RSQP_SMALL_WIDE_LANES=512, tuning builds only) spills 233 of them.
    // With one wave per SIMD every wave instruction costs its full 4+ cycles: the four-wave kernel is bound by the
    // instruction count per wave (~350 per 69 x 69 product stage), not by LDS bandwidth or barriers.
#if defined(RSQP_SMALL_EXPERIMENT) && RSQP_SMALL_EXPERIMENT == 2
    const int wideL = kn.wide_lanes;
#else
    constexpr int wideL = 256;
#endif
    if (wide0 && stride + 8 * wideL <= kMaxLds) { stride += 8 * wideL; wide = true; }   // one double per lane of the wide build
    // bank spread of packed waves: a 32-lane LDS access group holds 32 / L problems, each touching 2 L consecutive
    // banks of the 64 (ds_read_b64: bank = dword address mod 64; stores: 16-lane groups, mod 32). Their images must
    // therefore start 2 L dwords apart modulo 64, i.e. stride = 8 L (mod 256) bytes -- with stride = 0 (mod 256)
    // every vector access of an 8-lane build is a 4-way conflict (measured: 54 % of the LDS-array cycles, LDS busy
    // 73 % of the kernel). The stride is padded to the next such value when that does not cost a resident workgroup.
    if (L < 64) {
        const int nospread = kn.nospread;
        long long s1 = stride;
        while ((s1 & 255) != ((8 * L) & 255)) s1 += 16;
        auto wgs = [&](long long st) { const long long a = (((64 / L) * st) + 511) / 512 * 512; return a > 0 ? kMaxLds / a : 0; };
        if (!nospread && wgs(s1) == wgs(stride) && (64 / L) * s1 <= kMaxLds) stride = s1;
    }
    const int G = 64 / L, nblk = (nq + G - 1) / G;
    const size_t lds = (size_t)(G * stride);
    // minimum resident waves per SIMD = register budget. One problem per wave keeps the uniform
    // state in SGPRs and runs best with 6 (small images) or 4 waves; packed waves hold that state
    // in VGPRs and need ~230 of them, so they run 2 waves/SIMD without spills (measured on
    // 16 384 hs071-scale QPs: L=16 W=2 159 M solves/s, W=3 142 M, W=4 116 M; L=64 W=6 74 M; with the
    // single-trip loop hints L=16 187 M, and on 65 536 QPs L=8 219 M vs L=16 194 M).
    // Packed builds with W=6 (80 VGPRs, ~180 spilled) returned wrong results and are not built.
    int waves = L == 64 ? (nVmax <= 16 ? 6 : 4) : 2;
    if (forcedW >= 2 && forcedW <= (L == 64 ? 6 : 4)) waves = forcedW;
    // ---- batches of mid-size problems (cold starts and hot starts on new vectors): the tableau kernel first (qp_small_g.h: 3 phases
    // per working-set change instead of ~50); members it cannot carry (non-symmetric H, LP, undecidable tests) come back with
    // ret == RET_BAIL and are solved by the null-space kernel launched right behind it, which skips everybody else
    const int noK = kn.no_kkt;
    // 32 row blocks x 8 column blocks of lanes: up to 72 variables x 32 constraints -- the 69 x 28 class of the hs0xx batch.
    typedef EngineG<3, 1, 9, 4> EK;      // up to 72 variables x 32 constraints
    typedef EngineG<2, 2, 8, 8> EK2;     // up to 64 variables x 64 constraints
    // (only where the null-space kernel would give a problem four waves as well: batches of SMALL problems are throughput-bound
    //  and better served by 16 / 32 lanes per problem, several problems per wave)
    if (!noK && forcedE < 0 && eng == 1 && (nVmax > 32 || nCmax > 32) && (mode == 0 || mode == 1) && !p.done_flag) {
#define KK_LAUNCH(RV_, RC_, CV_, CC_)                                                                                            \
        do {                                                                                                                     \
            hipLaunchKernelGGL((small_qpg_kernel<RV_, RC_, CV_, CC_>), dim3(nq), dim3(256), 0, stream, p, nq, mode, maxWSR);     \
            p.only_bailed = 1;                                                                                                   \
        } while (0)
        if (nVmax <= EK::MAXV && nCmax <= EK::MAXC) KK_LAUNCH(3, 1, 9, 4);
        else if (nVmax <= EK2::MAXV && nCmax <= EK2::MAXC) KK_LAUNCH(2, 2, 8, 8);
#undef KK_LAUNCH
    }
    const int konly = kn.kkt_only;     // diagnostics: no second pass (bailed members keep ret = 9, nflips = reason)
    if (konly && p.only_bailed) return hipGetLastError();
#define SQ_LAUNCH_U(ENG, LL, ML, W, U)                                                                        \
    do {                                                                                                      \
        static std::atomic<unsigned long long> set_{0};                                                       \
        rsqp_allow_full_lds(reinterpret_cast<const void *>(&small_qp_kernel<ENG<LL, ML>, LL, ML, W, U>), set_, (int)kMaxLds); \
        hipLaunchKernelGGL((small_qp_kernel<ENG<LL, ML>, LL, ML, W, U>), dim3(nblk), dim3(LL > 64 ? LL : 64), lds, stream, p, nq, \
                           (int)stride, mode, maxWSR);                                                        \
    } while (0)
    // compile-time shape NV x NC, target vectors in registers
#define SQ_LAUNCH_SHAPE(LL, W, NV, NC)                                                                        \
    do {                                                                                                      \
        static std::atomic<unsigned long long> set_{0};                                                       \
        rsqp_allow_full_lds(reinterpret_cast<const void *>(&small_qp_kernel<Engine<LL, true, true>, LL, true, W, NV * 256 + NC>), set_, (int)kMaxLds); \
        hipLaunchKernelGGL((small_qp_kernel<Engine<LL, true, true>, LL, true, W, NV * 256 + NC>), dim3(nblk), dim3(64), lds, stream, p, nq, \
                           (int)stride, mode, maxWSR);                                                        \
    } while (0)
#define SQ_LAUNCH_E(ENG, LL, ML, W) SQ_LAUNCH_U(ENG, LL, ML, W, 0)
#define SQ_LAUNCH(LL, ML, W) SQ_LAUNCH_E(Engine, LL, ML, W)
#define SQ_WAVES(LL)                                                                                          \
    switch (waves) {                                                                                          \
    case 3: SQ_LAUNCH(LL, true, 3); break;                                                                    \
    case 4: SQ_LAUNCH(LL, true, 4); break;                                                                    \
    default: SQ_LAUNCH(LL, true, 2); break;                                                                   \
    }
#if defined(RSQP_SMALL_EXPERIMENT) && RSQP_SMALL_EXPERIMENT == 2
    // quick-turnaround build for tuning (tools/small_experiment.sh -DRSQP_SMALL_EXPERIMENT=2): only the four-wave
    // explicit-inverse kernel (mid-size problems, BASELINE configs[4])
    {
        if (!(eng == 1 && wide)) return hipErrorInvalidValue;
        if (wideL == 512) SQ_LAUNCH_E(EngineX, 512, true, 1);
        else SQ_LAUNCH_E(EngineX, 256, true, 1);
        return hipGetLastError();
    }
#elif defined(RSQP_SMALL_EXPERIMENT)
    // quick-turnaround build for tuning (tools/small_experiment.sh): only the 8-lane Givens / TQ kernel
    {
        const bool fixed = shape82;
#ifdef RSQP_EXP_L16W6
        // the build the launcher refuses (qp_small.hip "Packed builds with W=6"): 16 lanes per problem, 6 waves per
        // SIMD = 80 VGPRs with ~180 spilled values; kept reachable only here, for the root-cause hunt
        if (L == 16 && eng == 0 && mat_lds) { SQ_LAUNCH_U(Engine, 16, true, 6, 0); return hipGetLastError(); }
#endif
        if (exp_nomat && L == 8 && eng == 0 && fixed) {
#define SQ_LAUNCH_SHAPE_G(W)                                                                                                  \
            do {                                                                                                              \
                static std::atomic<unsigned long long> set_{0};                                                               \
                rsqp_allow_full_lds(reinterpret_cast<const void *>(&small_qp_kernel<Engine<8, false, true>, 8, false, W, 8 * 256 + 2>), set_, (int)kMaxLds); \
                hipLaunchKernelGGL((small_qp_kernel<Engine<8, false, true>, 8, false, W, 8 * 256 + 2>), dim3(nblk), dim3(64), lds, stream, p, nq, \
                                   (int)stride, mode, maxWSR);                                                                \
            } while (0)
            switch (waves) { case 3: SQ_LAUNCH_SHAPE_G(3); break; case 4: SQ_LAUNCH_SHAPE_G(4); break; default: SQ_LAUNCH_SHAPE_G(2); }
#undef SQ_LAUNCH_SHAPE_G
            return hipGetLastError();
        }
        if (L != 8 || eng != 0 || !mat_lds) return hipErrorInvalidValue;
        if (fixed) { switch (waves) { case 3: SQ_LAUNCH_SHAPE(8, 3, 8, 2); break; case 4: SQ_LAUNCH_SHAPE(8, 4, 8, 2); break; default: SQ_LAUNCH_SHAPE(8, 2, 8, 2); } }
        else { switch (waves) { case 3: SQ_LAUNCH_U(Engine, 8, true, 3, 0); break; case 4: SQ_LAUNCH_U(Engine, 8, true, 4, 0); break; default: SQ_LAUNCH_U(Engine, 8, true, 2, 0); } }
        return hipGetLastError();
    }
#else
    if (eng == 1) {
        if (!mat_lds) SQ_LAUNCH_E(EngineX, 64, false, 3);
        else if (L == 16) SQ_LAUNCH_E(EngineX, 16, true, 2);
        else if (L == 32) SQ_LAUNCH_E(EngineX, 32, true, 2);
        else if (wide) SQ_LAUNCH_E(EngineX, 256, true, 1);   // four waves per problem: the O(n^2) phases split over 256 lanes
        else SQ_LAUNCH_E(EngineX, 64, true, 4);
    } else if (!mat_lds) {
        SQ_LAUNCH(64, false, 3);
    } else if (L == 8) {
        // shape build: 160 instead of 253 VGPRs, no per-vector address registers, straight-line vector loops; the
        // occupancy of both builds is capped at 2 waves per SIMD by the LDS a wave of 8 problems needs
        if (shape82) SQ_LAUNCH_SHAPE(8, 2, 8, 2);
        else SQ_LAUNCH(8, true, 2);
    } else if (shape82) {
        return hipErrorInvalidValue;    // the image was sized for the 8-lane shape build: never launch another one on it
    } else if (L == 16) {
        SQ_WAVES(16)
    } else if (L == 32) {
        SQ_WAVES(32)
    } else {
        switch (waves) {
        case 3: SQ_LAUNCH(64, true, 3); break;
        case 6: SQ_LAUNCH(64, true, 6); break;
        default: SQ_LAUNCH(64, true, 4); break;
        }
    }
#endif
#undef SQ_LAUNCH_SHAPE
#undef SQ_LAUNCH_E
#undef SQ_LAUNCH_U
#undef SQ_WAVES
#undef SQ_LAUNCH
    return hipGetLastError();
}

long long rsqp_mat_lds_bytes(int nV, int nC, int annz, int hnnz) { return mat_lds_bytes(nV, nC, annz, hnnz); }
