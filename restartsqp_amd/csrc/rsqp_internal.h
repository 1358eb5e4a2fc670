// rsqp_internal.h -- shared between the HIP translation units of librsqp_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#define RSQP_INFTY 1.0e20           // qpOASES INFTY
#define RSQP_EPS 2.221e-16          // qpOASES EPS
#define RSQP_EPS_DEN (1.0e3 * RSQP_EPS)
#define RSQP_BOUND_RELAXATION 1.0e4
#define RSQP_BOUND_TOLERANCE (1.0e6 * RSQP_EPS)
#define RSQP_EPS_LI 1.0e-9
#define RSQP_EPS_PD_REL 1.0e-10
#define RSQP_EPS_PD_ABS 1.0e-25

// solver status codes kept in the per-problem result record (qpOASES QProblemStatus order)
enum { QPS_NOTINITIALISED = 0, QPS_PREPARINGAUXILIARYQP = 1, QPS_AUXILIARYQPSOLVED = 2,
       QPS_PERFORMINGHOMOTOPY = 3, QPS_HOMOTOPYQPSOLVED = 4, QPS_SOLVED = 5 };
enum { RET_OK = 0, RET_MAX_NWSR = 1, RET_INFEASIBLE = 2, RET_UNBOUNDED = 3, RET_SETUP_FAILED = 4,
       RET_BAIL = 9 };   // internal: the KKT-tableau kernel hands the problem to the null-space kernel (never reported)

// one problem of a batch: sizes and offsets into the pooled device arrays
struct QPDesc {
    int nV, nC;
    int offV, offC;       // vector pools: g/lb/ub/x/ws_b at offV, lbA/ubA/ws_c at offC
    int offAjc, offAnz;   // A (CSC): Ajc at offAjc (nV+1 ints), Air/Aval/Aci/Arv at offAnz
    int offArp;           // A (CSR copy): Arp at offArp (nC+1 ints)
    int offHjc, offHnz;   // H (CSC, full symmetric)
    int haveH;
    int annz, hnnz;       // entry counts of A and H when the host knows them (saves a dependent load), else -1
    double hreg;          // H + hreg*I (LP path: qpOASES regularises an all-zero Hessian)
    long long offState;   // persistent engine image (doubles) for hot starts
};

// The RSQP_SMALL_* / RSQP_TINY_* / RSQP_K_* / RSQP_NO_SPIN / RSQP_NO_SPEC_CERT environment switches, read ONCE PER HANDLE (rsqp_create /
// rsqp_batch_create) and kept with it: a handle's behaviour does not depend on what another handle's first use found in the
// environment. Table of every switch: INTEGRATION.md 6.
struct SmallKnobs {
    int engine = -1;          // RSQP_SMALL_ENGINE      0 / 1: force the Givens-TQ / the explicit-inverse LDS kernel (no tableau kernels)
    int k_debug_bail = -1;    // RSQP_K_DEBUG_BAIL      test hook: the mid-size tableau kernel bails out of a hot start before its n-th change
    int noshape = 0;          // RSQP_SMALL_NOSHAPE     no compile-time 8 x 2 shape build
    int lanes = -1;           // RSQP_SMALL_LANES       lanes per problem (8 / 16 / 32 / 64)
    int waves = -1;           // RSQP_SMALL_WAVES       waves per SIMD the build is compiled for
    int wide = -1;            // RSQP_SMALL_WIDE        0 / 1: never / always four waves per problem
    int wide_lanes = 256;     // RSQP_SMALL_WIDE_LANES  (tuning builds) 512
    int nospread = 0;         // RSQP_SMALL_NOSPREAD    no bank-spreading pad of the LDS stride
    int no_kkt = 0;           // RSQP_SMALL_NO_KKT      no mid-size tableau kernel
    int kkt_only = 0;         // RSQP_SMALL_KKT_ONLY    diagnostics: no second pass for bailed members
    int no_tiny = 0;          // RSQP_SMALL_NO_TINY     no hs071-scale tableau kernel
    int lane = -1;            // RSQP_LANE              0: never the lane-per-problem kernel (qp_lane.hip); n > 0: from n members on (default 16 385)
    int tiny_lds = 0;         // RSQP_TINY_LDS          the hs071-scale kernel with its tableau in LDS, three waves per SIMD
    int exp_matglobal = 0;    // RSQP_EXP_MATGLOBAL     (tuning builds) matrices left in global memory
    int arena_mapped = -1;    // RSQP_ARENA_MAPPED      single-QP handles: patterns / plans in host-mapped memory, no upload at set_A / set_H (-1: hs071 scale only)
    int no_spin = 0;          // RSQP_NO_SPIN           single-QP waits block in hipStreamSynchronize instead of spinning on a mapped word
    int no_spec_cert = 0;     // RSQP_NO_SPEC_CERT      the certificate of a single LDS-scale QP only on demand
};
SmallKnobs rsqp_small_knobs_from_env();      // qp_small.hip

struct QPPools {
    const QPDesc *desc;
    const int *Ajc, *Air; const double *Aval;
    const int *Arp, *Aci; const double *Arv;
    const int *Hjc, *Hir; const double *Hval;
    const double *g, *lb, *ub, *lbA, *ubA;
    const double *x0, *y0; const int *guess_b;   // warm re-init inputs (may be null)
    double *x, *y; int *ws_b, *ws_c;
    int *status, *ret, *nwsr, *nflips; double *obj;
    double *state;
    int uniV, uniC;   // nV / nC of every problem when the batch is uniform in shape, else -1
    int reinit_from_y0;   // warm re-initialisation (mode 3) without guessed constraints: 1 = sides from sign(y0), 0 = from A x0
    int *done_flag;   // single-QP solves: host-mapped word that receives done_val once the results are out (the host spins
    int done_val;     //   on it instead of sleeping in hipStreamSynchronize); nullptr for batches
    int k_debug_bail; // test hook (RSQP_K_DEBUG_BAIL=n): the KKT-tableau kernel bails out of a HOT start before its n-th change; -1 off
    int only_bailed;  // 1: the null-space kernel runs only the members the KKT-tableau kernel left with ret == RET_BAIL
    double *cert_out; int *cert_Wb, *cert_Wc;   // single-QP handles on the tableau kernel of qp_tiny.hip: the KKT certificate
                      //    (6 doubles: primal, dual, compl, stat, KKT_error, invalid) and the mapped working set are formed at
                      //    the END of the solve kernel -- QPhandler::solveQP always asks for them (src/QPhandler.cpp:470-499); or nullptr
    int tiny_ok;      // 1: every H of the batch is symmetric (or absent): problems of <= 8 variables may take the register-resident
                      //    tableau kernel (qp_tiny.hip), which keeps K symmetric by construction
    int uni_pat;      // 1: every member has the sizes AND the sparsity patterns of member 0 (checked by the host when the batch is
    int uni_annz, uni_hnnz, uni_haveH;   //    created: parameter scans, perturbations of one QP) -- offsets are q * size then, so the kernel
    long long uni_state; double uni_hreg; //    needs no descriptor load, and it reads the PATTERN arrays of member 0 (cache-resident for the
                      //    whole launch) instead of its own copy: two dependent HBM round trips less in front of every solve
                      //    (single-QP handles are the batch of one: set as well; uni_hreg = the LP regularisation, 0 in batches)
    int skip_mark;    // 1 (batches with keep_state = 0 on the hs071-scale tableau kernel): the HOST remembers that no state was kept
                      //    (rsqp_batch::state_engine = -1: the next hot start runs cold), so the kernel does not touch the state block at
                      //    all -- the "not initialised" mark was one scattered 64-byte line per QP
    int keep_state;   // 1: write the hot-start part of the engine image back to HBM at the end of a solve (what the
                      //    SQProblem object keeps between calls); 0: cold-start-only batches skip that write --
                      //    the image is marked "not initialised", a later hot start falls back to a cold start
};

// number of doubles / ints of the LDS (and persistent) image of one problem
__host__ __device__ inline int rsqp_ld(int nV) { return nV | 1; }
__host__ __device__ inline long long rsqp_image_doubles(int nV, int nC) {
    long long ld = rsqp_ld(nV), sT = nV < nC ? nV : nC;
    //        Q|Z      R|Wz     T|Y      vectors of nV   vectors of nC   y,dy        scalars
    //        (explicit-inverse engine: Y lives inside Z's array, Minv in the T slot) + four vectors of sT + 2
    return ld * nV + ld * nV + sT * ld + 19LL * nV + 9LL * nC + 2LL * (nV + nC) + 16 + 4 * (sT + 2);
}
__host__ __device__ inline long long rsqp_image_ints(int nV, int nC) { return nV + 3LL * nC + 8; }
__host__ __device__ inline long long rsqp_image_bytes(int nV, int nC) {
    long long b = rsqp_image_doubles(nV, nC) * 8 + rsqp_image_ints(nV, nC) * 4;
    return (b + 15) & ~15LL;
}

// persistent state of one problem in HBM: the image of the null-space engines, followed by the extension of the
// KKT-tableau kernel (qp_small_g.h): G = -SWEEP_S(K) with one slot per variable and constraint, (nV + nC)^2 doubles
__host__ __device__ inline long long rsqp_state_bytes(int nV, int nC) {
    long long b = rsqp_image_bytes(nV, nC) + 8LL * (long long)(nV + nC) * (nV + nC);
    // hs071-scale problems (qp_tiny.hip): the register-resident tableau engine keeps a fixed-slot state of (8 + MC)^2 + 96
    // doubles + 24 ints (<= 2912 B) whatever the problem's own sizes are (MC = 2 / 4 / 8 by the batch's largest nC)
    if (nV <= 8 && nC <= 8 && b < 2944) b = 2944;
    return b;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: set it once per
// (kernel, device) -- `mask` is the kernel's own bitmask of devices already done (thread-safe)
inline void rsqp_allow_full_lds(const void *fn, std::atomic<unsigned long long> &mask, int bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(mask.load(std::memory_order_acquire) & bit)) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        mask.fetch_or(bit, std::memory_order_release);
    }
}

// launchers (defined in the .hip files)
hipError_t rsqp_launch_small_qp(const SmallKnobs &kn, const QPPools &p, int nq, int nVmax, int nCmax, long long mat_bytes_max, int mode,
                                int maxWSR, hipStream_t stream);
long long rsqp_mat_lds_bytes(int nV, int nC, int annz, int hnnz);
int rsqp_small_qp_fits(int nVmax, int nCmax);
// qp_tiny.hip: the register-resident tableau kernel for problems of at most 8 variables and 8 constraints
int rsqp_tiny_fits(const SmallKnobs &kn, int nVmax, int nCmax);
// qp_lane.hip: one lane per problem, for cold starts of large one-shape batches of at most 8 x 2
int rsqp_lane_fits(const SmallKnobs &kn, const QPPools &p, int nq, int nVmax, int nCmax, int mode);
hipError_t rsqp_launch_lane_qp(const QPPools &p, int nq, int maxWSR, hipStream_t stream);
// 1 when rsqp_launch_small_qp hands this launch to the register-resident tableau kernel (qp_tiny.hip), whose hot-start state has
// another layout than the LDS-resident kernels': the caller forces a cold start when the answer changes between two solves of a
// handle or batch
int rsqp_small_launch_is_tiny(const SmallKnobs &kn, const QPPools &p, int nVmax, int nCmax);
hipError_t rsqp_launch_tiny_qp(const SmallKnobs &kn, const QPPools &p, int nq, int nVmax, int nCmax, int mode, int maxWSR, hipStream_t stream);
