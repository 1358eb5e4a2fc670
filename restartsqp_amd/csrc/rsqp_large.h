// rsqp_large.h -- HBM-resident engine for QPs that exceed the LDS-resident kernel (qp_large.hip)
#pragma once
#include "rsqp_sparse.h"

enum { RSQP_LMODE_COLD = 0, RSQP_LMODE_HOT_VECTORS = 1, RSQP_LMODE_HOT_MATRICES = 2, RSQP_LMODE_WARM = 3 };

// device views of the solver's matrices (owned by rsqp_solver)
struct RsqpLargeMatrices {
    const int *Ajc = nullptr, *Air = nullptr; const double *Aval = nullptr; const int4 *blk_c = nullptr; int nblk_c = 0;
    const int *Arp = nullptr, *Aci = nullptr; const double *Arv = nullptr; const int4 *blk_r = nullptr; int nblk_r = 0;
    const int *Hjc = nullptr, *Hir = nullptr; const double *Hval = nullptr; const int4 *blk_h = nullptr; int nblk_h = 0;
    int haveH = 0, Annz = 0;
    const int *h_Hjc = nullptr, *h_Hir = nullptr; int Hnnz = 0;   // HOST mirror of H's pattern (full symmetric CSC): the general
                                                                  // range-space path reads the band width from it
    int sparse_rows = 0;             // rows of A are sparse enough (fill < 1/8) for gathered row products
    double hreg = 0.0;               // H + hreg*I
    int diagH = 0;                   // H is stored as exactly one entry per column, on the diagonal (Hval = its diagonal): the
                                     // engine may take its range-space path (qp_large.hip, Impl::dual)
    // optional dense column-major copies (null when the matrix is sparse)
    const double *denseA = nullptr;   // nC x nV, ld = nC
    const double *denseAT = nullptr;  // the same matrix row-major (= A' column-major, nV x nC, ld = nV): A x as a transposed
                                      // product, which shares a launch with H x (k_gemv_t2)
    const double *denseH = nullptr;   // nV x nV, ld = nV
};

class RsqpLargeEngine {
public:
    RsqpLargeEngine();
    ~RsqpLargeEngine();
    RsqpLargeEngine(const RsqpLargeEngine &) = delete;
    RsqpLargeEngine &operator=(const RsqpLargeEngine &) = delete;
    static long long bytes_needed(int nV, int nC);
    hipError_t init(int nV, int nC, hipStream_t stream);
    void set_matrices(const RsqpLargeMatrices &m);
    // vectors are device pointers; x0 / y0 / guess_b are host pointers (may be null)
    int solve(int mode, const double *d_g, const double *d_lb, const double *d_ub, const double *d_lbA,
              const double *d_ubA, int *nWSR, const double *h_x0, const double *h_y0, const int *h_guess_b);
    const double *d_x() const;
    const double *d_y() const;
    const int *d_Sb() const;
    const int *d_Sc() const;
    int status_word() const;
    int nflips() const;
    int path() const;               // 0 null-space, 1 range-space (diagonal H), 2 general range-space (banded H^-1), 3 (dense H^-1), 4 (3 + tableau)
    double objective();
    hipError_t last_error() const;
    // per-kernel-class accounting (HIP events around every launch of the class: serialises the stream, so
    // only for measurement runs). Classes: names(); get() fills {calls, ms, algorithmic bytes, 0} per class.
    static constexpr int PROFILE_CLASSES = 10;
    static const char *profile_name(int k);
    void profile_enable(bool on);
    void set_reinit_from_y0(bool on);   // warm start without guessed constraints: from A x0 (default, the reference) or sides from sign(y0) (opt-in)
    void profile_get(double *out4n) const;
    // the last blocked (matrix-core) set-up of a non-empty working set: {nFR, nAC, nZ, ms QR + Q + R^-1, ms Z'HZ + Cholesky +
    // inverse, algorithmic flops of the first part, of the second, 0}; range-space path: {nFR, nAC, nFR - nAC, ms Gram matrix,
    // ms Cholesky + inverse, their algorithmic flops, 1}; returns 0 when no blocked set-up has run
    int setup_profile(double *out8) const;
    // device ms per call of kernel class `kind` (0 gemv_n, 1 gemv_t, 2 ger) on an nrows x ncols matrix (<= nV each)
    int time_kernel(int kind, int nrows, int ncols, int reps, float *ms);
    struct Impl;
private:
    Impl *p_;
};
