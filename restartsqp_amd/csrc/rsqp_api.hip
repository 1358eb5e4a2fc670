// rsqp_api.hip -- the C ABI of include/rsqp_hip.h (host side of librsqp_hip.so).
//
// Host logic restated from the reference adapter src/qpOASESInterface.cpp: the
// FIXED/VARIED warm-start dispatch (:137-224, :817-833), dirty flags (:361-496),
// handle_error (:686-758), status mapping (:332-357). Structure analysis of
// SpHbMat::setStructure (src/SpHbMat.cpp:196-355) runs once on the host (a sort); every
// per-iteration operation (value refresh, products, certificate, QP solve) is a kernel.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/rsqp_hip.h"
#include "rsqp_large.h"
#include "rsqp_sparse.h"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
#define HIPCHK(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(RSQP_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    T *host = nullptr;   // non-null: p is the device view of host-mapped pinned memory owned elsewhere
    T *stage = nullptr;  // non-null: p is a slice of a device arena owned elsewhere and `stage` the same slice of its pinned staging
                         // mirror -- uploads are written there, the owner copies the arena to the device in ONE piece (DevMatrix)
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p && !host && !stage) (void)hipFree(p);
        p = nullptr;
        host = nullptr;
        stage = nullptr;
        n = 0;
    }
    void map(T *dev, T *hst, size_t count) { release(); p = dev; host = hst; n = count; }
    void carve(T *dev, T *stg, size_t count) { release(); p = dev; stage = stg; n = count; }
    hipError_t alloc(size_t count, bool zero = true) {
        release();
        n = count;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), std::max<size_t>(count, 1) * sizeof(T));
        if (e != hipSuccess) { p = nullptr; return e; }
        if (zero) e = hipMemset(p, 0, std::max<size_t>(count, 1) * sizeof(T));
        return e;
    }
    hipError_t upload(const T *h, size_t count) {
        if (count == 0) return hipSuccess;
        if (host) { std::memcpy(host, h, count * sizeof(T)); return hipSuccess; }
        if (stage) { std::memcpy(stage, h, count * sizeof(T)); return hipSuccess; }     // (reaches the device with the arena)
        return hipMemcpy(p, h, count * sizeof(T), hipMemcpyHostToDevice);
    }
    hipError_t from(const std::vector<T> &h) {
        hipError_t e = alloc(h.size(), false);
        if (e != hipSuccess) return e;
        return upload(h.data(), h.size());
    }
    hipError_t download(T *h, size_t count) const {
        if (count == 0) return hipSuccess;
        if (host) { std::memcpy(h, host, count * sizeof(T)); return hipSuccess; }   // caller has synchronised
        return hipMemcpy(h, p, count * sizeof(T), hipMemcpyDeviceToHost);
    }
};

// ---------------------------------------------------------------------------------
// structure analysis (host, one-off)
// ---------------------------------------------------------------------------------
struct Compressed {
    int nrow = 0, ncol = 0;
    std::vector<int> jc, ir, order, tmap;  // CSC; order[ext] = position; tmap[ext] = triplet index
    std::vector<double> val;
    int nnz() const { return (int)ir.size(); }
};

// SpHbMat::setStructure: sort the (extended) triplet list by (col,row); ties by position.
void csc_from_entries(int nrow, int ncol, const std::vector<int> &row1, const std::vector<int> &col1,
                      const std::vector<double> &v, Compressed &out) {
    const int n = (int)v.size();
    std::vector<int> perm(n);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) {
        if (col1[a] != col1[b]) return col1[a] < col1[b];
        return row1[a] < row1[b];
    });
    out.nrow = nrow; out.ncol = ncol;
    out.jc.assign(ncol + 1, 0); out.ir.resize(n); out.val.resize(n); out.order.resize(n);
    for (int p = 0; p < n; p++) {
        int e = perm[p];
        out.ir[p] = row1[e] - 1;
        out.val[p] = v[e];
        out.order[e] = p;
        out.jc[col1[e]]++;  // 1-based col -> slot col (= 0-based col + 1)
    }
    for (int c = 0; c < ncol; c++) out.jc[c + 1] += out.jc[c];
}

struct CsrCopy {
    std::vector<int> rp, ci, perm;  // perm[p] = CSC position of CSR entry p
};
void csr_from_csc(int nrow, int ncol, const int *jc, const int *ir, CsrCopy &out) {
    const int nnz = jc[ncol];
    out.rp.assign(nrow + 1, 0); out.ci.resize(nnz); out.perm.resize(nnz);
    for (int k = 0; k < nnz; k++) out.rp[ir[k] + 1]++;
    for (int r = 0; r < nrow; r++) out.rp[r + 1] += out.rp[r];
    std::vector<int> fill(nrow, 0);
    for (int c = 0; c < ncol; c++)
        for (int k = jc[c]; k < jc[c + 1]; k++) {
            int r = ir[k], p = out.rp[r] + fill[r]++;
            out.ci[p] = c;
            out.perm[p] = k;
        }
}

// blocks of consecutive majors with at most `chunk` entries; a longer major stands alone
std::vector<int4> build_blocks(int nmajor, const int *ptr, int chunk) {
    std::vector<int4> blk;
    int start = 0;
    while (start < nmajor) {
        int end = start + 1;
        while (end < nmajor && ptr[end + 1] - ptr[start] <= chunk && end - start < 4096) end++;
        blk.push_back(make_int4(start, end, ptr[start], ptr[end]));
        start = end;
    }
    return blk;
}

// entry-parallel SpMV variant (sparse.hip csx_ldsvec_segscan): chunks of whole majors with at most 512 entries and 128 majors, one
// "starts a major" bit per entry (16 words per chunk), the non-empty majors in order, the empty ones
struct SegHost {
    std::vector<int4> chunks;
    std::vector<int> nz, empties;
    std::vector<unsigned> bits;
    bool ok = true;
};
SegHost build_seg(int nmajor, const int *ptr) {
    constexpr int CH = 512;
    SegHost h;
    for (int c = 0; c < nmajor; c++) {
        const int len = ptr[c + 1] - ptr[c];
        if (len == 0) h.empties.push_back(c);
        else { h.nz.push_back(c); if (len > CH) h.ok = false; }
    }
    if (!h.ok) return h;
    size_t i = 0;
    while (i < h.nz.size()) {
        const int e0 = ptr[h.nz[i]];
        size_t j = i;
        unsigned w[16] = {0};
        while (j < h.nz.size() && ptr[h.nz[j] + 1] - e0 <= CH && j - i < 128) {     // <= 128 majors: the kernel's LDS window
            const int b = ptr[h.nz[j]] - e0;
            w[b >> 5] |= 1u << (b & 31);
            j++;
        }
        const int nent = ptr[h.nz[j - 1] + 1] - e0;
        if (nent < CH) w[nent >> 5] |= 1u << (nent & 31);      // one bit behind the last entry: what a lane loads past the chunk
                                                               // becomes a dummy major that the kernel never stores
        h.chunks.push_back(make_int4(e0, (int)i, nent, (int)(j - i)));
        h.bits.insert(h.bits.end(), w, w + 16);
        i = j;
    }
    return h;
}

int exitflag_of(int status_word, int ret) {
    // qpOASESInterface::get_status (src/qpOASESInterface.cpp:332-357)
    if (status_word >= 200) return RSQP_QPERROR_UNBOUNDED;
    if (status_word >= 100) return RSQP_QPERROR_INFEASIBLE;
    if (status_word == QPS_SOLVED) return RSQP_QP_OPTIMAL;
    (void)ret;
    switch (status_word) {
    case QPS_NOTINITIALISED: return RSQP_QPERROR_NOTINITIALISED;
    case QPS_PREPARINGAUXILIARYQP: return RSQP_QPERROR_PREPARINGAUXILIARYQP;
    case QPS_AUXILIARYQPSOLVED: return RSQP_QPERROR_AUXILIARYQPSOLVED;
    case QPS_PERFORMINGHOMOTOPY: return RSQP_QPERROR_PERFORMINGHOMOTOPY;
    case QPS_HOMOTOPYQPSOLVED: return RSQP_QPERROR_HOMOTOPYQPSOLVED;
    }
    return RSQP_QPERROR_UNKNOWN;
}

}  // namespace

// =====================================================================================
// one matrix on the device (CSC + optional CSR copy + spmv blocks)
// =====================================================================================
struct DevMatrix {
    int nrow = 0, ncol = 0, nnz = 0;
    bool initialised = false, symmetric = false, from_triplet = false;
    int n_ident_entries = 0, n_triplet = 0;
    double structure_seconds = 0.0;   // one-off structure analysis (setStructure: sort + CSC / CSR / SpMV plan + upload), rsqp_get_structure_seconds
    std::vector<int> h_jc, h_ir, h_order;  // host mirror of the pattern
    DevBuf<int> jc, ir, order, tmap;                 // CSC
    DevBuf<int4> blk_c, blk_r;
    DevBuf<double> val, tv;                          // tv: staging for triplet values
    DevBuf<int> rp, ci, perm, rorder;                // CSR copy (A only); rorder[i] = CSR position of triplet entry i (fused value refresh)
    DevBuf<double> rval;
    int nblk_c = 0, nblk_r = 0;
    bool have_csr = false;
    // LDS-scale single-QP handles: the VALUES (CSC and CSR copy) live in host-mapped pinned memory that the kernels read
    // directly -- a value refresh (SpHbMat::setMatVal through order_) is then a host loop over a few dozen entries, no copy
    // and no launch (a blocking hipMemcpy + a scatter launch cost ~15 us per matrix per SQP iteration of hs071)
    void *pin = nullptr;
    size_t pin_cap = 0;                    // entries each of the two value arrays in `pin` can hold
    std::vector<int> h_rorder, h_tmap, h_perm;
    // ... and everything else the structure analysis uploads (pattern, permutations, SpMV plan) is a slice of ONE device arena
    // with a pinned staging mirror, both allocated by rsqp_create -- where the reference allocates as well
    // (Algorithm::allocate_memory), outside the first SQP iteration: set_A / set_H of that iteration then cost one asynchronous
    // copy instead of 17 hipMalloc + 15 blocking hipMemcpy + 2 hipHostMalloc (365 -> ~90 us for the first iteration of hs071)
    char *arena_dev = nullptr, *arena_stage = nullptr;
    size_t arena_cap = 0, arena_used = 0;
    bool arena_mapped = false;             // the arena IS its staging mirror (host-mapped memory): no copy at all -- the kernels read the few
                                           // dozen pattern words of an hs071-scale matrix over the link, as they read its values already
    hipError_t reserve(int nrow_, int ncol_, bool mapped = false) {
        const size_t cap = (size_t)nrow_ * (size_t)ncol_ + 2 * (size_t)(nrow_ + ncol_) + 8;      // dense + an identity block or two
        arena_cap = 64 * cap + 64 * (size_t)(nrow_ + ncol_) + 4096;
        hipError_t e;
        arena_mapped = mapped;
        if (mapped) {
            e = hipHostMalloc(reinterpret_cast<void **>(&arena_stage), arena_cap, hipHostMallocMapped);
            if (e != hipSuccess) { arena_stage = nullptr; arena_cap = 0; return e; }
            e = hipHostGetDevicePointer(reinterpret_cast<void **>(&arena_dev), arena_stage, 0);
            if (e != hipSuccess) { arena_dev = nullptr; return e; }
        } else {
        e = hipMalloc(reinterpret_cast<void **>(&arena_dev), arena_cap);
        if (e != hipSuccess) { arena_dev = nullptr; arena_cap = 0; return e; }
        e = hipHostMalloc(reinterpret_cast<void **>(&arena_stage), arena_cap, hipHostMallocDefault);
        if (e != hipSuccess) { arena_stage = nullptr; return e; }
        }
        std::memset(arena_stage, 0, arena_cap);
        e = hipHostMalloc(&pin, 2 * (cap + 2) * sizeof(double), hipHostMallocMapped);
        if (e != hipSuccess) { pin = nullptr; return e; }
        std::memset(pin, 0, 2 * (cap + 2) * sizeof(double));
        pin_cap = cap + 2;
        return hipSuccess;
    }
    void release_arena() {     // undo a partial reserve(): without an arena every array is allocated on its own
        if (pin) { (void)hipHostFree(pin); pin = nullptr; pin_cap = 0; }
        if (arena_dev && !arena_mapped) (void)hipFree(arena_dev);
        arena_dev = nullptr;
        if (arena_stage) { (void)hipHostFree(arena_stage); arena_stage = nullptr; }
        arena_cap = arena_used = 0;
    }
    template <class T> bool take(DevBuf<T> &b, size_t count) {      // next slice of the arena (16-byte aligned, zero-filled)
        const size_t bytes = (std::max<size_t>(count, 1) * sizeof(T) + 15) & ~(size_t)15;
        if (!arena_dev || arena_used + bytes > arena_cap) return false;
        std::memset(arena_stage + arena_used, 0, bytes);
        b.carve(reinterpret_cast<T *>(arena_dev + arena_used), reinterpret_cast<T *>(arena_stage + arena_used), count);
        arena_used += bytes;
        return true;
    }
    void drop_slices() {
        jc.release(); ir.release(); order.release(); tmap.release(); tv.release(); blk_c.release(); blk_r.release();
        rp.release(); ci.release(); perm.release(); rorder.release();
    }
    ~DevMatrix() {
        if (pin) { val.release(); rval.release(); (void)hipHostFree(pin); }
        if (arena_dev) { drop_slices(); if (!arena_mapped) (void)hipFree(arena_dev); }
        if (arena_stage) (void)hipHostFree(arena_stage);
    }
};

struct rsqp_solver {
    int nV = 0, nC = 0, device = 0;
    int qp_maxiter = 1000, lp_maxiter = 100;
    hipStream_t stream = nullptr;
    DevMatrix A, H;
    // host staging of the vectors (scalar setter storm of QPhandler::update_bounds)
    std::vector<double> h_vec[5];
    bool vec_dirty = true;
    DevBuf<double> d_vec[5];
    // engine pools (batch of one)
    DevBuf<QPDesc> d_desc;
    DevBuf<double> d_x, d_y, d_obj, d_state, d_x0, d_y0;
    DevBuf<int> d_wsb, d_wsc, d_status, d_ret, d_nwsr, d_nflips, d_guess;
    DevBuf<int> d_dummy_i; DevBuf<double> d_dummy_d;
    // certificate scratch
    DevBuf<double> d_Ax, d_ATy, d_Hx, d_kkt, d_in, d_out;
    DevBuf<int> d_Wb, d_Wc;
    // results mirrored on the host
    std::vector<double> h_x, h_y;
    std::vector<int> h_wsb, h_wsc;
    int status_word = QPS_NOTINITIALISED, last_ret = 0, last_nflips = 0;
    double obj = 0.0;
    // dispatch state (qpOASESInterface.hpp:225-250)
    bool firstQPsolved = false;
    bool upd_A = false, upd_H = false, upd_bounds = false, upd_g = false;
    int old_status = 0, new_status = 0;  // 0 UNDEFINED, 1 FIXED, 2 VARIED
    bool desc_ready = false;
    // small problems: vectors in and results out live in ONE host-mapped pinned block that the
    // kernels read / write directly (zero-copy): a solve costs one launch and one sync
    void *io_host = nullptr;
    int *d_done = nullptr, *h_done = nullptr;   // host-mapped completion word of single-QP solves (spun on by rsqp_solve)
    // the certificate of a single LDS-scale QP is launched right BEHIND its solve (QPhandler::solveQP always asks for it): by the time
    // rsqp_test_optimality is called it has run, and the call only waits for its completion value -- no launch on the critical path.
    // Dropped by every setter (the certificate then runs on demand, as before).
    bool spec_cert = false;
    int spec_cert_val = 0;
    bool cert_pending = false;   // a speculative certificate kernel may still be reading the (host-mapped) matrix values
    int done_seq = 0;
    bool lp_mode = false;   // optimizeLP: H ignored, H := hreg*I
    double hreg = 0.0;
    // engine: 1 = LDS-resident kernel, 2 = HBM-resident engine
    int engine = 1;
    bool fits_small = true;
    RsqpLargeEngine *large = nullptr;
    bool large_ready = false, profile_large = false;
    SmallKnobs kn = rsqp_small_knobs_from_env();   // the environment switches of the LDS-scale kernels as they were when this handle was created
    int state_engine = -1;        // which kernel family wrote the hot-start state of this handle: 1 the register-resident tableau kernel, 0 the
                                  // LDS-resident ones (different layouts in the same block), -1 none yet
    bool h_sym = true;            // H symmetric value by value (or absent): the tableau kernel of qp_tiny.hip may take the handle
    int last_mode = -1;           // RSQP_MODE_* of the last rsqp_solve (what the dispatch of optimizeQP / optimizeLP chose): rsqp_get_last_mode
    bool reinit_from_y0 = false;  // rsqp_set_reinit_guess: default = the reference rule (qpOASESInterface.cpp:199-207); 1 = opt-in shortcut
    DevBuf<double> denseA, denseAT, denseH;   // dense copies for the HBM-resident engine (dense matrices only)
    ~rsqp_solver() {
        delete large;
        if (io_host) {   // mapped views first, then the block
            for (int k = 0; k < 5; k++) d_vec[k].release();
            d_x.release(); d_y.release(); d_obj.release(); d_wsb.release(); d_wsc.release();
            d_kkt.release(); d_Wb.release(); d_Wc.release();
            d_status.release(); d_ret.release(); d_nwsr.release(); d_nflips.release();
            (void)hipHostFree(io_host);
        }
    }
};

namespace {

// arena form of the function below (LDS-scale single-QP handles whose arena holds the matrix): no allocation, one copy
int upload_matrix_arena(DevMatrix &M, const Compressed &c, bool want_csr, hipStream_t stream) {
    M.arena_used = 0;
    M.drop_slices();
    const size_t n = M.pin_cap;
    void *dev = nullptr;
    HIPCHK(hipHostGetDevicePointer(&dev, M.pin, 0));
    M.val.map(static_cast<double *>(dev), static_cast<double *>(M.pin), n);
    M.rval.map(static_cast<double *>(dev) + n, static_cast<double *>(M.pin) + n, n);
    std::memset(M.pin, 0, 2 * n * sizeof(double));
    HIPCHK(M.val.upload(c.val.data(), c.val.size()));
    bool ok = M.take(M.jc, c.jc.size()) && M.take(M.ir, (size_t)M.nnz + 2) && M.take(M.order, std::max(M.nnz, 1)) &&
              M.take(M.tv, std::max(M.nnz, 1));
    if (ok && !c.tmap.empty()) ok = M.take(M.tmap, c.tmap.size());
    std::vector<int4> blk = build_blocks(M.ncol, c.jc.data(), rsqp_spmv_chunk());
    M.nblk_c = (int)blk.size();
    ok = ok && M.take(M.blk_c, blk.size());
    if (!ok) return 1;
    (void)M.jc.upload(c.jc.data(), c.jc.size()); (void)M.ir.upload(c.ir.data(), c.ir.size());
    (void)M.order.upload(c.order.data(), c.order.size());
    if (!c.tmap.empty()) (void)M.tmap.upload(c.tmap.data(), c.tmap.size());
    (void)M.blk_c.upload(blk.data(), blk.size());
    M.have_csr = want_csr;
    if (want_csr) {
        CsrCopy r;
        csr_from_csc(M.nrow, M.ncol, c.jc.data(), c.ir.data(), r);
        std::vector<int> inv(std::max(M.nnz, 1), 0), ro(std::max(M.nnz, 1), 0);
        for (int k = 0; k < M.nnz; k++) inv[r.perm[k]] = k;
        for (size_t i = 0; i < c.order.size(); i++) ro[i] = inv[c.order[i]];
        std::vector<int4> blr = build_blocks(M.nrow, r.rp.data(), rsqp_spmv_chunk());
        M.nblk_r = (int)blr.size();
        ok = M.take(M.rp, r.rp.size()) && M.take(M.ci, (size_t)M.nnz + 2) && M.take(M.perm, std::max(M.nnz, 1)) &&
             M.take(M.rorder, ro.size()) && M.take(M.blk_r, blr.size());
        if (!ok) return 1;
        (void)M.rp.upload(r.rp.data(), r.rp.size()); (void)M.ci.upload(r.ci.data(), r.ci.size());
        (void)M.perm.upload(r.perm.data(), r.perm.size()); (void)M.rorder.upload(ro.data(), ro.size());
        (void)M.blk_r.upload(blr.data(), blr.size());
        M.h_rorder = ro; M.h_perm = r.perm;
        for (int k = 0; k < M.nnz; k++) M.rval.host[k] = M.val.host[r.perm[k]];
    }
    if (!M.arena_mapped) HIPCHK(hipMemcpyAsync(M.arena_dev, M.arena_stage, M.arena_used, hipMemcpyHostToDevice, stream));
    M.initialised = true;
    return RSQP_OK;
}

int upload_matrix(DevMatrix &M, const Compressed &c, bool want_csr, bool zero_copy = false, hipStream_t stream = nullptr) {
    M.nrow = c.nrow; M.ncol = c.ncol; M.nnz = c.nnz();
    M.h_jc = c.jc; M.h_ir = c.ir; M.h_order = c.order; M.h_tmap = c.tmap;
    if (zero_copy && M.arena_dev && M.pin && (size_t)M.nnz + 2 <= M.pin_cap) {
        // (an earlier copy of the staging mirror may still be on its way -- or, mapped arena, a kernel may still be reading the old
        //  structure; nothing can be when the matrix is set for the first time)
        if (!(M.arena_mapped && !M.initialised)) (void)hipStreamSynchronize(stream);
        const int rc = upload_matrix_arena(M, c, want_csr, stream);
        if (rc == RSQP_OK) return rc;
        if (rc < 0) return rc;
        // the arena is too small for this matrix (rc == 1): the allocating path below, and the arena is not used again
        M.drop_slices(); M.val.release(); M.rval.release();
        if (!M.arena_mapped) (void)hipFree(M.arena_dev);
        M.arena_dev = nullptr; M.arena_cap = 0;
    }
    if (M.pin) { M.val.release(); M.rval.release(); (void)hipHostFree(M.pin); M.pin = nullptr; M.pin_cap = 0; }
    HIPCHK(M.jc.from(c.jc));
    HIPCHK(M.ir.alloc(M.nnz + 2, true)); HIPCHK(M.ir.upload(c.ir.data(), c.ir.size()));
    if (zero_copy) {
        const size_t n = (size_t)M.nnz + 2;
        HIPCHK(hipHostMalloc(&M.pin, 2 * n * sizeof(double), hipHostMallocMapped));
        M.pin_cap = n;
        std::memset(M.pin, 0, 2 * n * sizeof(double));
        void *dev = nullptr;
        HIPCHK(hipHostGetDevicePointer(&dev, M.pin, 0));
        M.val.map(static_cast<double *>(dev), static_cast<double *>(M.pin), n);
        M.rval.map(static_cast<double *>(dev) + n, static_cast<double *>(M.pin) + n, n);
        HIPCHK(M.val.upload(c.val.data(), c.val.size()));
    } else {
        HIPCHK(M.val.alloc(M.nnz + 2, true)); HIPCHK(M.val.upload(c.val.data(), c.val.size()));
    }
    HIPCHK(M.order.alloc(std::max(M.nnz, 1), true)); HIPCHK(M.order.upload(c.order.data(), c.order.size()));
    if (!c.tmap.empty()) { HIPCHK(M.tmap.from(c.tmap)); }
    HIPCHK(M.tv.alloc(std::max(M.nnz, 1), true));
    std::vector<int4> blk = build_blocks(M.ncol, c.jc.data(), rsqp_spmv_chunk());
    M.nblk_c = (int)blk.size();
    HIPCHK(M.blk_c.from(blk));
    M.have_csr = want_csr;
    if (want_csr) {
        CsrCopy r;
        csr_from_csc(M.nrow, M.ncol, c.jc.data(), c.ir.data(), r);
        HIPCHK(M.rp.from(r.rp));
        HIPCHK(M.ci.alloc(M.nnz + 2, true)); HIPCHK(M.ci.upload(r.ci.data(), r.ci.size()));
        HIPCHK(M.perm.alloc(std::max(M.nnz, 1), true)); HIPCHK(M.perm.upload(r.perm.data(), r.perm.size()));
        if (!M.pin) HIPCHK(M.rval.alloc(M.nnz + 2, true));
        {   // rorder = (CSC slot -> CSR slot) o order: where a refreshed triplet value lands in the CSR copy
            std::vector<int> inv(std::max(M.nnz, 1), 0), ro(std::max(M.nnz, 1), 0);
            for (int k = 0; k < M.nnz; k++) inv[r.perm[k]] = k;
            for (size_t i = 0; i < c.order.size(); i++) ro[i] = inv[c.order[i]];
            HIPCHK(M.rorder.from(ro));
            M.h_rorder = ro; M.h_perm = r.perm;
        }
        std::vector<int4> blr = build_blocks(M.nrow, r.rp.data(), rsqp_spmv_chunk());
        M.nblk_r = (int)blr.size();
        HIPCHK(M.blk_r.from(blr));
        if (M.pin) { for (int k = 0; k < M.nnz; k++) M.rval.host[k] = M.val.host[r.perm[k]]; }
        else if (((long long)M.nnz == (long long)M.nrow * M.ncol ? rsqp_launch_gather_dense(M.nrow, M.ncol, M.val.p, M.rval.p, nullptr)
                                                                 : rsqp_launch_gather(M.nnz, M.perm.p, M.val.p, M.rval.p, nullptr)) != hipSuccess)
            return fail(RSQP_ERR_DEVICE, "gather launch failed");
    }
    M.initialised = true;
    return RSQP_OK;
}

// is the CSC matrix (n <= 8 columns) symmetric, value by value? (eligibility of the tableau kernel of qp_tiny.hip)
bool small_csc_symmetric(int n, const int *jc, const int *ir, const double *val) {
    if (n > 8) return false;
    double d[64] = {0.0};
    for (int c = 0; c < n; c++)
        for (int k = jc[c]; k < jc[c + 1]; k++) {
            if (ir[k] < 0 || ir[k] >= n) return false;
            d[ir[k] * 8 + c] = val[k];
        }
    for (int r = 0; r < n; r++)
        for (int c = 0; c < r; c++)
            if (d[r * 8 + c] != d[c * 8 + r]) return false;
    return true;
}

int flush_vectors(rsqp_solver *s) {
    if (!s->vec_dirty) return RSQP_OK;
    for (int k = 0; k < 5; k++) HIPCHK(s->d_vec[k].upload(s->h_vec[k].data(), s->h_vec[k].size()));
    s->vec_dirty = false;
    return RSQP_OK;
}

int ensure_desc(rsqp_solver *s) {
    if (s->desc_ready) return RSQP_OK;
    QPDesc d;
    std::memset(&d, 0, sizeof(d));
    d.nV = s->nV; d.nC = s->nC; d.haveH = (s->H.initialised && !s->lp_mode) ? 1 : 0;
    d.annz = d.hnnz = -1;
    d.hreg = s->hreg;
    if (s->d_desc.p) { HIPCHK(s->d_desc.upload(&d, 1)); }       // (host-mapped for LDS-scale handles: a store, no copy)
    else { std::vector<QPDesc> hd(1, d); HIPCHK(s->d_desc.from(hd)); }
    s->desc_ready = true;
    return RSQP_OK;
}

QPPools pools_of(rsqp_solver *s) {
    QPPools p;
    std::memset(&p, 0, sizeof(p));
    p.desc = s->d_desc.p;
    p.Ajc = s->A.initialised ? s->A.jc.p : s->d_dummy_i.p;
    p.Air = s->A.initialised ? s->A.ir.p : s->d_dummy_i.p;
    p.Aval = s->A.initialised ? s->A.val.p : s->d_dummy_d.p;
    p.Arp = s->A.initialised ? s->A.rp.p : s->d_dummy_i.p;
    p.Aci = s->A.initialised ? s->A.ci.p : s->d_dummy_i.p;
    p.Arv = s->A.initialised ? s->A.rval.p : s->d_dummy_d.p;
    p.Hjc = s->H.initialised ? s->H.jc.p : s->d_dummy_i.p;
    p.Hir = s->H.initialised ? s->H.ir.p : s->d_dummy_i.p;
    p.Hval = s->H.initialised ? s->H.val.p : s->d_dummy_d.p;
    p.g = s->d_vec[RSQP_VEC_G].p; p.lb = s->d_vec[RSQP_VEC_LB].p; p.ub = s->d_vec[RSQP_VEC_UB].p;
    p.lbA = s->d_vec[RSQP_VEC_LBA].p; p.ubA = s->d_vec[RSQP_VEC_UBA].p;
    p.x = s->d_x.p; p.y = s->d_y.p; p.ws_b = s->d_wsb.p; p.ws_c = s->d_wsc.p;
    p.status = s->d_status.p; p.ret = s->d_ret.p; p.nwsr = s->d_nwsr.p; p.nflips = s->d_nflips.p;
    p.obj = s->d_obj.p; p.state = s->d_state.p;
    p.uniV = s->nV; p.uniC = s->nC;
    p.keep_state = 1;
    p.done_flag = nullptr; p.done_val = 0;
    p.reinit_from_y0 = s->reinit_from_y0 ? 1 : 0;
    p.tiny_ok = (s->h_sym || !s->H.initialised || s->lp_mode) ? 1 : 0;
    // the batch of one: the hs071-scale kernel computes the (zero) offsets itself instead of loading the descriptor
    p.uni_pat = 1;
    p.uni_annz = s->A.initialised ? s->A.nnz : 0; p.uni_hnnz = s->H.initialised ? s->H.nnz : 0;
    p.uni_haveH = (s->H.initialised && !s->lp_mode) ? 1 : 0; p.uni_state = 0; p.uni_hreg = s->hreg;
    return p;
}

int fetch_results(rsqp_solver *s) {
    HIPCHK(s->d_x.download(s->h_x.data(), s->nV));
    HIPCHK(s->d_y.download(s->h_y.data(), s->nV + s->nC));
    HIPCHK(s->d_wsb.download(s->h_wsb.data(), s->nV));
    HIPCHK(s->d_wsc.download(s->h_wsc.data(), s->nC));
    HIPCHK(s->d_status.download(&s->status_word, 1));
    HIPCHK(s->d_ret.download(&s->last_ret, 1));
    HIPCHK(s->d_nflips.download(&s->last_nflips, 1));
    HIPCHK(s->d_obj.download(&s->obj, 1));
    return RSQP_OK;
}

// wait for a single-QP kernel that raises the host-mapped completion word to `val`: spin for up to 2 ms, then block
int wait_done(rsqp_solver *s, int val) {
    int *flag = s->h_done;
    const auto t0 = std::chrono::steady_clock::now();
    for (int it = 0;; it++) {
        // acquire: the results the kernel wrote to host-mapped memory BEFORE it raised the word are read after this load
        // (a plain volatile read orders nothing on non-x86 hosts and lets the compiler hoist the result reads)
        // (raising the word is the kernel's last action; a hipStreamQuery here measured +4 us per call, 8 us per solveQP --
        //  a fault is reported by the next HIP call of the handle, as for any asynchronous launch)
        // (>=: the word is a sequence number; the certificate launched behind a solve may have raised it further already)
        if ((int)(__atomic_load_n(flag, __ATOMIC_ACQUIRE) - val) >= 0) return RSQP_OK;
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
        if ((it & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    HIPCHK(hipStreamSynchronize(s->stream));
    return RSQP_OK;
}


bool solved(const rsqp_solver *s) { return s->status_word == QPS_SOLVED; }
bool infeasible(const rsqp_solver *s) { return s->status_word >= 100 && s->status_word < 200; }

}  // namespace

// =====================================================================================
// library
// =====================================================================================
extern "C" const char *rsqp_version(void) { return "restartsqp_amd 0.1 (gfx950)"; }
extern "C" const char *rsqp_last_error(void) { return g_err.c_str(); }
extern "C" int rsqp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// =====================================================================================
// one QP
// =====================================================================================
extern "C" int rsqp_create(int nV, int nC, int device, rsqp_solver **out) {
    if (!out || nV <= 0 || nC < 0) return fail(RSQP_ERR_ARG, "rsqp_create: bad sizes");
    if (rsqp_device_count() <= 0) return fail(RSQP_ERR_DEVICE, "rsqp_create: no HIP device visible");
    if (device >= 0) HIPCHK(hipSetDevice(device));
    rsqp_solver *s = new rsqp_solver();
    s->nV = nV; s->nC = nC;
    s->fits_small = rsqp_small_qp_fits(nV, nC) != 0;
    s->engine = s->fits_small ? 1 : 2;
    HIPCHK(hipGetDevice(&s->device));
    for (int k = 0; k < 5; k++) s->h_vec[k].assign((k <= RSQP_VEC_UB) ? nV : nC, 0.0);
    if (s->fits_small) {
        const size_t nd = 3 * (size_t)nV + 2 * (size_t)nC + nV + (nV + nC) + 1 + 6;      // doubles
        const size_t ni = 2 * ((size_t)nV + nC) + 4 + 2;                                   // ints (+ the done flag)
        const size_t bytes_io = (nd * 8 + ni * 4 + 64 + 15) & ~(size_t)15;
        const size_t bytes = bytes_io + sizeof(QPDesc) + 16;          // (+ the problem descriptor: rewritten by a store, ensure_desc)
        HIPCHK(hipHostMalloc(&s->io_host, bytes, hipHostMallocMapped));
        std::memset(s->io_host, 0, bytes);
        void *dev = nullptr;
        HIPCHK(hipHostGetDevicePointer(&dev, s->io_host, 0));
        double *hd = static_cast<double *>(s->io_host), *dd = static_cast<double *>(dev);
        size_t o = 0;
        for (int k = 0; k < 5; k++) {
            const size_t n = (k <= RSQP_VEC_UB) ? nV : nC;
            s->d_vec[k].map(dd + o, hd + o, n);
            o += n;
        }
        s->d_x.map(dd + o, hd + o, nV); o += nV;
        s->d_y.map(dd + o, hd + o, nV + nC); o += nV + nC;
        s->d_obj.map(dd + o, hd + o, 1); o += 1;
        s->d_kkt.map(dd + o, hd + o, 6); o += 6;
        int *hi = reinterpret_cast<int *>(hd + o), *di = reinterpret_cast<int *>(dd + o);
        size_t q = 0;
        s->d_wsb.map(di + q, hi + q, nV); q += nV;
        s->d_wsc.map(di + q, hi + q, nC); q += nC;
        s->d_status.map(di + q, hi + q, 1); q++;
        s->d_ret.map(di + q, hi + q, 1); q++;
        s->d_nwsr.map(di + q, hi + q, 1); q++;
        s->d_nflips.map(di + q, hi + q, 1); q++;
        s->d_Wb.map(di + q, hi + q, nV); q += nV;
        s->d_Wc.map(di + q, hi + q, nC); q += nC;
        s->d_done = di + q; s->h_done = hi + q; q++;
        s->d_desc.map(reinterpret_cast<QPDesc *>(static_cast<char *>(dev) + bytes_io),
                      reinterpret_cast<QPDesc *>(static_cast<char *>(s->io_host) + bytes_io), 1);
        // (a failed reservation is "no arena": set_A / set_H then allocate per array as the HBM-scale handles do -- ADVICE r4)
        // (hs071-scale handles keep the arena in host-mapped memory: set_A / set_H then cost a store each instead of a stream wait and a
        //  copy -- 78 -> 37 us on the first QP of an SQP run, later QPs unchanged; RSQP_ARENA_MAPPED=0 / 1 forces either form)
        const bool am = s->kn.arena_mapped >= 0 ? s->kn.arena_mapped != 0 : rsqp_tiny_fits(s->kn, nV, nC) != 0;
        if (nC > 0 && s->A.reserve(nC, nV, am) != hipSuccess) { s->A.release_arena(); (void)hipGetLastError(); }
        if (s->H.reserve(nV, nV, am) != hipSuccess) { s->H.release_arena(); (void)hipGetLastError(); }
    } else {
        for (int k = 0; k < 5; k++) HIPCHK(s->d_vec[k].alloc((k <= RSQP_VEC_UB) ? nV : nC));
        HIPCHK(s->d_x.alloc(nV)); HIPCHK(s->d_y.alloc(nV + nC)); HIPCHK(s->d_obj.alloc(1));
        HIPCHK(s->d_wsb.alloc(nV)); HIPCHK(s->d_wsc.alloc(nC));
        HIPCHK(s->d_status.alloc(1)); HIPCHK(s->d_ret.alloc(1)); HIPCHK(s->d_nwsr.alloc(1)); HIPCHK(s->d_nflips.alloc(1));
        HIPCHK(s->d_kkt.alloc(6)); HIPCHK(s->d_Wb.alloc(nV)); HIPCHK(s->d_Wc.alloc(nC));
    }
    HIPCHK(s->d_state.alloc(s->fits_small ? (size_t)rsqp_state_bytes(nV, nC) / 8 : 1));
    HIPCHK(s->d_x0.alloc(nV)); HIPCHK(s->d_y0.alloc(nV + nC)); HIPCHK(s->d_guess.alloc(nV));
    HIPCHK(s->d_dummy_i.alloc(std::max(nV, nC) + 2)); HIPCHK(s->d_dummy_d.alloc(4));
    HIPCHK(s->d_Ax.alloc(nC)); HIPCHK(s->d_ATy.alloc(nV)); HIPCHK(s->d_Hx.alloc(nV));
    HIPCHK(s->d_in.alloc(std::max(nV, nC))); HIPCHK(s->d_out.alloc(std::max(nV, nC)));
    s->h_x.assign(nV, 0.0); s->h_y.assign(nV + nC, 0.0); s->h_wsb.assign(nV, 0); s->h_wsc.assign(nC, 0);
    *out = s;
    return RSQP_OK;
}

extern "C" void rsqp_destroy(rsqp_solver *s) { delete s; }

extern "C" int rsqp_set_engine(rsqp_solver *s, int engine) {
    if (s) s->spec_cert = false;
    if (!s || engine < 0 || engine > 2) return fail(RSQP_ERR_ARG, "rsqp_set_engine");
    if (engine == 1 && !s->fits_small)
        return fail(RSQP_ERR_TOO_LARGE, "rsqp_set_engine: problem image exceeds the 160 KiB LDS-resident engine");
    s->engine = engine == 0 ? (s->fits_small ? 1 : 2) : engine;
    return RSQP_OK;
}
extern "C" int rsqp_get_engine(const rsqp_solver *s) { return s ? s->engine : -1; }
// measurement only: per-kernel-class device time of the HBM-resident engine (bench.py)
extern "C" int rsqp_set_engine_profiling(rsqp_solver *s, int on) {
    if (!s) return fail(RSQP_ERR_ARG, "null solver");
    s->profile_large = on != 0;
    if (s->large) s->large->profile_enable(s->profile_large);
    return RSQP_OK;
}
extern "C" int rsqp_get_engine_profile(const rsqp_solver *s, double *out4n, int n) {
    if (!s || !s->large) return 0;
    if (out4n && n >= RsqpLargeEngine::PROFILE_CLASSES) s->large->profile_get(out4n);
    return RsqpLargeEngine::PROFILE_CLASSES;
}
extern "C" int rsqp_engine_profile_names(const char **names, int n) {
    for (int k = 0; names && k < n && k < RsqpLargeEngine::PROFILE_CLASSES; k++) names[k] = RsqpLargeEngine::profile_name(k);
    return RsqpLargeEngine::PROFILE_CLASSES;
}
extern "C" int rsqp_get_setup_profile(const rsqp_solver *s, double *out8) {
    if (!s || !out8 || !s->large) return 0;
    return s->large->setup_profile(out8);
}
extern "C" double rsqp_get_structure_seconds(const rsqp_solver *s, int which) {
    if (!s) return -1.0;
    const DevMatrix &M = which == 0 ? s->A : s->H;
    return M.initialised ? M.structure_seconds : -1.0;
}
extern "C" int rsqp_get_last_mode(const rsqp_solver *s) { return s ? s->last_mode : -1; }
extern "C" int rsqp_get_large_path(const rsqp_solver *s) { return (s && s->large && s->large_ready) ? s->large->path() : -1; }
extern "C" int rsqp_get_nV(const rsqp_solver *s) { return s ? s->nV : -1; }
extern "C" int rsqp_get_nC(const rsqp_solver *s) { return s ? s->nC : -1; }

extern "C" int rsqp_set_reinit_guess(rsqp_solver *s, int from_y0) {
    if (!s) return fail(RSQP_ERR_ARG, "null solver");
    s->reinit_from_y0 = from_y0 != 0;
    return RSQP_OK;
}

extern "C" int rsqp_set_options(rsqp_solver *s, int qp_maxiter, int lp_maxiter) {
    if (!s || qp_maxiter < 0 || lp_maxiter < 0) return fail(RSQP_ERR_ARG, "rsqp_set_options");
    s->qp_maxiter = qp_maxiter; s->lp_maxiter = lp_maxiter;
    return RSQP_OK;
}

extern "C" int rsqp_set_A_triplet(rsqp_solver *s, int nnz, const int *irow, const int *jcol, const double *val,
                                  int n_ident, const int *id_irow, const int *id_jcol, const int *id_size,
                                  const double *id_value) {
    if (s) s->spec_cert = false;
    if (!s || nnz < 0 || (nnz > 0 && (!irow || !jcol || !val))) return fail(RSQP_ERR_ARG, "rsqp_set_A_triplet");
    if (s->firstQPsolved && !s->upd_A) s->upd_A = true;  // qpOASESInterface.cpp:427-429
    DevMatrix &M = s->A;
    if (!M.initialised) {
        std::vector<int> r(irow, irow + nnz), c(jcol, jcol + nnz);
        std::vector<double> v(val, val + nnz);
        int nid = 0;
        for (int b = 0; b < n_ident; b++)
            for (int j = 0; j < id_size[b]; j++) {
                r.push_back(id_irow[b] + j); c.push_back(id_jcol[b] + j); v.push_back(id_value[b]);
                nid++;
            }
        for (size_t k = 0; k < r.size(); k++)
            if (r[k] < 1 || r[k] > s->nC || c[k] < 1 || c[k] > s->nV)
                return fail(RSQP_ERR_ARG, "rsqp_set_A_triplet: index out of range (indices are 1-based)");
        const auto t0 = std::chrono::steady_clock::now();
        Compressed cs;
        csc_from_entries(s->nC, s->nV, r, c, v, cs);
        M.from_triplet = true; M.n_triplet = nnz; M.n_ident_entries = nid;
        int rc = upload_matrix(M, cs, true, s->fits_small, s->stream);
        if (rc != RSQP_OK) return rc;
        if (!M.arena_dev) (void)hipStreamSynchronize(s->stream);   // this handle's stream only (the uploads are blocking copies): other handles keep running
        M.structure_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        s->desc_ready = false;
        return RSQP_OK;
    }
    if (!M.from_triplet || nnz != M.n_triplet) return fail(RSQP_ERR_ARG, "rsqp_set_A_triplet: pattern changed");
    // SpHbMat::setMatVal(rhs, I_info): only the first nnz(J) entries are rewritten
    if (M.pin) {
        // host-mapped values: the scatter through order_ is a host loop (no kernel of this handle is running: every solve and
        // certificate of the single-QP boundary is waited for before its call returns)
        if (s->cert_pending) { HIPCHK(hipStreamSynchronize(s->stream)); s->cert_pending = false; }
        double *v = M.val.host, *rv = M.rval.host;
        for (int i = 0; i < nnz; i++) { v[M.h_order[i]] = val[i]; rv[M.h_rorder[i]] = val[i]; }
        return RSQP_OK;
    }
    HIPCHK(M.tv.upload(val, nnz));
    // one launch: every refreshed value goes to its CSC slot and to its slot of the CSR copy (the identity entries of [J I -I]
    // keep their values in both, SpHbMat.cpp:368-380)
    if (rsqp_launch_scatter_csc_csr(nnz, M.order.p, M.rorder.p, M.tv.p, M.val.p, M.rval.p, s->stream) != hipSuccess)
        return fail(RSQP_ERR_DEVICE, "value refresh launch failed");
    return RSQP_OK;
}

extern "C" int rsqp_set_H_triplet(rsqp_solver *s, int nnz, const int *irow, const int *jcol, const double *val,
                                  int is_symmetric) {
    if (s) s->spec_cert = false;
    if (!s || nnz < 0 || (nnz > 0 && (!irow || !jcol || !val))) return fail(RSQP_ERR_ARG, "rsqp_set_H_triplet");
    if (s->firstQPsolved && !s->upd_H) s->upd_H = true;  // :407-409
    DevMatrix &M = s->H;
    if (!M.initialised) {
        std::vector<int> r, c, tmap;
        std::vector<double> v;
        for (int i = 0; i < nnz; i++) {
            if (irow[i] < 1 || irow[i] > s->nV || jcol[i] < 1 || jcol[i] > s->nV)
                return fail(RSQP_ERR_ARG, "rsqp_set_H_triplet: index out of range (indices are 1-based)");
            r.push_back(irow[i]); c.push_back(jcol[i]); v.push_back(val[i]); tmap.push_back(i);
            if (is_symmetric && irow[i] != jcol[i]) {  // mirror right behind (SpHbMat.cpp:302-308)
                r.push_back(jcol[i]); c.push_back(irow[i]); v.push_back(val[i]); tmap.push_back(i);
            }
        }
        const auto t0 = std::chrono::steady_clock::now();
        Compressed cs;
        csc_from_entries(s->nV, s->nV, r, c, v, cs);
        cs.tmap = tmap;
        s->h_sym = is_symmetric != 0 || (s->nV <= 8 && small_csc_symmetric(s->nV, cs.jc.data(), cs.ir.data(), cs.val.data()));
        M.from_triplet = true; M.n_triplet = nnz; M.symmetric = is_symmetric != 0;
        int rc = upload_matrix(M, cs, false, s->fits_small, s->stream);
        if (rc != RSQP_OK) return rc;
        if (!M.arena_dev) (void)hipStreamSynchronize(s->stream);   // this handle's stream only (the uploads are blocking copies): other handles keep running
        M.structure_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        s->desc_ready = false;
        return RSQP_OK;
    }
    if (!M.from_triplet || nnz != M.n_triplet) return fail(RSQP_ERR_ARG, "rsqp_set_H_triplet: pattern changed");
    if (!M.symmetric) s->h_sym = false;       // (values of a general triplet matrix: re-examined below where they are at hand)
    if (M.pin) {
        if (s->cert_pending) { HIPCHK(hipStreamSynchronize(s->stream)); s->cert_pending = false; }
        double *v = M.val.host;
        for (int j = 0; j < M.nnz; j++) v[M.h_order[j]] = val[M.h_tmap.empty() ? j : M.h_tmap[j]];
        if (!M.symmetric) s->h_sym = s->nV <= 8 && small_csc_symmetric(s->nV, M.h_jc.data(), M.h_ir.data(), v);
        return RSQP_OK;
    }
    HIPCHK(M.tv.upload(val, nnz));
    if (rsqp_launch_scatter(M.nnz, M.order.p, M.tmap.p, M.tv.p, M.val.p, s->stream) != hipSuccess)
        return fail(RSQP_ERR_DEVICE, "value refresh launch failed");
    return RSQP_OK;
}

namespace {
int set_csc(rsqp_solver *s, DevMatrix &M, int nrow, int ncol, const int *jc, const int *ir, const double *val,
            bool want_csr, bool *flag) {
    if (!s || !jc || (jc[ncol] > 0 && (!ir || !val))) return fail(RSQP_ERR_ARG, "rsqp_set_*_csc");
    if (s->firstQPsolved && !*flag) *flag = true;
    const int nnz = jc[ncol];
    // same pattern (compared entry by entry, not just by count): refresh values
    if (M.initialised && nnz == M.nnz && !M.from_triplet && M.nrow == nrow && M.ncol == ncol &&
        std::equal(jc, jc + ncol + 1, M.h_jc.begin()) && std::equal(ir, ir + nnz, M.h_ir.begin())) {
        if (M.pin && s->cert_pending) { HIPCHK(hipStreamSynchronize(s->stream)); s->cert_pending = false; }
        HIPCHK(M.val.upload(val, nnz));
        if (M.pin) { if (M.have_csr) for (int k = 0; k < M.nnz; k++) M.rval.host[k] = val[M.h_perm[k]]; }
        else if (M.have_csr && ((long long)M.nnz == (long long)M.nrow * M.ncol ? rsqp_launch_gather_dense(M.nrow, M.ncol, M.val.p, M.rval.p, s->stream)
                                                                                : rsqp_launch_gather(M.nnz, M.perm.p, M.val.p, M.rval.p, s->stream)) != hipSuccess)
            return fail(RSQP_ERR_DEVICE, "gather launch failed");
        return RSQP_OK;
    }
    Compressed cs;
    cs.nrow = nrow; cs.ncol = ncol;
    cs.jc.assign(jc, jc + ncol + 1); cs.ir.assign(ir, ir + nnz); cs.val.assign(val, val + nnz);
    cs.order.resize(nnz);
    std::iota(cs.order.begin(), cs.order.end(), 0);
    for (int c = 0; c < ncol; c++) {
        if (jc[c] > jc[c + 1]) return fail(RSQP_ERR_ARG, "rsqp_set_*_csc: column pointers not monotone");
        for (int k = jc[c]; k < jc[c + 1]; k++)
            if (ir[k] < 0 || ir[k] >= nrow) return fail(RSQP_ERR_ARG, "rsqp_set_*_csc: row index out of range");
    }
    // a new pattern on an initialised matrix: everything derived from the old one (CSR copy, SpMV
    // blocks, host mirror) is rebuilt; the dirty flag set above makes optimizeQP re-factorise
    M.from_triplet = false;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = upload_matrix(M, cs, want_csr, s->fits_small, s->stream);
    s->desc_ready = false;
    if (rc == RSQP_OK) {
        if (!M.arena_dev) (void)hipStreamSynchronize(s->stream);   // this handle's stream only (the uploads are blocking copies): other handles keep running
        M.structure_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return rc;
}
int get_csc(const DevMatrix &M, hipStream_t stream, int *jc, int *ir, double *val, int *order) {
    if (!M.initialised) return fail(RSQP_ERR_ARG, "matrix not set");
    if (jc) std::copy(M.h_jc.begin(), M.h_jc.end(), jc);
    if (ir) std::copy(M.h_ir.begin(), M.h_ir.end(), ir);
    if (order) std::copy(M.h_order.begin(), M.h_order.end(), order);
    if (val) {
        HIPCHK(hipStreamSynchronize(stream));   // the value refresh kernels of this handle
        HIPCHK(M.val.download(val, M.nnz));
    }
    return RSQP_OK;
}
}  // namespace

extern "C" int rsqp_set_A_csc(rsqp_solver *s, const int *jc, const int *ir, const double *val) {
    if (s) s->spec_cert = false;
    if (!s) return fail(RSQP_ERR_ARG, "null solver");
    return set_csc(s, s->A, s->nC, s->nV, jc, ir, val, true, &s->upd_A);
}
extern "C" int rsqp_set_H_csc(rsqp_solver *s, const int *jc, const int *ir, const double *val) {
    if (s) s->spec_cert = false;
    if (!s) return fail(RSQP_ERR_ARG, "null solver");
    if (jc && (jc[s->nV] == 0 || (ir && val))) s->h_sym = s->nV <= 8 && small_csc_symmetric(s->nV, jc, ir, val);
    return set_csc(s, s->H, s->nV, s->nV, jc, ir, val, false, &s->upd_H);
}
extern "C" int rsqp_get_A_nnz(const rsqp_solver *s) { return s && s->A.initialised ? s->A.nnz : -1; }
extern "C" int rsqp_get_H_nnz(const rsqp_solver *s) { return s && s->H.initialised ? s->H.nnz : -1; }
extern "C" int rsqp_get_A_csc(const rsqp_solver *s, int *jc, int *ir, double *val, int *order) {
    if (!s) return fail(RSQP_ERR_ARG, "null solver");
    return get_csc(s->A, s->stream, jc, ir, val, order);
}
extern "C" int rsqp_get_H_csc(const rsqp_solver *s, int *jc, int *ir, double *val, int *order) {
    if (!s) return fail(RSQP_ERR_ARG, "null solver");
    return get_csc(s->H, s->stream, jc, ir, val, order);
}

extern "C" int rsqp_set_vector(rsqp_solver *s, int which, const double *v) {
    if (!s || which < 0 || which > 4 || !v) return fail(RSQP_ERR_ARG, "rsqp_set_vector");
    if (s->firstQPsolved) { if (which == RSQP_VEC_G) s->upd_g = true; else s->upd_bounds = true; }
    std::copy(v, v + s->h_vec[which].size(), s->h_vec[which].begin());
    s->vec_dirty = true;
    return RSQP_OK;
}
extern "C" int rsqp_set_entry(rsqp_solver *s, int which, int location, double value) {
    if (!s || which < 0 || which > 4 || location < 0 || location >= (int)s->h_vec[which].size())
        return fail(RSQP_ERR_ARG, "rsqp_set_entry");
    if (s->firstQPsolved) { if (which == RSQP_VEC_G) s->upd_g = true; else s->upd_bounds = true; }
    s->h_vec[which][location] = value;
    s->vec_dirty = true;
    return RSQP_OK;
}
extern "C" int rsqp_get_vector(const rsqp_solver *s, int which, double *v) {
    if (!s || which < 0 || which > 4 || !v) return fail(RSQP_ERR_ARG, "rsqp_get_vector");
    std::copy(s->h_vec[which].begin(), s->h_vec[which].end(), v);
    return RSQP_OK;
}
extern "C" int rsqp_reset_constraints(rsqp_solver *s) {
    if (s) s->spec_cert = false;
    if (!s) return fail(RSQP_ERR_ARG, "null solver");
    for (int k = RSQP_VEC_LB; k <= RSQP_VEC_UBA; k++) std::fill(s->h_vec[k].begin(), s->h_vec[k].end(), 0.0);
    s->vec_dirty = true;
    return RSQP_OK;
}

namespace {
int solve_large(rsqp_solver *s, int mode, int *nWSR, const double *x0, const double *y0, const int *guess_b) {
    if (!s->large) { s->large = new RsqpLargeEngine(); if (s->profile_large) s->large->profile_enable(true); }
    if (!s->large_ready) {
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        if ((long long)free_b < RsqpLargeEngine::bytes_needed(s->nV, s->nC))
            return fail(RSQP_ERR_TOO_LARGE, "HBM-resident engine: not enough device memory");
        HIPCHK(s->large->init(s->nV, s->nC, s->stream));
        s->large_ready = true;
    }
    RsqpLargeMatrices m;
    if (s->A.initialised) {
        m.Ajc = s->A.jc.p; m.Air = s->A.ir.p; m.Aval = s->A.val.p; m.blk_c = s->A.blk_c.p; m.nblk_c = s->A.nblk_c;
        m.Arp = s->A.rp.p; m.Aci = s->A.ci.p; m.Arv = s->A.rval.p; m.blk_r = s->A.blk_r.p; m.nblk_r = s->A.nblk_r;
        m.sparse_rows = 8.0 * (double)s->A.nnz < (double)s->nC * s->nV;
        m.Annz = s->A.nnz;
    }
    m.hreg = s->hreg;
    if (s->H.initialised && !s->lp_mode) {
        m.Hjc = s->H.jc.p; m.Hir = s->H.ir.p; m.Hval = s->H.val.p; m.blk_h = s->H.blk_c.p; m.nblk_h = s->H.nblk_c;
        m.haveH = 1;
        if ((int)s->H.h_jc.size() == s->nV + 1 && (int)s->H.h_ir.size() == s->H.nnz) { m.h_Hjc = s->H.h_jc.data(); m.h_Hir = s->H.h_ir.data(); m.Hnnz = s->H.nnz; }
        // a diagonal Hessian (one entry per column, on the diagonal): the engine's range-space path
        bool diag = s->H.nnz == s->nV && (int)s->H.h_jc.size() == s->nV + 1;
        for (int c = 0; c < s->nV && diag; c++) diag = s->H.h_jc[c] == c && s->H.h_ir[c] == c;
        m.diagH = diag ? 1 : 0;
    }
    // dense copies when more than a quarter of the entries are stored (refreshed every solve
    // that follows a matrix update: cheap next to the solve)
    if (s->A.initialised && (double)s->A.nnz > 0.25 * (double)s->nC * s->nV) {
        if (!s->denseA.p) HIPCHK(s->denseA.alloc((size_t)s->nC * s->nV, false));
        if (rsqp_launch_densify(s->nC, s->nV, s->A.jc.p, s->A.ir.p, s->A.val.p, s->denseA.p, s->stream) != hipSuccess)
            return fail(RSQP_ERR_DEVICE, "densify launch failed");
        m.denseA = s->denseA.p;
        // the row-major copy (the CSR arrays are the CSC arrays of A')
        if (!s->denseAT.p) HIPCHK(s->denseAT.alloc((size_t)s->nC * s->nV, false));
        if (rsqp_launch_densify(s->nV, s->nC, s->A.rp.p, s->A.ci.p, s->A.rval.p, s->denseAT.p, s->stream) != hipSuccess)
            return fail(RSQP_ERR_DEVICE, "densify launch failed");
        m.denseAT = s->denseAT.p;
    }
    if (s->H.initialised && !s->lp_mode && (double)s->H.nnz > 0.25 * (double)s->nV * s->nV) {
        if (!s->denseH.p) HIPCHK(s->denseH.alloc((size_t)s->nV * s->nV, false));
        if (rsqp_launch_densify(s->nV, s->nV, s->H.jc.p, s->H.ir.p, s->H.val.p, s->denseH.p, s->stream) != hipSuccess)
            return fail(RSQP_ERR_DEVICE, "densify launch failed");
        m.denseH = s->denseH.p;
    }
    s->large->set_matrices(m);
    s->large->set_reinit_from_y0(s->reinit_from_y0);
    int rc = s->large->solve(mode, s->d_vec[RSQP_VEC_G].p, s->d_vec[RSQP_VEC_LB].p, s->d_vec[RSQP_VEC_UB].p,
                             s->d_vec[RSQP_VEC_LBA].p, s->d_vec[RSQP_VEC_UBA].p, nWSR, x0, y0, guess_b);
    if (s->large->last_error() != hipSuccess)
        return fail(RSQP_ERR_DEVICE, std::string("HBM-resident engine: ") + hipGetErrorString(s->large->last_error()));
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(s->d_x.p, s->large->d_x(), sizeof(double) * s->nV, hipMemcpyDefault));
    HIPCHK(hipMemcpy(s->d_y.p, s->large->d_y(), sizeof(double) * (s->nV + s->nC), hipMemcpyDefault));
    HIPCHK(hipMemcpy(s->d_wsb.p, s->large->d_Sb(), sizeof(int) * s->nV, hipMemcpyDefault));
    if (s->nC > 0) HIPCHK(hipMemcpy(s->d_wsc.p, s->large->d_Sc(), sizeof(int) * s->nC, hipMemcpyDefault));
    HIPCHK(s->d_x.download(s->h_x.data(), s->nV));
    HIPCHK(s->d_y.download(s->h_y.data(), s->nV + s->nC));
    HIPCHK(s->d_wsb.download(s->h_wsb.data(), s->nV));
    HIPCHK(s->d_wsc.download(s->h_wsc.data(), s->nC));
    s->status_word = s->large->status_word();
    s->last_ret = rc;
    s->last_nflips = s->large->nflips();
    s->obj = s->large->objective();
    return RSQP_OK;
}
}  // namespace

namespace { void launch_speculative_certificate(rsqp_solver *s, const QPPools &p); }

extern "C" int rsqp_solve(rsqp_solver *s, int mode, int *nWSR, const double *x0, const double *y0,
                          const int *guess_b) {
    if (!s || !nWSR || mode < 0 || mode > 3) return fail(RSQP_ERR_ARG, "rsqp_solve");
    s->spec_cert = false;
    s->last_mode = mode;
    if (!s->A.initialised && s->nC > 0) return fail(RSQP_ERR_ARG, "rsqp_solve: A not set");
    HIPCHK(hipSetDevice(s->device));
    int rc = flush_vectors(s);
    if (rc != RSQP_OK) return rc;
    rc = ensure_desc(s);
    if (rc != RSQP_OK) return rc;
    if (s->engine == 2) return solve_large(s, mode, nWSR, x0, y0, guess_b);
    const int nWSR_in = *nWSR;
    QPPools p = pools_of(s);
    if (mode == RSQP_MODE_WARM_REINIT) {
        if (x0) { HIPCHK(s->d_x0.upload(x0, s->nV)); p.x0 = s->d_x0.p; }
        if (y0) { HIPCHK(s->d_y0.upload(y0, s->nV + s->nC)); p.y0 = s->d_y0.p; }
        if (guess_b) { HIPCHK(s->d_guess.upload(guess_b, s->nV)); p.guess_b = s->d_guess.p; }
    }
    if (s->d_done && !s->kn.no_spin) { p.done_flag = s->d_done; p.done_val = ++s->done_seq; }
    // the hs071-scale tableau kernel forms the certificate QPhandler::solveQP asks for at the end of the SAME launch
    const bool fused_cert = p.done_flag && p.tiny_ok && rsqp_tiny_fits(s->kn, s->nV, s->nC) && !s->kn.no_spec_cert && !s->lp_mode &&
                            s->A.initialised == (s->nC > 0) && s->kn.engine < 0;
    if (fused_cert) { p.cert_out = s->d_kkt.p; p.cert_Wb = s->d_Wb.p; p.cert_Wc = s->d_Wc.p; }
    {   // the tableau kernel and the LDS-resident kernels keep different layouts in the same state block: a solve that changes
        // the family (H lost or regained its symmetry between two solves) starts cold instead of restoring the other's bytes
        const int fam = rsqp_small_launch_is_tiny(s->kn, p, s->nV, s->nC);
        if ((mode == RSQP_MODE_HOT_VECTORS || mode == RSQP_MODE_HOT_MATRICES) && s->state_engine != fam) { mode = RSQP_MODE_COLD; s->last_mode = mode; }
        s->state_engine = fam;
    }
    hipError_t e = rsqp_launch_small_qp(s->kn, p, 1, s->nV, s->nC,
                                        rsqp_mat_lds_bytes(s->nV, s->nC, s->A.initialised ? s->A.nnz : 0, s->H.initialised ? s->H.nnz : 0),
                                        mode, *nWSR, s->stream);
    if (e != hipSuccess) return fail(RSQP_ERR_DEVICE, std::string("QP kernel launch: ") + hipGetErrorString(e));
    if (fused_cert) { s->spec_cert = true; s->spec_cert_val = p.done_val; s->cert_pending = false; }
    else launch_speculative_certificate(s, p);
    // the results live in host-mapped memory and the kernel raises a host-mapped flag behind them: spinning on it saves
    // the ~10 us a blocking hipStreamSynchronize takes to wake up (a single hs071-scale solve is ~30 us end to end)
    if (p.done_flag) { if ((rc = wait_done(s, p.done_val)) != RSQP_OK) return rc; }
    else HIPCHK(hipStreamSynchronize(s->stream));
    rc = fetch_results(s);
    if (rc != RSQP_OK) return rc;
    if (s->state_engine == 1 && s->last_ret == RET_SETUP_FAILED) {
        // the register-resident tableau kernel gave up on a pivot in its rounding band (it has no hand-over of its own: the mid-size
        // tableau kernel bails to the null-space kernel inside the launcher): the LDS-resident Givens / TQ kernel takes the
        // same call over -- from scratch where the call wanted the tableau kernel's stored state
        SmallKnobs k2 = s->kn;
        k2.no_tiny = 1;
        int mode2 = (mode == RSQP_MODE_HOT_VECTORS || mode == RSQP_MODE_HOT_MATRICES) ? RSQP_MODE_COLD : mode;
        p.cert_out = nullptr; p.cert_Wb = nullptr; p.cert_Wc = nullptr;
        if (p.done_flag) p.done_val = ++s->done_seq;
        s->spec_cert = false; s->cert_pending = false;
        s->state_engine = 0;
        s->last_mode = mode2;
        int n2 = nWSR_in;
        e = rsqp_launch_small_qp(k2, p, 1, s->nV, s->nC,
                                 rsqp_mat_lds_bytes(s->nV, s->nC, s->A.initialised ? s->A.nnz : 0, s->H.initialised ? s->H.nnz : 0), mode2, n2, s->stream);
        if (e != hipSuccess) return fail(RSQP_ERR_DEVICE, std::string("QP kernel launch: ") + hipGetErrorString(e));
        if (p.done_flag) { if ((rc = wait_done(s, p.done_val)) != RSQP_OK) return rc; }
        else HIPCHK(hipStreamSynchronize(s->stream));
        rc = fetch_results(s);
        if (rc != RSQP_OK) return rc;
    }
    HIPCHK(s->d_nwsr.download(nWSR, 1));
    return RSQP_OK;
}

namespace {
// qpOASESInterface::handle_error, QP branch (src/qpOASESInterface.cpp:718-757)
int handle_error(rsqp_solver *s, int *total) {
    int nWSR = s->qp_maxiter, rc;
    if (infeasible(s) && s->nV >= 2 * s->nC) {
        std::vector<double> x0(s->nV, 0.0);
        for (int i = 0; i < s->nC; i++) {
            x0[i + s->nV - 2 * s->nC] = std::max(0.0, s->h_vec[RSQP_VEC_LBA][i]);
            x0[i + s->nV - s->nC] = -std::min(0.0, s->h_vec[RSQP_VEC_UBA][i]);
        }
        rc = rsqp_solve(s, RSQP_MODE_WARM_REINIT, &nWSR, x0.data(), nullptr, nullptr);
    } else {
        rc = rsqp_solve(s, RSQP_MODE_COLD, &nWSR, nullptr, nullptr, nullptr);
    }
    s->old_status = s->new_status = 0;
    *total += nWSR;
    return rc;  // the adapter throws QP_NOT_OPTIMAL when !rsqp_is_solved()
}
}  // namespace

extern "C" int rsqp_optimize_qp(rsqp_solver *s, int *nWSR_used) {
    if (!s) return fail(RSQP_ERR_ARG, "null solver");
    if (s->lp_mode) {
        // an LP was solved on this handle before (the reference keeps separate LP and QP objects,
        // Algorithm.cpp:561-562): the stored factors belong to H = regVal*I, so start over
        s->lp_mode = false; s->hreg = 0.0; s->desc_ready = false;
        s->firstQPsolved = false; s->old_status = s->new_status = 0;
    }
    int nWSR = s->qp_maxiter, total = 0, rc;
    if (!s->firstQPsolved) {
        rc = rsqp_solve(s, RSQP_MODE_COLD, &nWSR, nullptr, nullptr, nullptr);
        if (rc != RSQP_OK) return rc;
        if (solved(s)) s->firstQPsolved = true;
        else {
            // the reference leaves firstQPsolved_ false here even when the retry succeeds
            rc = handle_error(s, &total);
            if (rc != RSQP_OK) return rc;
            // still unsolved: the reference THROWS QP_NOT_OPTIMAL inside handle_error (:754-756) -- nothing behind the call runs: no
            // reset_flags, no second handle_error, and the first init's nWSR never reaches Stats::qp_iter (:211-212 are skipped), only
            // the retry's (:751-752). The adapter raises the exception from !rsqp_is_solved()
            if (!solved(s)) { if (nWSR_used) *nWSR_used = total; return RSQP_OK; }
        }
    } else {
        // get_Matrix_change_status (:817-833)
        const int cur = (s->upd_A || s->upd_H) ? 2 : 1;
        if (s->old_status == 0) s->old_status = cur;
        else {
            if (s->new_status != 0) s->old_status = s->new_status;
            s->new_status = cur;
        }
        if (s->new_status == 0)
            rc = rsqp_solve(s, s->old_status == 1 ? RSQP_MODE_HOT_VECTORS : RSQP_MODE_HOT_MATRICES, &nWSR, nullptr,
                            nullptr, nullptr);
        else if (s->new_status == 1 && s->old_status == 1)
            rc = rsqp_solve(s, RSQP_MODE_HOT_VECTORS, &nWSR, nullptr, nullptr, nullptr);
        else if (s->new_status == 2 && s->old_status == 2)
            rc = rsqp_solve(s, RSQP_MODE_HOT_MATRICES, &nWSR, nullptr, nullptr, nullptr);
        else {  // status flip: init(..., x_qp, y_qp, &bounds)  (:201-208)
            std::vector<double> x0 = s->h_x, y0 = s->h_y;
            std::vector<int> gb = s->h_wsb;
            rc = rsqp_solve(s, RSQP_MODE_WARM_REINIT, &nWSR, x0.data(), y0.data(), gb.data());
            s->new_status = s->old_status = 0;
        }
        if (rc != RSQP_OK) return rc;
    }
    s->upd_A = s->upd_H = s->upd_bounds = s->upd_g = false;  // reset_flags (:488-496)
    total += nWSR;
    if (!solved(s)) {
        rc = handle_error(s, &total);
        if (rc != RSQP_OK) return rc;
    }
    if (nWSR_used) *nWSR_used = total;
    return RSQP_OK;
}

namespace {
// qpOASESInterface::handle_error, LP branch (src/qpOASESInterface.cpp:688-717)
int handle_error_lp(rsqp_solver *s, int *total) {
    int nWSR = s->lp_maxiter, rc;
    if (infeasible(s) && s->nV >= 2 * s->nC) {
        std::vector<double> x0 = s->h_x;   // x_0 := x_qp, slack entries overwritten (:693-699)
        for (int i = 0; i < s->nC; i++) {
            x0[i + s->nV - 2 * s->nC] = std::max(0.0, s->h_vec[RSQP_VEC_LBA][i]);
            x0[i + s->nV - s->nC] = -std::min(0.0, s->h_vec[RSQP_VEC_UBA][i]);
        }
        rc = rsqp_solve(s, RSQP_MODE_WARM_REINIT, &nWSR, x0.data(), nullptr, nullptr);
    } else {
        rc = rsqp_solve(s, RSQP_MODE_COLD, &nWSR, nullptr, nullptr, nullptr);
    }
    s->old_status = s->new_status = 0;
    *total += nWSR;
    return rc;   // the adapter throws LP_NOT_OPTIMAL when !rsqp_is_solved()
}
}  // namespace

extern "C" int rsqp_optimize_lp(rsqp_solver *s, int *nWSR_used) {
    if (!s) return fail(RSQP_ERR_ARG, "null solver");
    if (!s->lp_mode) {   // a QP was solved on this handle before: its factors belong to another Hessian
        s->lp_mode = true; s->firstQPsolved = false; s->old_status = s->new_status = 0;
    }
    // qpOASES fixes regVal = getNorm(g) * epsRegularisation when the problem is initialised and keeps it
    // across hot starts (the factors stored in the engine were built with it); a later init -- the first
    // solve, a status flip, handle_error -- computes it afresh from the gradient of that call
    auto set_reg_for_init = [s]() {
        double ng = 0.0;
        for (double v : s->h_vec[RSQP_VEC_G]) ng += v * v;
        ng = std::sqrt(ng);
        s->hreg = (ng > 0.0 ? ng : 1.0) * 1.0e3 * RSQP_EPS;
        s->desc_ready = false;
    };
    int nWSR = s->lp_maxiter, total = 0, rc;
    if (!s->firstQPsolved) {
        set_reg_for_init();
        rc = rsqp_solve(s, RSQP_MODE_COLD, &nWSR, nullptr, nullptr, nullptr);
        if (rc != RSQP_OK) return rc;
        if (solved(s)) s->firstQPsolved = true;
        else {
            set_reg_for_init();
            if ((rc = handle_error_lp(s, &total)) != RSQP_OK) return rc;
            // (LP_NOT_OPTIMAL thrown inside handle_error, :714-716: the count behind the call, :278-279, is never added)
            if (!solved(s)) { if (nWSR_used) *nWSR_used = total; return RSQP_OK; }
        }
    } else {
        const int cur = (s->upd_A || s->upd_H) ? 2 : 1;
        if (s->old_status == 0) s->old_status = cur;
        else {
            if (s->new_status != 0) s->old_status = s->new_status;
            s->new_status = cur;
        }
        if (s->new_status == 0)
            rc = rsqp_solve(s, s->old_status == 1 ? RSQP_MODE_HOT_VECTORS : RSQP_MODE_HOT_MATRICES, &nWSR, nullptr,
                            nullptr, nullptr);
        else if (s->new_status == 1 && s->old_status == 1)
            rc = rsqp_solve(s, RSQP_MODE_HOT_VECTORS, &nWSR, nullptr, nullptr, nullptr);
        else if (s->new_status == 2 && s->old_status == 2)
            rc = rsqp_solve(s, RSQP_MODE_HOT_MATRICES, &nWSR, nullptr, nullptr, nullptr);
        else {   // :266-270: plain re-init on a status flip
            set_reg_for_init();
            rc = rsqp_solve(s, RSQP_MODE_COLD, &nWSR, nullptr, nullptr, nullptr);
            s->new_status = s->old_status = 0;
        }
        if (rc != RSQP_OK) return rc;
        s->upd_A = s->upd_H = s->upd_bounds = s->upd_g = false;
        if (!solved(s)) {
            set_reg_for_init();
            if ((rc = handle_error_lp(s, &total)) != RSQP_OK) return rc;
            if (!solved(s)) { if (nWSR_used) *nWSR_used = total; return RSQP_OK; }
        }
    }
    total += nWSR;
    if (solved(s)) {
        // one regularisation step: min (reg/2)|x - x_k|^2 + g'x  <=>  gradient g - reg*x_k
        const double reg = s->hreg;
        std::vector<double> gmod = s->h_vec[RSQP_VEC_G];
        for (int i = 0; i < s->nV; i++) gmod[i] -= reg * s->h_x[i];
        rc = flush_vectors(s);
        if (rc != RSQP_OK) return rc;
        HIPCHK(s->d_vec[RSQP_VEC_G].upload(gmod.data(), s->nV));
        int n2 = s->lp_maxiter;
        rc = rsqp_solve(s, RSQP_MODE_HOT_VECTORS, &n2, nullptr, nullptr, nullptr);
        s->vec_dirty = true;   // the true gradient goes back to the device with the next flush
        if (rc != RSQP_OK) return rc;
        total += n2;
        double o = 0.0;
        for (int i = 0; i < s->nV; i++) o += s->h_vec[RSQP_VEC_G][i] * s->h_x[i];
        s->obj = o;
    }
    if (nWSR_used) *nWSR_used = total;
    return RSQP_OK;
}

extern "C" int rsqp_get_primal(const rsqp_solver *s, double *x) {
    if (!s || !x) return fail(RSQP_ERR_ARG, "rsqp_get_primal");
    std::copy(s->h_x.begin(), s->h_x.end(), x);
    return RSQP_OK;
}
extern "C" int rsqp_get_dual(const rsqp_solver *s, double *y) {
    if (!s || !y) return fail(RSQP_ERR_ARG, "rsqp_get_dual");
    std::copy(s->h_y.begin(), s->h_y.end(), y);
    return RSQP_OK;
}
extern "C" double rsqp_get_objective(const rsqp_solver *s) { return s ? s->obj : 0.0; }
extern "C" int rsqp_get_status(const rsqp_solver *s) {
    return s ? exitflag_of(s->status_word, s->last_ret) : RSQP_QPERROR_UNKNOWN;
}
extern "C" int rsqp_is_solved(const rsqp_solver *s) { return s && solved(s); }
extern "C" int rsqp_get_working_set_raw(const rsqp_solver *s, int *ws_b, int *ws_c) {
    if (!s) return fail(RSQP_ERR_ARG, "null solver");
    if (ws_b) std::copy(s->h_wsb.begin(), s->h_wsb.end(), ws_b);
    if (ws_c) std::copy(s->h_wsc.begin(), s->h_wsc.end(), ws_c);
    return RSQP_OK;
}

namespace {
int spmv_csc(rsqp_solver *s, DevMatrix &M, const double *in, double *out) {  // out[col] = sum val*in[row]
    hipError_t e = rsqp_launch_spmv(M.blk_c.p, M.nblk_c, M.jc.p, M.ir.p, M.val.p, in, out, 1, 0, 0, 0, 0, s->stream);
    if (e != hipSuccess) return fail(RSQP_ERR_DEVICE, "spmv launch failed");
    return RSQP_OK;
}
int spmv_csr(rsqp_solver *s, DevMatrix &M, const double *in, double *out) {  // out[row] = sum val*in[col]
    hipError_t e = rsqp_launch_spmv(M.blk_r.p, M.nblk_r, M.rp.p, M.ci.p, M.rval.p, in, out, 1, 0, 0, 0, 0, s->stream);
    if (e != hipSuccess) return fail(RSQP_ERR_DEVICE, "spmv launch failed");
    return RSQP_OK;
}

int collect_certificate(rsqp_solver *s, rsqp_optimality_status *out, int *W_c, int *W_b, int *invalid) {
    double o[6];
    HIPCHK(s->d_kkt.download(o, 6));
    if (out) {
        out->primal_violation = o[0]; out->dual_violation = o[1]; out->compl_violation = o[2];
        out->stationarity_violation = o[3]; out->KKT_error = o[4];
    }
    *invalid = o[5] != 0.0;
    if (W_b) HIPCHK(s->d_Wb.download(W_b, s->nV));
    if (W_c) HIPCHK(s->d_Wc.download(W_c, s->nC));
    return RSQP_OK;
}
void fill_kkt_args(rsqp_solver *s, RsqpKktArgs &a) {
    std::memset(&a, 0, sizeof(a));
    a.nV1 = s->nV; a.nC1 = s->nC;
    a.x = s->d_x.p; a.y = s->d_y.p; a.g = s->d_vec[RSQP_VEC_G].p; a.lb = s->d_vec[RSQP_VEC_LB].p;
    a.ub = s->d_vec[RSQP_VEC_UB].p; a.lbA = s->d_vec[RSQP_VEC_LBA].p; a.ubA = s->d_vec[RSQP_VEC_UBA].p;
    a.Ax = s->d_Ax.p; a.ATy = s->d_ATy.p; a.Hx = s->d_Hx.p;
    a.ws_b = s->d_wsb.p; a.ws_c = s->d_wsc.p; a.W_b = s->d_Wb.p; a.W_c = s->d_Wc.p; a.out = s->d_kkt.p;
}

// the fused products + certificate kernel of an LDS-scale QP right behind its solve kernel (same stream), see rsqp_solver::spec_cert
void launch_speculative_certificate(rsqp_solver *s, const QPPools &p) {
    s->spec_cert = false;
    if (s->kn.no_spec_cert || s->lp_mode || !p.done_flag || !(s->fits_small && s->A.initialised == (s->nC > 0))) return;
    RsqpKktArgs a;
    fill_kkt_args(s, a);
    a.done_flag = s->d_done; a.done_val = ++s->done_seq;
    if (rsqp_launch_small_certificate(p, a, 1, s->d_Ax.p, s->d_ATy.p, s->d_Hx.p, s->stream) != hipSuccess) return;
    s->spec_cert = true; s->spec_cert_val = a.done_val; s->cert_pending = true;
}

int run_certificate(rsqp_solver *s, rsqp_optimality_status *out, int *W_c, int *W_b, int *invalid) {
    HIPCHK(hipSetDevice(s->device));
    if (s->spec_cert && !s->vec_dirty) {      // launched behind the solve, nothing changed since: only wait for it
        s->spec_cert = false;
        int rcw = wait_done(s, s->spec_cert_val);
        s->cert_pending = false;
        if (rcw != RSQP_OK) return rcw;
        return collect_certificate(s, out, W_c, W_b, invalid);
    }
    s->spec_cert = false;
    int rc = flush_vectors(s);
    if (rc != RSQP_OK) return rc;
    const bool fused = s->fits_small && s->A.initialised == (s->nC > 0);
    if (fused) {
        // hs0xx-scale: the three products and the certificate in ONE launch (one workgroup), below
        if ((rc = ensure_desc(s)) != RSQP_OK) return rc;
    } else if (s->nC > 0) {
        if ((rc = spmv_csr(s, s->A, s->d_x.p, s->d_Ax.p)) != RSQP_OK) return rc;             // A x
        if ((rc = spmv_csc(s, s->A, s->d_y.p + s->nV, s->d_ATy.p)) != RSQP_OK) return rc;    // A'y_c
    } else {
        HIPCHK(hipMemsetAsync(s->d_ATy.p, 0, sizeof(double) * s->nV, s->stream));
    }
    if (s->fits_small && s->A.initialised == (s->nC > 0)) {
    } else if (s->H.initialised) {
        if ((rc = spmv_csc(s, s->H, s->d_x.p, s->d_Hx.p)) != RSQP_OK) return rc;             // H x (symmetric)
    } else {
        HIPCHK(hipMemsetAsync(s->d_Hx.p, 0, sizeof(double) * s->nV, s->stream));
    }
    RsqpKktArgs a;
    fill_kkt_args(s, a);
    if (s->d_done && !s->kn.no_spin) { a.done_flag = s->d_done; a.done_val = ++s->done_seq; }
    if (fused) {
        QPPools p = pools_of(s);
        if (rsqp_launch_small_certificate(p, a, 1, s->d_Ax.p, s->d_ATy.p, s->d_Hx.p, s->stream) != hipSuccess)
            return fail(RSQP_ERR_DEVICE, "certificate launch failed");
    } else if (rsqp_launch_kkt(a, 1, s->stream) != hipSuccess) return fail(RSQP_ERR_DEVICE, "kkt launch failed");
    if (a.done_flag) { if ((rc = wait_done(s, a.done_val)) != RSQP_OK) return rc; }
    else HIPCHK(hipStreamSynchronize(s->stream));
    return collect_certificate(s, out, W_c, W_b, invalid);
}
}  // namespace

extern "C" int rsqp_get_working_set(rsqp_solver *s, int *W_c, int *W_b) {
    if (!s) return fail(RSQP_ERR_ARG, "null solver");
    int invalid = 0;
    int rc = run_certificate(s, nullptr, W_c, W_b, &invalid);
    if (rc != RSQP_OK) return rc;
    return invalid ? fail(RSQP_ERR_WORKING_SET, "INVALID_WORKING_SET") : RSQP_OK;
}

extern "C" int rsqp_test_optimality(rsqp_solver *s, int *W_c, int *W_b, rsqp_optimality_status *out) {
    if (!s) return fail(RSQP_ERR_ARG, "null solver");
    rsqp_optimality_status st;
    int invalid = 0;
    int rc = run_certificate(s, &st, W_c, W_b, &invalid);
    if (rc != RSQP_OK) return rc;
    if (out) *out = st;
    if (invalid) return fail(RSQP_ERR_WORKING_SET, "INVALID_WORKING_SET");
    return st.KKT_error > 1.0e-6 ? 0 : 1;  // :673
}

namespace {
int product(rsqp_solver *s, DevMatrix &M, bool use_csr, int nin, int nout, const double *p, double *result) {
    if (!s || !p || !result) return fail(RSQP_ERR_ARG, "product: null argument");
    if (!M.initialised) return fail(RSQP_ERR_ARG, "product: matrix not set");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(s->d_in.upload(p, nin));
    int rc = use_csr ? spmv_csr(s, M, s->d_in.p, s->d_out.p) : spmv_csc(s, M, s->d_in.p, s->d_out.p);
    if (rc != RSQP_OK) return rc;
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(s->d_out.download(result, nout));
    return RSQP_OK;
}
}  // namespace

extern "C" int rsqp_A_times(rsqp_solver *s, const double *p, double *result) {
    if (!s) return fail(RSQP_ERR_ARG, "null solver");
    return product(s, s->A, true, s->nV, s->nC, p, result);
}
extern "C" int rsqp_A_transposed_times(rsqp_solver *s, const double *p, double *result) {
    if (!s) return fail(RSQP_ERR_ARG, "null solver");
    return product(s, s->A, false, s->nC, s->nV, p, result);
}
extern "C" int rsqp_H_times(rsqp_solver *s, const double *p, double *result) {
    if (!s) return fail(RSQP_ERR_ARG, "null solver");
    return product(s, s->H, false, s->nV, s->nV, p, result);
}

// =====================================================================================
// batch of independent QPs
// =====================================================================================
struct rsqp_batch {
    int nq = 0, device = 0, nVmax = 0, nCmax = 0, uniV = -1, uniC = -1;
    bool uni_pat = false; int uni_annz = 0, uni_hnnz = 0; long long uni_state = 0;     // (QPPools::uni_pat)
    long long sumV = 0, sumC = 0, sumAnz = 0, sumHnz = 0, mat_bytes_max = 0;
    bool haveH = false;
    SmallKnobs kn = rsqp_small_knobs_from_env();
    int state_engine = -1;                // kernel family that wrote the members' hot-start states (see rsqp_solver::state_engine)
    int last_kernel = -1;                 // rsqp_batch_get_last_kernel
    bool h_sym = true;                    // every H symmetric value by value (the tableau kernel of qp_tiny.hip may take the batch)
    std::vector<int> h_Hjc, h_Hir;        // host copy of the H patterns (re-examined when the values change), small batches only
    std::vector<QPDesc> desc;
    std::vector<int> h_csr_perm;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    DevBuf<QPDesc> d_desc;
    DevBuf<int> Ajc, Air, Arp, Aci, perm, Hjc, Hir;
    DevBuf<double> Aval, Arv, Hval;
    DevBuf<double> g, lb, ub, lbA, ubA, x, y, obj, state;
    DevBuf<int> ws_b, ws_c, status, ret, nwsr, nflips;
    DevBuf<double> Ax, ATy, Hx, kkt;
    DevBuf<int> Wb, Wc, kV, kC;
    DevBuf<double> recbuf;   // rsqp_batch_pack_records_host
    DevBuf<long long> koV, koC;
    float last_ms = 0.f;
    bool keep_state = true;
    bool timing = false;   // between timer_start and timer_stop: no per-launch events (they cost ~10 us of stream time each)
    ~rsqp_batch() {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (ev2) (void)hipEventDestroy(ev2);
        if (ev3) (void)hipEventDestroy(ev3);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {
QPPools pools_of(rsqp_batch *b) {
    QPPools p;
    std::memset(&p, 0, sizeof(p));
    p.desc = b->d_desc.p;
    p.Ajc = b->Ajc.p; p.Air = b->Air.p; p.Aval = b->Aval.p;
    p.Arp = b->Arp.p; p.Aci = b->Aci.p; p.Arv = b->Arv.p;
    p.Hjc = b->Hjc.p; p.Hir = b->Hir.p; p.Hval = b->Hval.p;
    p.g = b->g.p; p.lb = b->lb.p; p.ub = b->ub.p; p.lbA = b->lbA.p; p.ubA = b->ubA.p;
    p.x = b->x.p; p.y = b->y.p; p.ws_b = b->ws_b.p; p.ws_c = b->ws_c.p;
    p.status = b->status.p; p.ret = b->ret.p; p.nwsr = b->nwsr.p; p.nflips = b->nflips.p;
    p.obj = b->obj.p; p.state = b->state.p;
    p.uniV = b->uniV; p.uniC = b->uniC;
    p.keep_state = b->keep_state ? 1 : 0;
    p.done_flag = nullptr; p.done_val = 0;
    p.tiny_ok = (b->h_sym || !b->haveH) ? 1 : 0;
    p.uni_pat = b->uni_pat ? 1 : 0;
    p.uni_annz = b->uni_annz; p.uni_hnnz = b->uni_hnnz; p.uni_haveH = b->haveH ? 1 : 0; p.uni_state = b->uni_state;
    return p;
}
}  // namespace

extern "C" int rsqp_batch_create(int nq, const int *nV, const int *nC, const int *Ajc, const int *Air,
                                 const double *Aval, const int *Hjc, const int *Hir, const double *Hval,
                                 int device, rsqp_batch **out) {
    if (!out || nq <= 0 || !nV || !nC || !Ajc) return fail(RSQP_ERR_ARG, "rsqp_batch_create");
    if (rsqp_device_count() <= 0) return fail(RSQP_ERR_DEVICE, "rsqp_batch_create: no HIP device visible");
    if (device >= 0) HIPCHK(hipSetDevice(device));
    rsqp_batch *b = new rsqp_batch();
    struct Guard { rsqp_batch *b; ~Guard() { delete b; } } guard{b};
    b->nq = nq;
    HIPCHK(hipGetDevice(&b->device));
    b->haveH = Hjc != nullptr;
    b->desc.resize(nq);
    std::vector<int> h_Arp, h_Aci, h_perm;
    long long offV = 0, offC = 0, offAjc = 0, offAnz = 0, offArp = 0, offHjc = 0, offHnz = 0, offState = 0;
    for (int q = 0; q < nq; q++) {
        if (nV[q] <= 0 || nC[q] < 0) return fail(RSQP_ERR_ARG, "rsqp_batch_create: bad sizes");
        QPDesc &d = b->desc[q];
        d.nV = nV[q]; d.nC = nC[q];
        d.offV = (int)offV; d.offC = (int)offC; d.offAjc = (int)offAjc; d.offAnz = (int)offAnz;
        d.offArp = (int)offArp; d.offHjc = (int)offHjc; d.offHnz = (int)offHnz; d.haveH = b->haveH;
        d.offState = offState;
        const int *jc = Ajc + offAjc;
        const int annz = jc[d.nV];
        d.annz = annz; d.hnnz = b->haveH ? Hjc[offHjc + d.nV] : 0;
        for (int k = 0; k < annz; k++)
            if (Air[offAnz + k] < 0 || Air[offAnz + k] >= d.nC) return fail(RSQP_ERR_ARG, "rsqp_batch_create: A row index");
        CsrCopy r;
        csr_from_csc(d.nC, d.nV, jc, Air + offAnz, r);
        h_Arp.insert(h_Arp.end(), r.rp.begin(), r.rp.end());
        h_Aci.insert(h_Aci.end(), r.ci.begin(), r.ci.end());
        for (int v : r.perm) h_perm.push_back((int)offAnz + v);
        b->nVmax = std::max(b->nVmax, d.nV); b->nCmax = std::max(b->nCmax, d.nC);
        if (q == 0) { b->uniV = d.nV; b->uniC = d.nC; }
        else { if (b->uniV != d.nV) b->uniV = -1; if (b->uniC != d.nC) b->uniC = -1; }
        offV += d.nV; offC += d.nC; offAjc += d.nV + 1; offAnz += annz; offArp += d.nC + 1;
        b->mat_bytes_max = std::max(b->mat_bytes_max, rsqp_mat_lds_bytes(d.nV, d.nC, annz, b->haveH ? Hjc[offHjc + d.nV] : 0));
        if (b->haveH) {
            const int hnnz = Hjc[offHjc + d.nV];
            for (int k = 0; k < hnnz; k++)
                if (Hir[offHnz + k] < 0 || Hir[offHnz + k] >= d.nV) return fail(RSQP_ERR_ARG, "rsqp_batch_create: H row index");
            offHjc += d.nV + 1; offHnz += hnnz;
        }
        offState += rsqp_state_bytes(d.nV, d.nC) / 8;
    }
    if (!rsqp_small_qp_fits(b->nVmax, b->nCmax))
        return fail(RSQP_ERR_TOO_LARGE, "rsqp_batch_create: a problem exceeds the LDS-resident engine");
    if (b->haveH && b->nVmax <= 8) {
        b->h_Hjc.assign(Hjc, Hjc + offHjc); b->h_Hir.assign(Hir, Hir + offHnz);
        for (int q = 0; q < nq && b->h_sym; q++) {
            const QPDesc &d = b->desc[q];
            b->h_sym = small_csc_symmetric(d.nV, Hjc + d.offHjc, Hir + d.offHnz, Hval + d.offHnz);
        }
    } else if (b->haveH) b->h_sym = false;
    b->sumV = offV; b->sumC = offC; b->sumAnz = offAnz; b->sumHnz = offHnz;
    // uniform batch: every member has the sizes and the patterns of member 0 (QPPools::uni_pat)
    b->uni_pat = b->uniV > 0 && b->uniC >= 0;
    if (b->uni_pat) {
        const QPDesc &d0 = b->desc[0];
        b->uni_annz = d0.annz; b->uni_hnnz = d0.hnnz; b->uni_state = rsqp_state_bytes(d0.nV, d0.nC) / 8;
        for (int q = 1; q < nq && b->uni_pat; q++) {
            const QPDesc &d = b->desc[q];
            b->uni_pat = d.annz == d0.annz && d.hnnz == d0.hnnz &&
                         std::memcmp(Ajc + d.offAjc, Ajc, sizeof(int) * (d0.nV + 1)) == 0 &&
                         std::memcmp(Air + d.offAnz, Air, sizeof(int) * d0.annz) == 0 &&
                         (!b->haveH || (std::memcmp(Hjc + d.offHjc, Hjc, sizeof(int) * (d0.nV + 1)) == 0 &&
                                        std::memcmp(Hir + d.offHnz, Hir, sizeof(int) * d0.hnnz) == 0));
        }
    }
    HIPCHK(hipStreamCreate(&b->stream));
    HIPCHK(hipEventCreate(&b->ev0)); HIPCHK(hipEventCreate(&b->ev1));
    HIPCHK(hipEventCreate(&b->ev2)); HIPCHK(hipEventCreate(&b->ev3));
    HIPCHK(b->d_desc.from(b->desc));
    HIPCHK(b->Ajc.alloc(offAjc)); HIPCHK(b->Ajc.upload(Ajc, offAjc));
    HIPCHK(b->Air.alloc(offAnz)); HIPCHK(b->Air.upload(Air, offAnz));
    HIPCHK(b->Aval.alloc(offAnz)); HIPCHK(b->Aval.upload(Aval, offAnz));
    HIPCHK(b->Arp.alloc(offArp)); HIPCHK(b->Arp.upload(h_Arp.data(), h_Arp.size()));
    HIPCHK(b->Aci.alloc(offAnz)); HIPCHK(b->Aci.upload(h_Aci.data(), h_Aci.size()));
    HIPCHK(b->perm.alloc(offAnz)); HIPCHK(b->perm.upload(h_perm.data(), h_perm.size()));
    HIPCHK(b->Arv.alloc(offAnz));
    if (rsqp_launch_gather((int)offAnz, b->perm.p, b->Aval.p, b->Arv.p, b->stream) != hipSuccess)
        return fail(RSQP_ERR_DEVICE, "gather launch failed");
    HIPCHK(b->Hjc.alloc(b->haveH ? offHjc : 2)); HIPCHK(b->Hir.alloc(offHnz)); HIPCHK(b->Hval.alloc(offHnz));
    if (b->haveH) {
        HIPCHK(b->Hjc.upload(Hjc, offHjc)); HIPCHK(b->Hir.upload(Hir, offHnz)); HIPCHK(b->Hval.upload(Hval, offHnz));
    }
    HIPCHK(b->g.alloc(offV)); HIPCHK(b->lb.alloc(offV)); HIPCHK(b->ub.alloc(offV));
    HIPCHK(b->lbA.alloc(offC)); HIPCHK(b->ubA.alloc(offC));
    HIPCHK(b->x.alloc(offV)); HIPCHK(b->y.alloc(offV + offC)); HIPCHK(b->obj.alloc(nq));
    HIPCHK(b->ws_b.alloc(offV)); HIPCHK(b->ws_c.alloc(offC));
    HIPCHK(b->status.alloc(nq)); HIPCHK(b->ret.alloc(nq)); HIPCHK(b->nwsr.alloc(nq)); HIPCHK(b->nflips.alloc(nq));
    HIPCHK(b->state.alloc((size_t)offState));
    HIPCHK(hipStreamSynchronize(b->stream));
    guard.b = nullptr;
    *out = b;
    return RSQP_OK;
}

extern "C" void rsqp_batch_destroy(rsqp_batch *b) { delete b; }

extern "C" int rsqp_batch_set_vectors(rsqp_batch *b, const double *g, const double *lb, const double *ub,
                                      const double *lbA, const double *ubA) {
    if (!b || !g || !lb || !ub || (b->sumC > 0 && (!lbA || !ubA))) return fail(RSQP_ERR_ARG, "rsqp_batch_set_vectors");
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(b->g.upload(g, b->sumV)); HIPCHK(b->lb.upload(lb, b->sumV)); HIPCHK(b->ub.upload(ub, b->sumV));
    HIPCHK(b->lbA.upload(lbA, b->sumC)); HIPCHK(b->ubA.upload(ubA, b->sumC));
    return RSQP_OK;
}

extern "C" int rsqp_batch_set_matrix_values(rsqp_batch *b, const double *Aval, const double *Hval) {
    if (!b) return fail(RSQP_ERR_ARG, "null batch");
    HIPCHK(hipSetDevice(b->device));
    if (Aval) {
        HIPCHK(b->Aval.upload(Aval, b->sumAnz));
        if (rsqp_launch_gather((int)b->sumAnz, b->perm.p, b->Aval.p, b->Arv.p, b->stream) != hipSuccess)
            return fail(RSQP_ERR_DEVICE, "gather launch failed");
    }
    if (Hval && b->haveH) {
        HIPCHK(b->Hval.upload(Hval, b->sumHnz));
        if (!b->h_Hjc.empty()) {
            b->h_sym = true;
            for (int q = 0; q < b->nq && b->h_sym; q++) {
                const QPDesc &d = b->desc[q];
                b->h_sym = small_csc_symmetric(d.nV, b->h_Hjc.data() + d.offHjc, b->h_Hir.data() + d.offHnz, Hval + d.offHnz);
            }
        }
    }
    HIPCHK(hipStreamSynchronize(b->stream));
    return RSQP_OK;
}

extern "C" int rsqp_batch_solve(rsqp_batch *b, int mode, int max_nWSR) {
    if (!b || mode < 0 || mode > 2 || max_nWSR < 0) return fail(RSQP_ERR_ARG, "rsqp_batch_solve");
    HIPCHK(hipSetDevice(b->device));
    QPPools p = pools_of(b);
    if (!b->timing) HIPCHK(hipEventRecord(b->ev0, b->stream));
    {
        const int fam = rsqp_small_launch_is_tiny(b->kn, p, b->nVmax, b->nCmax);
        if ((mode == RSQP_MODE_HOT_VECTORS || mode == RSQP_MODE_HOT_MATRICES) && b->state_engine != fam) mode = RSQP_MODE_COLD;
        b->state_engine = fam;
        // a cold-start-only batch on the tableau kernel keeps no state and leaves no mark: the handle remembers it instead
        if (fam == 1 && !b->keep_state) { p.skip_mark = 1; b->state_engine = -1; }
        b->last_kernel = fam == 1 ? (rsqp_lane_fits(b->kn, p, b->nq, b->nVmax, b->nCmax, mode) ? 2 : 1) : 0;
    }
    hipError_t e = rsqp_launch_small_qp(b->kn, p, b->nq, b->nVmax, b->nCmax, b->mat_bytes_max, mode, max_nWSR, b->stream);
    if (e != hipSuccess) return fail(RSQP_ERR_DEVICE, std::string("QP kernel launch: ") + hipGetErrorString(e));
    if (!b->timing) HIPCHK(hipEventRecord(b->ev1, b->stream));
    return RSQP_OK;
}

extern "C" int rsqp_batch_get_last_kernel(const rsqp_batch *b) { return b ? b->last_kernel : -1; }

extern "C" int rsqp_batch_set_keep_state(rsqp_batch *b, int keep) {
    if (!b) return fail(RSQP_ERR_ARG, "null batch");
    b->keep_state = keep != 0;
    return RSQP_OK;
}

extern "C" int rsqp_batch_sync(rsqp_batch *b) {
    if (!b) return fail(RSQP_ERR_ARG, "null batch");
    HIPCHK(hipStreamSynchronize(b->stream));
    return RSQP_OK;
}

extern "C" float rsqp_batch_last_solve_ms(rsqp_batch *b) {
    if (!b) return -1.f;
    if (hipEventSynchronize(b->ev1) != hipSuccess) return -1.f;
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, b->ev0, b->ev1) != hipSuccess) return -1.f;
    b->last_ms = ms;
    return ms;
}

extern "C" int rsqp_batch_timer_start(rsqp_batch *b) {
    if (!b) return fail(RSQP_ERR_ARG, "null batch");
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipEventRecord(b->ev2, b->stream));
    b->timing = true;
    return RSQP_OK;
}
extern "C" float rsqp_batch_timer_stop_ms(rsqp_batch *b) {
    if (!b) return -1.f;
    float ms = -1.f;
    b->timing = false;
    if (hipEventRecord(b->ev3, b->stream) != hipSuccess) return -1.f;
    if (hipEventSynchronize(b->ev3) != hipSuccess) return -1.f;
    if (hipEventElapsedTime(&ms, b->ev2, b->ev3) != hipSuccess) return -1.f;
    return ms;
}

extern "C" int rsqp_batch_get_results(rsqp_batch *b, double *x, double *y, int *ws_b, int *ws_c, int *status,
                                      int *nWSR, double *obj) {
    if (!b) return fail(RSQP_ERR_ARG, "null batch");
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipStreamSynchronize(b->stream));
    if (x) HIPCHK(b->x.download(x, b->sumV));
    if (y) HIPCHK(b->y.download(y, b->sumV + b->sumC));
    if (ws_b) HIPCHK(b->ws_b.download(ws_b, b->sumV));
    if (ws_c) HIPCHK(b->ws_c.download(ws_c, b->sumC));
    if (status) {
        std::vector<int> sw(b->nq), rt(b->nq);
        HIPCHK(b->status.download(sw.data(), b->nq)); HIPCHK(b->ret.download(rt.data(), b->nq));
        for (int q = 0; q < b->nq; q++) status[q] = exitflag_of(sw[q], rt[q]);
    }
    if (nWSR) HIPCHK(b->nwsr.download(nWSR, b->nq));
    if (obj) HIPCHK(b->obj.download(obj, b->nq));
    return RSQP_OK;
}

extern "C" int rsqp_batch_test_optimality(rsqp_batch *b, rsqp_optimality_status *out, int *ok) {
    if (!b) return fail(RSQP_ERR_ARG, "null batch");
    HIPCHK(hipSetDevice(b->device));
    if (!b->Ax.p) {
        HIPCHK(b->Ax.alloc(b->sumC)); HIPCHK(b->ATy.alloc(b->sumV)); HIPCHK(b->Hx.alloc(b->sumV));
        HIPCHK(b->kkt.alloc(6 * (size_t)b->nq)); HIPCHK(b->Wb.alloc(b->sumV)); HIPCHK(b->Wc.alloc(b->sumC));
        std::vector<int> kv(b->nq), kc(b->nq);
        std::vector<long long> ov(b->nq), oc(b->nq);
        for (int q = 0; q < b->nq; q++) {
            kv[q] = b->desc[q].nV; kc[q] = b->desc[q].nC; ov[q] = b->desc[q].offV; oc[q] = b->desc[q].offC;
        }
        HIPCHK(b->kV.from(kv)); HIPCHK(b->kC.from(kc)); HIPCHK(b->koV.from(ov)); HIPCHK(b->koC.from(oc));
    }
    QPPools p = pools_of(b);
    RsqpKktArgs a;
    std::memset(&a, 0, sizeof(a));
    a.nV = b->kV.p; a.nC = b->kC.p; a.offV = b->koV.p; a.offC = b->koC.p;
    a.x = b->x.p; a.y = b->y.p; a.g = b->g.p; a.lb = b->lb.p; a.ub = b->ub.p; a.lbA = b->lbA.p; a.ubA = b->ubA.p;
    a.Ax = b->Ax.p; a.ATy = b->ATy.p; a.Hx = b->Hx.p; a.ws_b = b->ws_b.p; a.ws_c = b->ws_c.p;
    a.W_b = b->Wb.p; a.W_c = b->Wc.p; a.out = b->kkt.p;
    if (rsqp_launch_small_certificate(p, a, b->nq, b->Ax.p, b->ATy.p, b->Hx.p, b->stream) != hipSuccess)
        return fail(RSQP_ERR_DEVICE, "certificate launch failed");
    HIPCHK(hipStreamSynchronize(b->stream));
    std::vector<double> o(6 * (size_t)b->nq);
    HIPCHK(b->kkt.download(o.data(), o.size()));
    for (int q = 0; q < b->nq; q++) {
        if (out) {
            out[q].primal_violation = o[6 * q]; out[q].dual_violation = o[6 * q + 1];
            out[q].compl_violation = o[6 * q + 2]; out[q].stationarity_violation = o[6 * q + 3];
            out[q].KKT_error = o[6 * q + 4];
        }
        if (ok) ok[q] = o[6 * q + 5] != 0.0 ? RSQP_ERR_WORKING_SET : (o[6 * q + 4] > 1.0e-6 ? 0 : 1);
    }
    return RSQP_OK;
}

// ---------------------------------------------------------------------------------
// fixed-stride result records on the device: what a rank contributes to the gather of a sharded batch
// (SURVEY 8(e)). Layout = restartsqp_amd/parallel.py pack_records:
//   {exitflag, nWSR, objective, KKT_error, x[nVmax], y_bounds[nVmax], y_constr[nCmax], ws_b[nVmax], ws_c[nCmax]}
// One thread per record entry; unused tail entries of a smaller problem are zero.
// ---------------------------------------------------------------------------------
namespace {
__global__ void pack_records_kernel(int nq, int nVmax, int nCmax, const QPDesc *__restrict__ desc,
                                    const double *__restrict__ x, const double *__restrict__ y,
                                    const int *__restrict__ ws_b, const int *__restrict__ ws_c,
                                    const int *__restrict__ status, const int *__restrict__ nwsr,
                                    const double *__restrict__ obj, const double *__restrict__ kkt,
                                    double *__restrict__ rec) {
    const int stride = 4 + 3 * nVmax + 2 * nCmax;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)nq * stride) return;
    const int q = (int)(t / stride), e = (int)(t % stride);
    const QPDesc d = desc[q];
    double v = 0.0;
    if (e < 4) {
        if (e == 0) {   // exitflag_of(): qpOASESInterface::get_status (src/qpOASESInterface.cpp:332-357)
            const int sw = status[q];
            v = sw >= 200 ? RSQP_QPERROR_UNBOUNDED : (sw >= 100 ? RSQP_QPERROR_INFEASIBLE : (sw == QPS_SOLVED ? RSQP_QP_OPTIMAL : RSQP_QPERROR_NOTINITIALISED + sw));
        } else if (e == 1) v = nwsr[q];
        else if (e == 2) v = obj[q];
        else v = kkt ? kkt[6 * q + 4] : 0.0;
    } else {
        int o = e - 4;
        if (o < nVmax) { if (o < d.nV) v = x[d.offV + o]; }
        else if ((o -= nVmax) < nVmax) { if (o < d.nV) v = y[d.offV + d.offC + o]; }
        else if ((o -= nVmax) < nCmax) { if (o < d.nC) v = y[d.offV + d.offC + d.nV + o]; }
        else if ((o -= nCmax) < nVmax) { if (o < d.nV) v = ws_b[d.offV + o]; }
        else { o -= nVmax; if (o < d.nC) v = ws_c[d.offC + o]; }
    }
    rec[t] = v;
}
}  // namespace

extern "C" int rsqp_batch_record_stride(const rsqp_batch *b) {
    return b ? 4 + 3 * b->nVmax + 2 * b->nCmax : -1;
}

extern "C" int rsqp_batch_pack_records_dev(rsqp_batch *b, double *rec_dev) {
    if (!b || !rec_dev) return fail(RSQP_ERR_ARG, "rsqp_batch_pack_records_dev");
    HIPCHK(hipSetDevice(b->device));
    const long long tot = (long long)b->nq * rsqp_batch_record_stride(b);
    hipLaunchKernelGGL(pack_records_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, b->stream, b->nq, b->nVmax,
                       b->nCmax, b->d_desc.p, b->x.p, b->y.p, b->ws_b.p, b->ws_c.p, b->status.p, b->nwsr.p, b->obj.p,
                       b->kkt.p /* null until the certificate has run */, rec_dev);
    HIPCHK(hipGetLastError());
    return RSQP_OK;
}

extern "C" int rsqp_batch_pack_records_host(rsqp_batch *b, double *rec_host) {
    if (!b || !rec_host) return fail(RSQP_ERR_ARG, "rsqp_batch_pack_records_host");
    HIPCHK(hipSetDevice(b->device));
    const size_t tot = (size_t)b->nq * rsqp_batch_record_stride(b);
    if (b->recbuf.n < tot) HIPCHK(b->recbuf.alloc(tot, false));
    int rc = rsqp_batch_pack_records_dev(b, b->recbuf.p);
    if (rc != RSQP_OK) return rc;
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(b->recbuf.download(rec_host, tot));
    return RSQP_OK;
}

// (rsqp_rccl.cpp: the native RCCL call sites reach the batch through these)
int rsqp_fail_msg(int code, const char *msg) { return fail(code, msg ? msg : ""); }
hipStream_t rsqp_batch_stream_internal(rsqp_batch *b) { return b->stream; }
int rsqp_batch_device_internal(const rsqp_batch *b) { return b->device; }
int rsqp_batch_nq_internal(const rsqp_batch *b) { return b ? b->nq : 0; }

// setMatVal on the device (SpHbMat.cpp:368-393): time `repeats` launches of the scatter through `order`
// and of the CSR-copy gather on the values already staged by rsqp_set_A_triplet (no host transfer inside)
extern "C" int rsqp_time_value_refresh_fused(rsqp_solver *s, int repeats, float *ms) {
    if (!s || repeats <= 0 || !ms || !s->A.initialised || !s->A.from_triplet || !s->A.have_csr) return fail(RSQP_ERR_ARG, "rsqp_time_value_refresh_fused");
    if (s->A.pin) return fail(RSQP_ERR_ARG, "rsqp_time_value_refresh_fused: this handle refreshes its values on the host (no kernel to time)");
    HIPCHK(hipSetDevice(s->device));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    DevMatrix &M = s->A;
    HIPCHK(hipEventRecord(e0, s->stream));
    for (int r = 0; r < repeats; r++)
        if (rsqp_launch_scatter_csc_csr(M.n_triplet, M.order.p, M.rorder.p, M.tv.p, M.val.p, M.rval.p, s->stream) != hipSuccess)
            return fail(RSQP_ERR_DEVICE, "value refresh launch failed");
    HIPCHK(hipEventRecord(e1, s->stream));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipEventElapsedTime(ms, e0, e1));
    *ms /= repeats;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return RSQP_OK;
}

extern "C" int rsqp_time_value_refresh(rsqp_solver *s, int repeats, float *ms_scatter, float *ms_gather) {
    if (!s || repeats <= 0 || !s->A.initialised || !s->A.from_triplet) return fail(RSQP_ERR_ARG, "rsqp_time_value_refresh");
    if (s->A.pin) return fail(RSQP_ERR_ARG, "rsqp_time_value_refresh: this handle refreshes its values on the host (no kernel to time)");
    HIPCHK(hipSetDevice(s->device));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    DevMatrix &M = s->A;
    for (int pass = 0; pass < 2; pass++) {
        float *out = pass == 0 ? ms_scatter : ms_gather;
        HIPCHK(hipEventRecord(e0, s->stream));
        for (int r = 0; r < repeats; r++) {
            hipError_t e = pass == 0 ? rsqp_launch_scatter(M.n_triplet, M.order.p, nullptr, M.tv.p, M.val.p, s->stream)
                                     : rsqp_launch_gather(M.nnz, M.perm.p, M.val.p, M.rval.p, s->stream);
            if (e != hipSuccess) return fail(RSQP_ERR_DEVICE, "value refresh launch failed");
        }
        HIPCHK(hipEventRecord(e1, s->stream));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        if (out) *out = ms / repeats;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return RSQP_OK;
}

extern "C" int rsqp_time_large_kernel(int device, int n, int kind, int nrows, int ncols, int repeats, float *ms) {
    if (n <= 0 || !ms) return fail(RSQP_ERR_ARG, "rsqp_time_large_kernel");
    HIPCHK(hipSetDevice(device));
    RsqpLargeEngine eng;
    hipStream_t st = nullptr;
    HIPCHK(hipStreamCreate(&st));
    hipError_t e = eng.init(n, 0, st);
    int rc = e == hipSuccess ? eng.time_kernel(kind, nrows, ncols, repeats, ms) : -1;
    (void)hipStreamSynchronize(st);
    (void)hipStreamDestroy(st);
    if (e != hipSuccess) return fail(RSQP_ERR_DEVICE, std::string("rsqp_time_large_kernel: ") + hipGetErrorString(e));
    return rc == 0 ? RSQP_OK : fail(RSQP_ERR_ARG, "rsqp_time_large_kernel: bad shape or launch failure");
}

// =====================================================================================
// batched SpMV plan (device resident)
// =====================================================================================
struct rsqp_spmv_plan {
    int nrow = 0, ncol = 0, nnz = 0, nbatch = 0, device = 0;
    int nblk_c = 0, nblk_r = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // every member owns a full copy of the index arrays: distinct HBM traffic per matrix
    DevBuf<int> jc, ir, rp, ci, perm;
    DevBuf<int4> blk_c, blk_r;
    DevBuf<int> slice_c, slice_r;   // one slice = all majors (1 workgroup per member)
    DevBuf<unsigned short> ir16, ci16;  // 16-bit index copies (vector length < 65536)
    // entry-parallel variant 40: per orientation (t: CSC majors = columns, n: CSR majors = rows)
    DevBuf<int4> seg_chunks[2];
    DevBuf<int> seg_nz[2], seg_empt[2];
    DevBuf<unsigned> seg_bits[2];       // one copy per member, like the index arrays
    int seg_nchunks[2] = {0, 0}, seg_nempty[2] = {0, 0};
    bool seg_ok[2] = {false, false};
    bool use16 = false;
    int variant_t = 0, variant_n = 0;  // 0: stream kernel; >0: LDS-resident-vector kernel
    int nslices = 1;
    DevBuf<double> val, rval, vin_r, vin_c, vout_r, vout_c;  // _r: length nrow, _c: length ncol
    ~rsqp_spmv_plan() {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

extern "C" int rsqp_spmv_plan_create(int nrow, int ncol, const int *jc, const int *ir, int nbatch, int device,
                                     rsqp_spmv_plan **out) {
    if (!out || nrow <= 0 || ncol <= 0 || !jc || !ir || nbatch <= 0) return fail(RSQP_ERR_ARG, "rsqp_spmv_plan_create");
    if (rsqp_device_count() <= 0) return fail(RSQP_ERR_DEVICE, "rsqp_spmv_plan_create: no HIP device visible");
    if (device >= 0) HIPCHK(hipSetDevice(device));
    rsqp_spmv_plan *p = new rsqp_spmv_plan();
    struct Guard { rsqp_spmv_plan *p; ~Guard() { delete p; } } guard{p};
    p->nrow = nrow; p->ncol = ncol; p->nnz = jc[ncol]; p->nbatch = nbatch;
    HIPCHK(hipGetDevice(&p->device));
    HIPCHK(hipStreamCreate(&p->stream));
    HIPCHK(hipEventCreate(&p->ev0)); HIPCHK(hipEventCreate(&p->ev1));
    CsrCopy r;
    csr_from_csc(nrow, ncol, jc, ir, r);
    std::vector<int4> bc = build_blocks(ncol, jc, rsqp_spmv_chunk()), br = build_blocks(nrow, r.rp.data(), rsqp_spmv_chunk());
    p->nblk_c = (int)bc.size(); p->nblk_r = (int)br.size();
    HIPCHK(p->blk_c.from(bc)); HIPCHK(p->blk_r.from(br));
    {
        const char *es = getenv("RSQP_SPMV_SLICES");
        p->nslices = es ? std::max(1, atoi(es)) : 1;
        auto cut = [&](int nmajor, const int *pt) {
            std::vector<int> sl(p->nslices + 1, nmajor);
            sl[0] = 0;
            for (int q = 1; q < p->nslices; q++) {   // equal share of entries per slice
                long long target = (long long)pt[nmajor] * q / p->nslices;
                sl[q] = (int)(std::lower_bound(pt, pt + nmajor + 1, (int)target) - pt);
            }
            return sl;
        };
        std::vector<int> sc = cut(ncol, jc), sr = cut(nrow, r.rp.data());
        HIPCHK(p->slice_c.from(sc)); HIPCHK(p->slice_r.from(sr));
        // default kernel choice: vector in LDS when it fits and the batch can fill the chip
        const char *ev = getenv("RSQP_SPMV_VARIANT");
        int forced = ev ? atoi(ev) : -1;
        bool fits_t = (size_t)nrow * 8 + 16 <= 160 * 1024, fits_n = (size_t)ncol * 8 + 16 <= 160 * 1024;
        // lanes per major by average segment length (measured on MI355X, tools/spmv_sweep.py):
        // >= 16 entries: 4 lanes x 2 entries x 3 steps; shorter: 2 lanes x 2 entries x 4 steps
        auto pick = [](double avg) { return avg >= 16.0 ? 35 : 38; };
        p->variant_t = forced >= 0 ? forced : (fits_t && nbatch >= 64 ? pick((double)p->nnz / ncol) : 0);
        p->variant_n = forced >= 0 ? forced : (fits_n && nbatch >= 64 ? pick((double)p->nnz / nrow) : 0);
        if (!fits_t) p->variant_t = 0;
        if (!fits_n) p->variant_n = 0;
    }
    const size_t B = nbatch, nnz = p->nnz;
    HIPCHK(p->jc.alloc(B * (ncol + 1), false)); HIPCHK(p->ir.alloc(B * nnz + 2, false));
    HIPCHK(p->rp.alloc(B * (nrow + 1), false)); HIPCHK(p->ci.alloc(B * nnz + 2, false));
    HIPCHK(p->perm.from(r.perm));
    for (size_t m = 0; m < B; m++) {
        HIPCHK(hipMemcpy(p->jc.p + m * (ncol + 1), jc, sizeof(int) * (ncol + 1), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(p->ir.p + m * nnz, ir, sizeof(int) * nnz, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(p->rp.p + m * (nrow + 1), r.rp.data(), sizeof(int) * (nrow + 1), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(p->ci.p + m * nnz, r.ci.data(), sizeof(int) * nnz, hipMemcpyHostToDevice));
    }
    HIPCHK(p->val.alloc(B * nnz + 528)); HIPCHK(p->rval.alloc(B * nnz + 528));   // variant 40 reads (and masks) up to 520 entries past a chunk
    {
        const char *e16 = getenv("RSQP_SPMV_IDX16");
        p->use16 = nrow < 65536 && ncol < 65536 && !(e16 && atoi(e16) == 0);
        if (p->use16) {
            std::vector<unsigned short> a16(ir, ir + nnz), c16(r.ci.begin(), r.ci.end());
            HIPCHK(p->ir16.alloc(B * nnz + 528, true)); HIPCHK(p->ci16.alloc(B * nnz + 528, true));
            for (size_t m = 0; m < B; m++) {
                HIPCHK(hipMemcpy(p->ir16.p + m * nnz, a16.data(), 2 * nnz, hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(p->ci16.p + m * nnz, c16.data(), 2 * nnz, hipMemcpyHostToDevice));
            }
        }
    }
    // entry-parallel kernel (variant 40): needs the 16-bit indices and majors of at most 512 entries; preferred over the
    // sub-wave-per-major kernels wherever those would be chosen (see the measurements below)
    if (p->use16) {
        const char *ev = getenv("RSQP_SPMV_VARIANT");
        const int forced = ev ? atoi(ev) : -1;
        for (int o = 0; o < 2; o++) {
            SegHost h = build_seg(o == 0 ? ncol : nrow, o == 0 ? jc : r.rp.data());
            if (!h.ok || h.chunks.empty()) continue;
            HIPCHK(p->seg_chunks[o].from(h.chunks)); HIPCHK(p->seg_nz[o].from(h.nz));
            HIPCHK(p->seg_empt[o].alloc(std::max<size_t>(h.empties.size(), 1))); HIPCHK(p->seg_empt[o].upload(h.empties.data(), h.empties.size()));
            HIPCHK(p->seg_bits[o].alloc(B * h.bits.size(), false));
            for (size_t m = 0; m < B; m++)
                HIPCHK(hipMemcpy(p->seg_bits[o].p + m * h.bits.size(), h.bits.data(), 4 * h.bits.size(), hipMemcpyHostToDevice));
            p->seg_nchunks[o] = (int)h.chunks.size(); p->seg_nempty[o] = (int)h.empties.size();
            p->seg_ok[o] = true;
            // measured on the 10k x 20k shape (tools/spmv_bound_check.py): majors of ~20 entries 0.120 ms vs 0.133 ms for the
            // sub-wave kernel <4,3>; majors of ~10 entries 0.128 vs 0.136 ms for <2,4>. Majors shorter than ~8 entries
            // (the [J I -I] columns) would leave the 128-major chunks mostly empty: those stay with <2,4>
            int &var = o == 0 ? p->variant_t : p->variant_n;
            const double avg = (double)p->nnz / std::max<size_t>(h.nz.size(), 1);
            if (forced < 0 ? (var == 35 || (var == 38 && avg >= 8.0)) : forced == 40) var = 40;
        }
    }
    if (p->variant_t == 40 && !p->seg_ok[0]) p->variant_t = 0;
    if (p->variant_n == 40 && !p->seg_ok[1]) p->variant_n = 0;
    HIPCHK(p->vin_r.alloc(B * nrow)); HIPCHK(p->vin_c.alloc(B * ncol));
    HIPCHK(p->vout_r.alloc(B * nrow)); HIPCHK(p->vout_c.alloc(B * ncol));
    guard.p = nullptr;
    *out = p;
    return RSQP_OK;
}

extern "C" void rsqp_spmv_plan_destroy(rsqp_spmv_plan *p) { delete p; }

extern "C" int rsqp_spmv_plan_upload(rsqp_spmv_plan *p, const double *vals, const double *xin, int transposed) {
    if (!p) return fail(RSQP_ERR_ARG, "null plan");
    HIPCHK(hipSetDevice(p->device));
    const size_t B = p->nbatch;
    if (vals) {
        HIPCHK(p->val.upload(vals, B * p->nnz));
        for (size_t m = 0; m < B; m++)
            if (rsqp_launch_gather(p->nnz, p->perm.p, p->val.p + m * p->nnz, p->rval.p + m * p->nnz, p->stream) != hipSuccess)
                return fail(RSQP_ERR_DEVICE, "gather launch failed");
        HIPCHK(hipStreamSynchronize(p->stream));
    }
    if (xin) {
        if (transposed) HIPCHK(p->vin_r.upload(xin, B * p->nrow));
        else HIPCHK(p->vin_c.upload(xin, B * p->ncol));
    }
    return RSQP_OK;
}

extern "C" int rsqp_spmv_plan_run(rsqp_spmv_plan *p, int transposed, int repeats, float *ms_per_launch) {
    if (!p || repeats <= 0) return fail(RSQP_ERR_ARG, "rsqp_spmv_plan_run");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipEventRecord(p->ev0, p->stream));
    for (int r = 0; r < repeats; r++) {
        hipError_t e;
        if (transposed && p->variant_t == 40)
            e = rsqp_launch_spmv_segscan(p->nrow, p->seg_chunks[0].p, p->seg_nchunks[0], p->seg_nz[0].p, p->seg_empt[0].p, p->seg_nempty[0],
                                         p->seg_bits[0].p, p->ir16.p, p->val.p, p->vin_r.p, p->vout_c.p, p->nbatch, p->nnz, p->nrow, p->ncol, p->stream);
        else if (!transposed && p->variant_n == 40)
            e = rsqp_launch_spmv_segscan(p->ncol, p->seg_chunks[1].p, p->seg_nchunks[1], p->seg_nz[1].p, p->seg_empt[1].p, p->seg_nempty[1],
                                         p->seg_bits[1].p, p->ci16.p, p->rval.p, p->vin_c.p, p->vout_r.p, p->nbatch, p->nnz, p->ncol, p->nrow, p->stream);
        else if (transposed && p->variant_t > 0)
            e = rsqp_launch_spmv_ldsvec(p->variant_t, p->nrow, p->nslices, p->slice_c.p, p->jc.p, p->ir.p, p->use16 ? p->ir16.p : nullptr, p->val.p, p->vin_r.p,
                                        p->vout_c.p, p->nbatch, p->ncol + 1, p->nnz, p->nrow, p->ncol, p->stream);
        else if (!transposed && p->variant_n > 0)
            e = rsqp_launch_spmv_ldsvec(p->variant_n, p->ncol, p->nslices, p->slice_r.p, p->rp.p, p->ci.p, p->use16 ? p->ci16.p : nullptr, p->rval.p, p->vin_c.p,
                                        p->vout_r.p, p->nbatch, p->nrow + 1, p->nnz, p->ncol, p->nrow, p->stream);
        else if (transposed)  // A'y on the CSC arrays (SpHbMat::transposed_times)
            e = rsqp_launch_spmv(p->blk_c.p, p->nblk_c, p->jc.p, p->ir.p, p->val.p, p->vin_r.p, p->vout_c.p, p->nbatch,
                                 p->ncol + 1, p->nnz, p->nrow, p->ncol, p->stream);
        else             // A x on the CSR copy (SpHbMat::times)
            e = rsqp_launch_spmv(p->blk_r.p, p->nblk_r, p->rp.p, p->ci.p, p->rval.p, p->vin_c.p, p->vout_r.p, p->nbatch,
                                 p->nrow + 1, p->nnz, p->ncol, p->nrow, p->stream);
        if (e != hipSuccess) return fail(RSQP_ERR_DEVICE, "spmv launch failed");
    }
    HIPCHK(hipEventRecord(p->ev1, p->stream));
    HIPCHK(hipEventSynchronize(p->ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, p->ev0, p->ev1));
    if (ms_per_launch) *ms_per_launch = ms / repeats;
    return RSQP_OK;
}

extern "C" int rsqp_spmv_plan_variant(const rsqp_spmv_plan *p, int transposed, int *idx16) {
    if (!p) return fail(RSQP_ERR_ARG, "null plan");
    if (idx16) *idx16 = p->use16 ? 1 : 0;
    return transposed ? p->variant_t : p->variant_n;
}

extern "C" int rsqp_spmv_plan_download(rsqp_spmv_plan *p, double *out, int transposed) {
    if (!p || !out) return fail(RSQP_ERR_ARG, "rsqp_spmv_plan_download");
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    if (transposed) HIPCHK(p->vout_c.download(out, (size_t)p->nbatch * p->ncol));
    else HIPCHK(p->vout_r.download(out, (size_t)p->nbatch * p->nrow));
    return RSQP_OK;
}
