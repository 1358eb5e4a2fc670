// qp_dump.cpp -- the on-disk QP formats of the reference (host-only part of librsqp_hip.so).
//
// Writers restate what QPSolverInterface::WriteQPDataToFile produces (called for every failed QP,
// reference src/QPhandler.cpp:569-571 <- src/Algorithm.cpp:69,1191):
//   layout RSQP_DUMP_QPOASES  src/qpOASESInterface.cpp:791-814: lb, lbA, ub, ubA, g, then A and H through
//                             SpHbMat::write_to_file(.., QPOASES) (src/SpHbMat.cpp:568-578): ir[nnz],
//                             jc[ncol+1], val[nnz]
//   layout RSQP_DUMP_QORE     src/QOREInterface.cpp:582-598: nV, nC, nnz(A), nnz(H), lb[nV+nC], ub[nV+nC],
//                             g, then A and H as CSR (src/SpHbMat.cpp:556-567): rowptr, col, val
// every number on its own line, doubles as "%23.16e" (src/Vector.cpp:212-215), ints as "%d".
// The reader is the QORE-layout reader of the reference's test driver (test/QPsolvers_testers.cpp:48-150),
// returning CSC like convert_csr_to_csc there (:18-29).
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/rsqp_hip.h"

namespace {

void put_d(FILE *f, const double *v, int n) {
    for (int i = 0; i < n; i++) std::fprintf(f, "%23.16e\n", v[i]);
}
void put_i(FILE *f, const int *v, int n) {
    for (int i = 0; i < n; i++) std::fprintf(f, "%d\n", v[i]);
}

// CSC (ncol + 1 pointers) -> CSR (nrow + 1 pointers): rows ascending, columns ascending within a row
void csc_to_csr(int nrow, int ncol, const int *jc, const int *ir, const double *val, std::vector<int> &rp,
                std::vector<int> &ci, std::vector<double> &rv) {
    const int nnz = jc[ncol];
    rp.assign(nrow + 1, 0); ci.resize(nnz); rv.resize(nnz);
    for (int k = 0; k < nnz; k++) rp[ir[k] + 1]++;
    for (int r = 0; r < nrow; r++) rp[r + 1] += rp[r];
    std::vector<int> fill(rp.begin(), rp.end() - 1);
    for (int c = 0; c < ncol; c++)
        for (int k = jc[c]; k < jc[c + 1]; k++) {
            const int p = fill[ir[k]]++;
            ci[p] = c;
            rv[p] = val[k];
        }
}

struct Tokens {
    std::vector<char> buf;
    char *cur = nullptr;
    bool load(const char *path) {
        FILE *f = std::fopen(path, "rb");
        if (!f) return false;
        std::fseek(f, 0, SEEK_END);
        const long n = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        buf.resize((size_t)n + 1);
        const size_t got = std::fread(buf.data(), 1, (size_t)n, f);
        std::fclose(f);
        buf[got] = 0;
        cur = buf.data();
        return true;
    }
    bool next_int(int &v) {
        char *end = nullptr;
        const long x = std::strtol(cur, &end, 10);
        if (end == cur) return false;
        cur = end; v = (int)x;
        return true;
    }
    bool next_double(double &v) {
        char *end = nullptr;
        const double x = std::strtod(cur, &end);
        if (end == cur) return false;
        cur = end; v = x;
        return true;
    }
    bool at_end() {
        while (*cur == ' ' || *cur == '\n' || *cur == '\r' || *cur == '\t') cur++;
        return *cur == 0;
    }
};

}  // namespace

extern "C" int rsqp_write_qp_dump(const char *path, int layout, int nV, int nC, const double *lb, const double *ub,
                                  const double *lbA, const double *ubA, const double *g, const int *A_jc,
                                  const int *A_ir, const double *A_val, const int *H_jc, const int *H_ir,
                                  const double *H_val) {
    if (!path || nV <= 0 || nC < 0 || !lb || !ub || !g || (nC > 0 && (!lbA || !ubA)) || !A_jc || !H_jc ||
        (layout != RSQP_DUMP_QPOASES && layout != RSQP_DUMP_QORE))
        return RSQP_ERR_ARG;
    const int annz = A_jc[nV], hnnz = H_jc[nV];
    if ((annz > 0 && (!A_ir || !A_val)) || (hnnz > 0 && (!H_ir || !H_val))) return RSQP_ERR_ARG;
    FILE *f = std::fopen(path, "w");
    if (!f) return RSQP_ERR_ARG;
    if (layout == RSQP_DUMP_QPOASES) {
        put_d(f, lb, nV); put_d(f, lbA, nC); put_d(f, ub, nV); put_d(f, ubA, nC); put_d(f, g, nV);
        put_i(f, A_ir, annz); put_i(f, A_jc, nV + 1); put_d(f, A_val, annz);
        put_i(f, H_ir, hnnz); put_i(f, H_jc, nV + 1); put_d(f, H_val, hnnz);
    } else {
        std::fprintf(f, "%d\n%d\n%d\n%d\n", nV, nC, annz, hnnz);
        put_d(f, lb, nV); put_d(f, lbA, nC);      // QORE keeps one lb / ub of length nV + nC
        put_d(f, ub, nV); put_d(f, ubA, nC);
        put_d(f, g, nV);
        std::vector<int> rp, ci;
        std::vector<double> rv;
        csc_to_csr(nC, nV, A_jc, A_ir, A_val, rp, ci, rv);
        put_i(f, rp.data(), nC + 1); put_i(f, ci.data(), annz); put_d(f, rv.data(), annz);
        csc_to_csr(nV, nV, H_jc, H_ir, H_val, rp, ci, rv);
        put_i(f, rp.data(), nV + 1); put_i(f, ci.data(), hnnz); put_d(f, rv.data(), hnnz);
    }
    return std::fclose(f) == 0 ? RSQP_OK : RSQP_ERR_ARG;
}

extern "C" int rsqp_write_qp_data(const rsqp_solver *s, const char *path, int layout) {
    if (!s || !path) return RSQP_ERR_ARG;
    const int nV = rsqp_get_nV(s), nC = rsqp_get_nC(s);
    std::vector<double> v[5];
    for (int k = 0; k < 5; k++) {
        v[k].resize(k <= RSQP_VEC_UB ? nV : nC);
        const int rc = rsqp_get_vector(s, k, v[k].data());
        if (rc != RSQP_OK) return rc;
    }
    // a matrix that was never set (LP handler: no H; nC = 0: no A) is written as an empty one
    std::vector<int> jc[2], ir[2];
    std::vector<double> val[2];
    for (int m = 0; m < 2; m++) {
        const int nnz = m == 0 ? rsqp_get_A_nnz(s) : rsqp_get_H_nnz(s);
        jc[m].assign(nV + 1, 0);
        if (nnz < 0) continue;
        ir[m].resize(nnz); val[m].resize(nnz);
        const int rc = m == 0 ? rsqp_get_A_csc(s, jc[m].data(), ir[m].data(), val[m].data(), nullptr)
                              : rsqp_get_H_csc(s, jc[m].data(), ir[m].data(), val[m].data(), nullptr);
        if (rc != RSQP_OK) return rc;
    }
    return rsqp_write_qp_dump(path, layout, nV, nC, v[RSQP_VEC_LB].data(), v[RSQP_VEC_UB].data(), v[RSQP_VEC_LBA].data(),
                              v[RSQP_VEC_UBA].data(), v[RSQP_VEC_G].data(), jc[0].data(), ir[0].data(), val[0].data(),
                              jc[1].data(), ir[1].data(), val[1].data());
}

extern "C" int rsqp_read_qore_dump_sizes(const char *path, int *nV, int *nC, int *nnzA, int *nnzH) {
    Tokens t;
    if (!path || !nV || !nC || !nnzA || !nnzH || !t.load(path)) return RSQP_ERR_ARG;
    if (!t.next_int(*nV) || !t.next_int(*nC) || !t.next_int(*nnzA) || !t.next_int(*nnzH)) return RSQP_ERR_ARG;
    return (*nV > 0 && *nC >= 0 && *nnzA >= 0 && *nnzH >= 0) ? RSQP_OK : RSQP_ERR_ARG;
}

extern "C" int rsqp_read_qore_dump(const char *path, double *lb, double *ub, double *lbA, double *ubA, double *g,
                                   int *A_jc, int *A_ir, double *A_val, int *H_jc, int *H_ir, double *H_val) {
    Tokens t;
    int nV, nC, annz, hnnz;
    if (!path || !lb || !ub || !g || !A_jc || !H_jc || !t.load(path)) return RSQP_ERR_ARG;
    if (!t.next_int(nV) || !t.next_int(nC) || !t.next_int(annz) || !t.next_int(hnnz)) return RSQP_ERR_ARG;
    if (nV <= 0 || nC < 0 || annz < 0 || hnnz < 0 || (nC > 0 && (!lbA || !ubA))) return RSQP_ERR_ARG;
    auto doubles = [&](double *dst, int n) { for (int i = 0; i < n; i++) if (!t.next_double(dst[i])) return false; return true; };
    auto ints = [&](int *dst, int n) { for (int i = 0; i < n; i++) if (!t.next_int(dst[i])) return false; return true; };
    if (!doubles(lb, nV) || !doubles(lbA, nC) || !doubles(ub, nV) || !doubles(ubA, nC) || !doubles(g, nV)) return RSQP_ERR_ARG;
    for (int m = 0; m < 2; m++) {
        const int nrow = m == 0 ? nC : nV, nnz = m == 0 ? annz : hnnz;
        std::vector<int> rp(nrow + 1), ci(nnz);
        std::vector<double> rv(nnz);
        if (!ints(rp.data(), nrow + 1) || !ints(ci.data(), nnz) || !doubles(rv.data(), nnz)) return RSQP_ERR_ARG;
        if (rp[0] != 0 || rp[nrow] != nnz) return RSQP_ERR_ARG;
        for (int r = 0; r < nrow; r++) if (rp[r] > rp[r + 1]) return RSQP_ERR_ARG;
        for (int k = 0; k < nnz; k++) if (ci[k] < 0 || ci[k] >= nV) return RSQP_ERR_ARG;
        int *jc = m == 0 ? A_jc : H_jc, *ir = m == 0 ? A_ir : H_ir;
        double *val = m == 0 ? A_val : H_val;
        if (nnz > 0 && (!ir || !val)) return RSQP_ERR_ARG;
        // the transpose of a CSR matrix read as CSC is its CSR form again: same routine, roles swapped
        std::vector<int> jcv, irv;
        std::vector<double> vv;
        csc_to_csr(nV, nrow, rp.data(), ci.data(), rv.data(), jcv, irv, vv);
        for (int c = 0; c <= nV; c++) jc[c] = jcv[c];
        for (int k = 0; k < nnz; k++) { ir[k] = irv[k]; val[k] = vv[k]; }
    }
    return t.at_end() ? RSQP_OK : RSQP_ERR_ARG;
}


// ---------------------------------------------------------------------------------------------------------------
// sharding of a batch over the ranks of a multi-GPU job (host-only; the same partitions as restartsqp_amd/parallel.py)
// ---------------------------------------------------------------------------------------------------------------
#include <algorithm>
#include <numeric>
#include <vector>
extern "C" int rsqp_shard_range(int nq, int rank, int world, int *lo, int *hi) {
    if (nq < 0 || world <= 0 || rank < 0 || rank >= world || !lo || !hi) return RSQP_ERR_ARG;
    const int base = nq / world, rem = nq % world;
    *lo = rank * base + std::min(rank, rem);
    *hi = *lo + base + (rank < rem ? 1 : 0);
    return RSQP_OK;
}
extern "C" int rsqp_balanced_shard(int nq, const int *nV, const int *nC, int rank, int world, int *idx, int *count) {
    if (nq < 0 || world <= 0 || rank < 0 || rank >= world || !nV || !nC || !idx || !count) return RSQP_ERR_ARG;
    std::vector<int> order(nq);
    std::iota(order.begin(), order.end(), 0);
    auto cost = [&](int k) { return (long long)nV[k] * std::max(nC[k], 1); };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost(a) > cost(b); });   // largest first, ties by index
    int n = 0;
    for (int j = 0; j < nq; j++) {
        const int rnd = j / world, pos = j % world;
        if ((rnd % 2 == 0 ? pos : world - 1 - pos) == rank) idx[n++] = order[j];
    }
    *count = n;
    return RSQP_OK;
}
