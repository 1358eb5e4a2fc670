/*
 * sphb_oracle.c -- CPU ORACLE (test infrastructure, not product code).
 * Restates the sparse containers and handler formulas of the reference:
 *   src/SpHbMat.cpp, src/SpTripletMat.cpp, src/Utils.cpp, src/QPhandler.cpp.
 * See rsqp_oracle.h for the parity status.
 */
#include "rsqp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int row, col; /* 1-based, as stored by SpTripletMat */
    double val;
    int orig; /* position in the (extended) triplet list */
} entry_t;

/* comparison rules of include/sqphot/SpHbMat.hpp:370-398. std::sort is not stable;
 * duplicates (never produced by the reference's callers) are ordered by `orig` here. */
static int cmp_col_major(const void *a, const void *b) {
    const entry_t *l = (const entry_t *)a, *r = (const entry_t *)b;
    if (l->col != r->col) return l->col < r->col ? -1 : 1;
    if (l->row != r->row) return l->row < r->row ? -1 : 1;
    return l->orig < r->orig ? -1 : (l->orig > r->orig);
}
static int cmp_row_major(const void *a, const void *b) {
    const entry_t *l = (const entry_t *)a, *r = (const entry_t *)b;
    if (l->row != r->row) return l->row < r->row ? -1 : 1;
    if (l->col != r->col) return l->col < r->col ? -1 : 1;
    return l->orig < r->orig ? -1 : (l->orig > r->orig);
}

/* shared tail of both setStructure overloads: sort, then emit index / value / order
 * arrays and the compressed pointer (SpHbMat.cpp:238-265 and :324-352). The reference
 * builds the pointer by incrementing every later slot per entry (O(nnz*ncol)); the
 * result is the plain prefix count computed here. */
static void emit_compressed(entry_t *e, int nnz, int nrow, int ncol, int compressed_row,
                            int *ptr, int *idx, double *out_val, int *order) {
    int nmajor = compressed_row ? nrow : ncol;
    qsort(e, (size_t)nnz, sizeof(entry_t), compressed_row ? cmp_row_major : cmp_col_major);
    for (int k = 0; k <= nmajor; k++) ptr[k] = 0;
    for (int i = 0; i < nnz; i++) {
        int major = compressed_row ? e[i].row : e[i].col; /* 1-based */
        idx[i] = (compressed_row ? e[i].col : e[i].row) - 1;
        out_val[i] = e[i].val;
        order[e[i].orig] = i;
        ptr[major]++; /* entries of 0-based major index (major-1) counted in slot major */
    }
    for (int k = 0; k < nmajor; k++) ptr[k + 1] += ptr[k];
}

int orc_sphb_set_structure(int nrow, int ncol, int nnz_t, const int *irow, const int *jcol,
                           const double *val, int n_ident, const int *id_irow,
                           const int *id_jcol, const int *id_size, const double *id_value,
                           int compressed_row, int *ptr, int *idx, double *out_val,
                           int *order) {
    int total = nnz_t;
    for (int b = 0; b < n_ident; b++) total += id_size[b];
    entry_t *e = (entry_t *)malloc(sizeof(entry_t) * (size_t)(total > 0 ? total : 1));
    int counter = 0;
    for (int i = 0; i < nnz_t; i++, counter++) {
        e[counter].row = irow[i];
        e[counter].col = jcol[i];
        e[counter].val = val[i];
        e[counter].orig = counter;
    }
    /* identity blocks appended after the triplet entries (SpHbMat.cpp:215-225) */
    for (int b = 0; b < n_ident; b++)
        for (int j = 0; j < id_size[b]; j++, counter++) {
            e[counter].row = id_irow[b] + j;
            e[counter].col = id_jcol[b] + j;
            e[counter].val = id_value[b];
            e[counter].orig = counter;
        }
    emit_compressed(e, total, nrow, ncol, compressed_row, ptr, idx, out_val, order);
    free(e);
    return total;
}

int orc_sphb_sym_nnz(int nnz_t, const int *irow, const int *jcol, int is_symmetric) {
    int n = 0;
    for (int i = 0; i < nnz_t; i++) n += (is_symmetric && irow[i] != jcol[i]) ? 2 : 1;
    return n;
}

int orc_sphb_set_structure_sym(int nrow, int ncol, int nnz_t, const int *irow, const int *jcol,
                               const double *val, int is_symmetric, int compressed_row,
                               int *ptr, int *idx, double *out_val, int *order) {
    int total = orc_sphb_sym_nnz(nnz_t, irow, jcol, is_symmetric);
    entry_t *e = (entry_t *)malloc(sizeof(entry_t) * (size_t)(total > 0 ? total : 1));
    int counter = 0;
    /* each off-diagonal is followed immediately by its mirror (SpHbMat.cpp:296-309) */
    for (int i = 0; i < nnz_t; i++) {
        e[counter].row = irow[i];
        e[counter].col = jcol[i];
        e[counter].val = val[i];
        e[counter].orig = counter;
        counter++;
        if (is_symmetric && irow[i] != jcol[i]) {
            e[counter].row = jcol[i];
            e[counter].col = irow[i];
            e[counter].val = val[i];
            e[counter].orig = counter;
            counter++;
        }
    }
    emit_compressed(e, total, nrow, ncol, compressed_row, ptr, idx, out_val, order);
    free(e);
    return total;
}

void orc_sphb_set_matval(int nnz_total, int n_ident_entries, const int *order,
                         const double *triplet_val, double *matval) {
    /* identity entries keep the value written by setStructure (SpHbMat.cpp:377-379) */
    for (int i = 0; i < nnz_total - n_ident_entries; i++) matval[order[i]] = triplet_val[i];
}

void orc_sphb_set_matval_sym(int nnz_t, const int *irow, const int *jcol, int is_symmetric,
                             const int *order, const double *triplet_val, double *matval) {
    int j = 0;
    for (int i = 0; i < nnz_t; i++) {
        matval[order[j++]] = triplet_val[i];
        if (is_symmetric && irow[i] != jcol[i]) matval[order[j++]] = triplet_val[i];
    }
}

void orc_sphb_times(int nrow, int ncol, int compressed_row, const int *ptr, const int *idx,
                    const double *val, const double *p, double *result) {
    for (int i = 0; i < nrow; i++) result[i] = 0.0;
    if (compressed_row) {
        for (int r = 0; r < nrow; r++)
            for (int k = ptr[r]; k < ptr[r + 1]; k++) result[r] += val[k] * p[idx[k]];
    } else {
        /* entry order == column by column (SpHbMat.cpp:729-735) */
        for (int c = 0; c < ncol; c++)
            for (int k = ptr[c]; k < ptr[c + 1]; k++) result[idx[k]] += val[k] * p[c];
    }
}

void orc_sphb_transposed_times(int nrow, int ncol, int compressed_row, const int *ptr,
                               const int *idx, const double *val, const double *p,
                               double *result) {
    for (int i = 0; i < ncol; i++) result[i] = 0.0;
    if (compressed_row) {
        for (int r = 0; r < nrow; r++)
            for (int k = ptr[r]; k < ptr[r + 1]; k++) result[idx[k]] += val[k] * p[r];
    } else {
        for (int c = 0; c < ncol; c++)
            for (int k = ptr[c]; k < ptr[c + 1]; k++) result[c] += val[k] * p[idx[k]];
    }
}

int orc_sphb_from_dense(const double *data, int nrow, int ncol, int row_oriented,
                        int compressed_row, int *ptr, int *idx, double *val) {
    const double m_eps = 1.0e-16; /* include/sqphot/Utils.hpp:36 */
    int n = 0;
    if (compressed_row) {
        ptr[0] = 0;
        for (int i = 0; i < nrow; i++) {
            for (int j = 0; j < ncol; j++) {
                double v = row_oriented ? data[i * ncol + j] : data[i + j * nrow];
                if (fabs(v) > m_eps) {
                    val[n] = v;
                    idx[n] = j;
                    n++;
                }
            }
            ptr[i + 1] = n;
        }
    } else {
        ptr[0] = 0;
        for (int j = 0; j < ncol; j++) {
            for (int i = 0; i < nrow; i++) {
                double v = row_oriented ? data[j + i * ncol] : data[j * nrow + i];
                if (fabs(v) > m_eps) {
                    val[n] = v;
                    idx[n] = i;
                    n++;
                }
            }
            ptr[j + 1] = n;
        }
    }
    return n;
}

void orc_sphb_to_dense(int nrow, int ncol, int compressed_row, const int *ptr, const int *idx,
                       const double *val, double *dense_row_major) {
    int nmajor = compressed_row ? nrow : ncol;
    for (int m = 0; m < nmajor; m++)
        for (int k = ptr[m]; k < ptr[m + 1]; k++) {
            int r = compressed_row ? m : idx[k];
            int c = compressed_row ? idx[k] : m;
            dense_row_major[ncol * r + c] = val[k];
        }
}

void orc_triplet_times(int nrow, int ncol, int nnz, const int *irow, const int *jcol,
                       const double *val, int is_symmetric, const double *p, double *result) {
    (void)ncol;
    for (int i = 0; i < nrow; i++) result[i] = 0.0;
    for (int i = 0; i < nnz; i++) {
        result[irow[i] - 1] += val[i] * p[jcol[i] - 1];
        if (is_symmetric && irow[i] != jcol[i]) result[jcol[i] - 1] += val[i] * p[irow[i] - 1];
    }
}

void orc_triplet_transposed_times(int nrow, int ncol, int nnz, const int *irow,
                                  const int *jcol, const double *val, int is_symmetric,
                                  const double *p, double *result) {
    if (is_symmetric) {
        orc_triplet_times(nrow, ncol, nnz, irow, jcol, val, 1, p, result);
        return;
    }
    for (int i = 0; i < ncol; i++) result[i] = 0.0;
    for (int i = 0; i < nnz; i++) result[jcol[i] - 1] += val[i] * p[irow[i] - 1];
}

double orc_one_norm(const double *x, int n) {
    double s = 0.0;
    for (int i = 0; i < n; i++) s += fabs(x[i]);
    return s;
}

double orc_inf_norm(const double *x, int n) {
    double m = 0.0;
    for (int i = 0; i < n; i++)
        if (fabs(x[i]) > m) m = fabs(x[i]);
    return m;
}

/* ------------------------------------------------------------------ */
/* QPhandler formulas                                                  */
/* ------------------------------------------------------------------ */
#define ORC_HANDLER_INF 1.0e18 /* include/sqphot/Utils.hpp:35 */

void orc_handler_set_bounds(int n, int m, double delta, const double *x_l, const double *x_u,
                            const double *x_k, const double *c_l, const double *c_u,
                            const double *c_k, double *lb, double *ub, double *lbA,
                            double *ubA) {
    for (int i = 0; i < m; i++) {
        lbA[i] = c_l[i] - c_k[i];
        ubA[i] = c_u[i] - c_k[i];
    }
    for (int i = 0; i < n; i++) {
        lb[i] = fmax(x_l[i] - x_k[i], -delta);
        ub[i] = fmin(x_u[i] - x_k[i], delta);
    }
    for (int i = 0; i < 2 * m; i++) ub[n + i] = ORC_HANDLER_INF;
}

void orc_handler_update_bounds(int n, int m, double delta, const double *x_l,
                               const double *x_u, const double *x_k, const double *c_l,
                               const double *c_k, double *lb, double *ub, double *lbA) {
    for (int i = 0; i < m; i++) lbA[i] = c_l[i] - c_k[i];
    for (int i = 0; i < n; i++) {
        lb[i] = fmax(x_l[i] - x_k[i], -delta);
        ub[i] = fmin(x_u[i] - x_k[i], delta);
    }
}

void orc_handler_set_g(int n, int m, const double *grad, double rho, double *g) {
    for (int i = 0; i < n + 2 * m; i++) g[i] = i < n ? grad[i] : rho;
}
