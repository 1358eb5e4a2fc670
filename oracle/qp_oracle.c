/*
 * qp_oracle.c -- CPU ORACLE (test infrastructure, not product code).
 *
 * Stand-in for qpOASES 3.2.1 `SQProblem` (init / hotstart), the third-party solver the
 * reference calls at src/qpOASESInterface.cpp:155,180,184,191,197,204 and which is NOT
 * part of /root/reference (fetched by cmake/ExternalQPOASES.cmake at configure time).
 * The algorithm is restated from its published description -- the online active-set
 * strategy (Ferreau, Bock, Diehl 2008; Ferreau et al. 2014) in dense null-space form:
 *
 *   - homotopy from an auxiliary QP, whose solution is known, to the requested QP;
 *   - cold start: x = 0, y = 0, every finite bound in the working set at its lower
 *     side (qpOASES `initialStatusBounds = ST_LOWER`), no constraint active;
 *     inactive sides relaxed by `boundRelaxation = 1e4`;
 *   - TQ factorisation A_AC,FR * Q = [0 T] (T reverse triangular), Cholesky
 *     R'R = Z'HZ, both kept current with Givens rotations;
 *   - primal + dual ratio tests with the qpOASES tolerance epsDen = 1e3*EPS; ties resolved by the lowest candidate id in the order
 *     [active constraints][fixed bounds][inactive constr. lower][inactive constr. upper]
 *     [free var lower][free var upper];
 *   - linear-dependence handling by exchange (ensureLI), infeasibility when no
 *     exchange partner exists;
 *   - "flipping bounds" when a removal would leave Z'HZ not positive definite
 *     (qpOASES enableFlippingBounds, on in setToReliable()), unbounded if the
 *     opposite side is infinite.
 * Not restated (qpOASES internals that only alter the homotopy path): ramping, far
 * bounds, iterative refinement, periodic Cholesky refactorisation.
 *
 * PARITY: the reference commits no expected QP solution, and qpOASES cannot be built
 * here => "parity unpinned" against qpOASES. What IS checked: the reference's own KKT
 * certificate (kkt_oracle.c) on every answer, plus an independent brute-force
 * enumeration on tiny convex QPs (tests/test_oracle_qp.py).
 *
 * Storage: Q[c*nV + v] (column c, variable v); T[i*nV + c] (active-constraint position
 * i, Q column c; non-zero for c >= nFR-1-i); R[c*nV + r] upper triangular, r <= c < nZ.
 */
#include "rsqp_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define EPS_DEN (1.0e3 * ORC_EPS)
#define BOUND_RELAXATION 1.0e4
#define BOUND_TOLERANCE (1.0e6 * ORC_EPS)
#define EPS_LI 1.0e-9
#define EPS_PD_REL 1.0e-10
#define EPS_PD_ABS 1.0e-25

struct orc_qp {
    int nV, nC;
    int *Ajc, *Air; double *Aval; /* CSC */
    int *Arp, *Aci; double *Arv;  /* CSR copy of A */
    int *Hjc, *Hir; double *Hval; /* CSC, full symmetric; NULL => H = 0 */
    int haveA, haveH;
    /* requested data (targets of the homotopy), clamped to +-INFTY */
    double *gN, *lbN, *ubN, *lbAN, *ubAN;
    /* current homotopy data */
    double *g, *lb, *ub, *lbA, *ubA;
    double *x, *y, *Ax;
    int *Sb, *Sc; /* -1 lower, 0 inactive/free, +1 upper */
    int nFR, nAC;
    int *AC;
    double *Q, *T, *R;
    /* workspaces */
    double *dg, *dlb, *dub, *dlbA, *dubA, *dx, *dy, *dAx;
    double *w, *wv1, *wv2, *wv3, *wc1, *wc2, *wq;
    int status, infeasible, unbounded, nflips;
    int sizeT;
    double hreg; /* H + hreg*I: qpOASES treats an all-zero Hessian (LP) as regVal*I */
    int guess_c_from_y0; /* 0: the reference's rule for init(.., xOpt, yOpt, guessedBounds) -- constraints clipped
                            at A x0; 1: the HIP engine's rule -- constraint sides from the signs of y0 (see setup_aux) */
};

static double clampinf(double v) {
    if (v != v) return v;
    if (v > ORC_INFTY) return ORC_INFTY;
    if (v < -ORC_INFTY) return -ORC_INFTY;
    return v;
}

static void *xcalloc(size_t n, size_t sz) {
    void *p = calloc(n > 0 ? n : 1, sz);
    if (!p) {
        fprintf(stderr, "oracle: out of memory\n");
        abort();
    }
    return p;
}

orc_qp *orc_qp_create(int nV, int nC) {
    orc_qp *qp = (orc_qp *)xcalloc(1, sizeof(orc_qp));
    qp->nV = nV;
    qp->nC = nC;
    qp->sizeT = nV < nC ? nV : nC;
#define DV(name, n) qp->name = (double *)xcalloc((size_t)(n), sizeof(double))
    DV(gN, nV); DV(lbN, nV); DV(ubN, nV); DV(lbAN, nC); DV(ubAN, nC);
    DV(g, nV); DV(lb, nV); DV(ub, nV); DV(lbA, nC); DV(ubA, nC);
    DV(x, nV); DV(y, nV + nC); DV(Ax, nC);
    DV(dg, nV); DV(dlb, nV); DV(dub, nV); DV(dlbA, nC); DV(dubA, nC);
    DV(dx, nV); DV(dy, nV + nC); DV(dAx, nC);
    DV(w, nV); DV(wv1, nV); DV(wv2, nV); DV(wv3, nV); DV(wc1, nC); DV(wc2, nC); DV(wq, nV);
    DV(Q, (size_t)nV * nV);
    DV(R, (size_t)nV * nV);
    DV(T, (size_t)qp->sizeT * nV);
#undef DV
    qp->Sb = (int *)xcalloc((size_t)nV, sizeof(int));
    qp->Sc = (int *)xcalloc((size_t)nC, sizeof(int));
    qp->AC = (int *)xcalloc((size_t)nC, sizeof(int));
    qp->status = ORC_QPS_NOTINITIALISED;
    return qp;
}

void orc_qp_destroy(orc_qp *qp) {
    if (!qp) return;
    free(qp->Ajc); free(qp->Air); free(qp->Aval); free(qp->Arp); free(qp->Aci); free(qp->Arv);
    free(qp->Hjc); free(qp->Hir); free(qp->Hval);
    free(qp->gN); free(qp->lbN); free(qp->ubN); free(qp->lbAN); free(qp->ubAN);
    free(qp->g); free(qp->lb); free(qp->ub); free(qp->lbA); free(qp->ubA);
    free(qp->x); free(qp->y); free(qp->Ax);
    free(qp->dg); free(qp->dlb); free(qp->dub); free(qp->dlbA); free(qp->dubA);
    free(qp->dx); free(qp->dy); free(qp->dAx);
    free(qp->w); free(qp->wv1); free(qp->wv2); free(qp->wv3); free(qp->wc1); free(qp->wc2);
    free(qp->wq);
    free(qp->Q); free(qp->R); free(qp->T);
    free(qp->Sb); free(qp->Sc); free(qp->AC);
    free(qp);
}

int orc_qp_set_A_csc(orc_qp *qp, const int *jc, const int *ir, const double *val) {
    int nV = qp->nV, nC = qp->nC, nnz = jc[nV];
    free(qp->Ajc); free(qp->Air); free(qp->Aval); free(qp->Arp); free(qp->Aci); free(qp->Arv);
    qp->Ajc = (int *)xcalloc((size_t)nV + 1, sizeof(int));
    qp->Air = (int *)xcalloc((size_t)nnz, sizeof(int));
    qp->Aval = (double *)xcalloc((size_t)nnz, sizeof(double));
    memcpy(qp->Ajc, jc, sizeof(int) * ((size_t)nV + 1));
    memcpy(qp->Air, ir, sizeof(int) * (size_t)nnz);
    memcpy(qp->Aval, val, sizeof(double) * (size_t)nnz);
    /* CSR copy (columns ascending within each row) */
    qp->Arp = (int *)xcalloc((size_t)nC + 1, sizeof(int));
    qp->Aci = (int *)xcalloc((size_t)nnz, sizeof(int));
    qp->Arv = (double *)xcalloc((size_t)nnz, sizeof(double));
    for (int k = 0; k < nnz; k++) qp->Arp[ir[k] + 1]++;
    for (int r = 0; r < nC; r++) qp->Arp[r + 1] += qp->Arp[r];
    int *fill = (int *)xcalloc((size_t)nC, sizeof(int));
    for (int c = 0; c < nV; c++)
        for (int k = jc[c]; k < jc[c + 1]; k++) {
            int r = ir[k], p = qp->Arp[r] + fill[r]++;
            qp->Aci[p] = c;
            qp->Arv[p] = val[k];
        }
    free(fill);
    qp->haveA = 1;
    return 0;
}

int orc_qp_set_H_csc(orc_qp *qp, const int *jc, const int *ir, const double *val) {
    int nV = qp->nV;
    free(qp->Hjc); free(qp->Hir); free(qp->Hval);
    qp->Hjc = NULL; qp->Hir = NULL; qp->Hval = NULL;
    qp->haveH = 0;
    if (!jc) return 0;
    int nnz = jc[nV];
    qp->Hjc = (int *)xcalloc((size_t)nV + 1, sizeof(int));
    qp->Hir = (int *)xcalloc((size_t)nnz, sizeof(int));
    qp->Hval = (double *)xcalloc((size_t)nnz, sizeof(double));
    memcpy(qp->Hjc, jc, sizeof(int) * ((size_t)nV + 1));
    memcpy(qp->Hir, ir, sizeof(int) * (size_t)nnz);
    memcpy(qp->Hval, val, sizeof(double) * (size_t)nnz);
    qp->haveH = 1;
    return 0;
}

/* ---------------- small linear-algebra helpers ---------------- */
static void A_times(const orc_qp *qp, const double *v, double *out) {
    for (int i = 0; i < qp->nC; i++) out[i] = 0.0;
    if (!qp->haveA) return;
    for (int r = 0; r < qp->nC; r++) {
        double s = 0.0;
        for (int k = qp->Arp[r]; k < qp->Arp[r + 1]; k++) s += qp->Arv[k] * v[qp->Aci[k]];
        out[r] = s;
    }
}
static void AT_times(const orc_qp *qp, const double *yc, double *out) {
    for (int c = 0; c < qp->nV; c++) {
        double s = 0.0;
        if (qp->haveA)
            for (int k = qp->Ajc[c]; k < qp->Ajc[c + 1]; k++) s += qp->Aval[k] * yc[qp->Air[k]];
        out[c] = s;
    }
}
static void H_times(const orc_qp *qp, const double *v, double *out) {
    for (int c = 0; c < qp->nV; c++) {
        double s = 0.0;
        if (qp->haveH) /* symmetric: column c of H == row c */
            for (int k = qp->Hjc[c]; k < qp->Hjc[c + 1]; k++) s += qp->Hval[k] * v[qp->Hir[k]];
        out[c] = s + qp->hreg * v[c];
    }
}
static double dotn(const double *a, const double *b, int n) {
    double s = 0.0;
    for (int i = 0; i < n; i++) s += a[i] * b[i];
    return s;
}

/* rotation of the pair (first, second) that maps first -> 0, second -> r */
static void givens(double a_elim, double b_keep, double *c, double *s) {
    if (a_elim == 0.0) {
        *c = 1.0;
        *s = 0.0;
        return;
    }
    double r = hypot(a_elim, b_keep);
    *c = b_keep / r;
    *s = a_elim / r;
}
static void rot_pair(double *p, double *q, double c, double s) {
    double a = *p, b = *q;
    *p = c * a - s * b;
    *q = s * a + c * b;
}
static void rot_Q_cols(orc_qp *qp, int j, double c, double s) {
    int nV = qp->nV;
    double *cj = qp->Q + (size_t)j * nV, *ck = cj + nV;
    for (int v = 0; v < nV; v++) rot_pair(cj + v, ck + v, c, s);
}
static void rot_T_cols(orc_qp *qp, int j, double c, double s) {
    int nV = qp->nV;
    for (int i = 0; i < qp->nAC; i++) rot_pair(qp->T + (size_t)i * nV + j, qp->T + (size_t)i * nV + j + 1, c, s);
}
/* column rotation (j,j+1) of R followed by the row rotation that restores triangularity */
static void rot_R(orc_qp *qp, int j, int nZ, double c, double s) {
    int nV = qp->nV;
    double *cj = qp->R + (size_t)j * nV, *ck = cj + nV;
    for (int r = 0; r <= j + 1 && r < nZ; r++) rot_pair(cj + r, ck + r, c, s);
    /* kill the sub-diagonal R[j+1][j] with a row rotation on rows (j, j+1) */
    double sub = cj[j + 1], diag = cj[j];
    if (sub != 0.0) {
        double r = hypot(diag, sub), cc = diag / r, ss = sub / r;
        for (int col = j; col < nZ; col++) {
            double *pc = qp->R + (size_t)col * nV;
            double a = pc[j], b = pc[j + 1];
            pc[j] = cc * a + ss * b;
            pc[j + 1] = -ss * a + cc * b;
        }
        cj[j + 1] = 0.0;
    }
}

/* ---------------- working-set updates ---------------- */
static void free_row_of_A(const orc_qp *qp, int i, double *a) {
    for (int v = 0; v < qp->nV; v++) a[v] = 0.0;
    for (int k = qp->Arp[i]; k < qp->Arp[i + 1]; k++)
        if (qp->Sb[qp->Aci[k]] == 0) a[qp->Aci[k]] = qp->Arv[k];
}

/* w = Q' a (first nFR columns) */
static void QT_times(const orc_qp *qp, const double *a, double *w) {
    int nV = qp->nV;
    for (int c = 0; c < qp->nFR; c++) w[c] = dotn(qp->Q + (size_t)c * nV, a, nV);
}

/* returns 1 if constraint i is linearly independent of the working set */
static int constraint_is_LI(orc_qp *qp, int i) {
    int nZ = qp->nFR - qp->nAC;
    if (nZ <= 0) return 0;
    free_row_of_A(qp, i, qp->wv1);
    double na = sqrt(dotn(qp->wv1, qp->wv1, qp->nV));
    if (na == 0.0) return 0;
    double s = 0.0;
    for (int c = 0; c < nZ; c++) {
        double d = dotn(qp->Q + (size_t)c * qp->nV, qp->wv1, qp->nV);
        s += d * d;
    }
    return sqrt(s) > EPS_LI * na;
}
static int bound_is_LI(orc_qp *qp, int v) {
    int nZ = qp->nFR - qp->nAC;
    if (nZ <= 0) return 0;
    double s = 0.0;
    for (int c = 0; c < nZ; c++) {
        double d = qp->Q[(size_t)c * qp->nV + v];
        s += d * d;
    }
    return sqrt(s) > EPS_LI;
}

/* skipZ: the incoming row is (numerically) orthogonal to all but the LAST null-space
 * column -- the exchange case -- so the null-space sweep is the identity and is skipped */
static void add_constraint(orc_qp *qp, int i, int status, int upd_chol, int skipZ) {
    int nV = qp->nV, nFR = qp->nFR, nZ = nFR - qp->nAC;
    double *a = qp->wv1, *w = qp->wq;
    free_row_of_A(qp, i, a);
    QT_times(qp, a, w);
    for (int j = 0; !skipZ && j + 1 < nZ; j++) {
        double c, s;
        givens(w[j], w[j + 1], &c, &s);
        if (s == 0.0) continue;
        rot_pair(w + j, w + j + 1, c, s);
        rot_Q_cols(qp, j, c, s);
        if (upd_chol) rot_R(qp, j, nZ, c, s);
    }
    double *row = qp->T + (size_t)qp->nAC * nV;
    for (int c = 0; c < nV; c++) row[c] = (c >= nZ - 1 && c < nFR) ? w[c] : 0.0;
    qp->AC[qp->nAC++] = i;
    qp->Sc[i] = status;
}

static void add_bound(orc_qp *qp, int v, int status, int upd_chol, int skipZ) {
    int nV = qp->nV, nFR = qp->nFR, nZ = nFR - qp->nAC;
    double *q = qp->wq;
    for (int c = 0; c < nFR; c++) q[c] = qp->Q[(size_t)c * nV + v];
    for (int j = 0; !skipZ && j + 1 < nZ; j++) {
        double c, s;
        givens(q[j], q[j + 1], &c, &s);
        if (s == 0.0) continue;
        rot_pair(q + j, q + j + 1, c, s);
        rot_Q_cols(qp, j, c, s);
        if (upd_chol) rot_R(qp, j, nZ, c, s);
    }
    for (int j = (nZ > 0 ? nZ - 1 : 0); j + 1 < nFR; j++) {
        double c, s;
        givens(q[j], q[j + 1], &c, &s);
        if (s == 0.0) continue;
        rot_pair(q + j, q + j + 1, c, s);
        rot_Q_cols(qp, j, c, s);
        rot_T_cols(qp, j, c, s);
    }
    /* row v is now +-e_{nFR-1}: drop that row and column */
    for (int c = 0; c < nFR; c++) qp->Q[(size_t)c * nV + v] = 0.0;
    for (int u = 0; u < nV; u++) qp->Q[(size_t)(nFR - 1) * nV + u] = 0.0;
    for (int i = 0; i < qp->nAC; i++) qp->T[(size_t)i * nV + nFR - 1] = 0.0;
    qp->Sb[v] = status;
    qp->nFR = nFR - 1;
}

/* append the Cholesky column for the new null-space column zc (R is zc x zc). 0 = ok */
static int chol_append(orc_qp *qp, int zc) {
    int nV = qp->nV;
    double *z = qp->Q + (size_t)zc * nV, *Hz = qp->wv2, *r = qp->wv3;
    H_times(qp, z, Hz);
    double zHz = dotn(z, Hz, nV), rr = 0.0;
    for (int j = 0; j < zc; j++) {
        double s = dotn(qp->Q + (size_t)j * nV, Hz, nV);
        for (int k = 0; k < j; k++) s -= qp->R[(size_t)j * nV + k] * r[k];
        r[j] = s / qp->R[(size_t)j * nV + j];
        rr += r[j] * r[j];
    }
    double rho2 = zHz - rr;
    if (!(rho2 > EPS_PD_REL * (fabs(zHz) + rr) + EPS_PD_ABS)) return 1;
    for (int j = 0; j < zc; j++) qp->R[(size_t)zc * nV + j] = r[j];
    qp->R[(size_t)zc * nV + zc] = sqrt(rho2);
    for (int j = zc + 1; j < nV; j++) qp->R[(size_t)zc * nV + j] = 0.0;
    return 0;
}

/* TQ part of removing the active constraint at logical position k; returns new Z column */
static int remove_constraint_tq(orc_qp *qp, int k) {
    int nV = qp->nV, nFR = qp->nFR, cons = qp->AC[k];
    for (int i = k; i + 1 < qp->nAC; i++) {
        memcpy(qp->T + (size_t)i * nV, qp->T + (size_t)(i + 1) * nV, sizeof(double) * (size_t)nV);
        qp->AC[i] = qp->AC[i + 1];
    }
    qp->nAC--;
    memset(qp->T + (size_t)qp->nAC * nV, 0, sizeof(double) * (size_t)nV);
    for (int i = k; i < qp->nAC; i++) {
        int c0 = nFR - 2 - i;
        double c, s;
        givens(qp->T[(size_t)i * nV + c0], qp->T[(size_t)i * nV + c0 + 1], &c, &s);
        if (s == 0.0) continue;
        rot_T_cols(qp, c0, c, s);
        qp->T[(size_t)i * nV + c0] = 0.0;
        rot_Q_cols(qp, c0, c, s);
    }
    qp->Sc[cons] = 0;
    return nFR - qp->nAC - 1;
}

static int remove_bound_tq(orc_qp *qp, int v) {
    int nV = qp->nV, cn = qp->nFR;
    qp->nFR++;
    qp->Sb[v] = 0;
    for (int u = 0; u < nV; u++) qp->Q[(size_t)cn * nV + u] = 0.0;
    for (int c = 0; c <= cn; c++) qp->Q[(size_t)c * nV + v] = 0.0;
    qp->Q[(size_t)cn * nV + v] = 1.0;
    /* new T column: A[AC[i]][v] */
    for (int i = 0; i < qp->nAC; i++) {
        double av = 0.0;
        int r = qp->AC[i];
        for (int k = qp->Arp[r]; k < qp->Arp[r + 1]; k++)
            if (qp->Aci[k] == v) av = qp->Arv[k];
        qp->T[(size_t)i * nV + cn] = av;
    }
    for (int i = 0; i < qp->nAC; i++) {
        int c0 = cn - 1 - i;
        double c, s;
        givens(qp->T[(size_t)i * nV + c0], qp->T[(size_t)i * nV + c0 + 1], &c, &s);
        if (s == 0.0) continue;
        rot_T_cols(qp, c0, c, s);
        qp->T[(size_t)i * nV + c0] = 0.0;
        rot_Q_cols(qp, c0, c, s);
    }
    return qp->nFR - qp->nAC - 1;
}

static int position_in_AC(const orc_qp *qp, int cons) {
    for (int i = 0; i < qp->nAC; i++)
        if (qp->AC[i] == cons) return i;
    return -1;
}

/* full Cholesky of Z'HZ. 0 = ok, 1 = not positive definite */
static int chol_setup(orc_qp *qp) {
    int nZ = qp->nFR - qp->nAC;
    for (int c = 0; c < nZ; c++)
        if (chol_append(qp, c)) return 1;
    return 0;
}

/* ---------------- auxiliary QP ---------------- */
static void store_targets(orc_qp *qp, const double *g, const double *lb, const double *ub,
                          const double *lbA, const double *ubA) {
    for (int i = 0; i < qp->nV; i++) {
        qp->gN[i] = g[i];
        qp->lbN[i] = lb ? clampinf(lb[i]) : -ORC_INFTY;
        qp->ubN[i] = ub ? clampinf(ub[i]) : ORC_INFTY;
    }
    for (int i = 0; i < qp->nC; i++) {
        qp->lbAN[i] = lbA ? clampinf(lbA[i]) : -ORC_INFTY;
        qp->ubAN[i] = ubA ? clampinf(ubA[i]) : ORC_INFTY;
    }
}

/* guess_c may be NULL; mode: 0 cold, 1 from x0 (and optional y0 / guess_b), 2 explicit */
static int setup_aux(orc_qp *qp, const double *x0, const double *y0, const int *guess_b,
                     const int *guess_c) {
    int nV = qp->nV, nC = qp->nC;
    qp->status = ORC_QPS_PREPARINGAUXILIARYQP;
    qp->infeasible = qp->unbounded = 0;
    for (int v = 0; v < nV; v++) qp->x[v] = x0 ? x0[v] : 0.0;
    for (int i = 0; i < nV + nC; i++) qp->y[i] = y0 ? y0[i] : 0.0;

    /* working-set guess for the bounds */
    for (int v = 0; v < nV; v++) {
        int s;
        if (guess_b) s = guess_b[v];
        else if (x0) {
            if (qp->x[v] <= qp->lbN[v] + BOUND_TOLERANCE) s = -1;
            else if (qp->x[v] >= qp->ubN[v] - BOUND_TOLERANCE) s = 1;
            else s = 0;
        } else if (y0) s = y0[v] > ORC_EPS ? -1 : (y0[v] < -ORC_EPS ? 1 : 0);
        else s = -1; /* initialStatusBounds = ST_LOWER */
        if (s == -1 && qp->lbN[v] <= -ORC_INFTY) s = qp->ubN[v] < ORC_INFTY && !x0 && !guess_b ? 1 : 0;
        if (s == 1 && qp->ubN[v] >= ORC_INFTY) s = 0;
        qp->Sb[v] = s;
    }
    /* TQ for the bounds alone: Q = identity on the free variables */
    memset(qp->Q, 0, sizeof(double) * (size_t)nV * nV);
    memset(qp->T, 0, sizeof(double) * (size_t)qp->sizeT * nV);
    memset(qp->R, 0, sizeof(double) * (size_t)nV * nV);
    qp->nFR = 0;
    for (int v = 0; v < nV; v++)
        if (qp->Sb[v] == 0) qp->Q[(size_t)(qp->nFR++) * nV + v] = 1.0;
    qp->nAC = 0;
    for (int i = 0; i < nC; i++) qp->Sc[i] = 0;
    A_times(qp, qp->x, qp->Ax);

    /* constraints */
    for (int i = 0; i < nC; i++) {
        int s = 0;
        if (guess_c) s = guess_c[i];
        else if (y0 && (!x0 || qp->guess_c_from_y0)) s = y0[nV + i] > ORC_EPS ? -1 : (y0[nV + i] < -ORC_EPS ? 1 : 0);
        else if (x0) {
            if (qp->Ax[i] <= qp->lbAN[i] + BOUND_TOLERANCE) s = -1;
            else if (qp->Ax[i] >= qp->ubAN[i] - BOUND_TOLERANCE) s = 1;
        }
        if (s == -1 && qp->lbAN[i] <= -ORC_INFTY) s = 0;
        if (s == 1 && qp->ubAN[i] >= ORC_INFTY) s = 0;
        if (s != 0 && constraint_is_LI(qp, i)) add_constraint(qp, i, s, 0, 0);
    }
    /* multipliers: zero when inactive, clipped to the sign their side requires */
    for (int v = 0; v < nV; v++) {
        if (qp->Sb[v] == 0) qp->y[v] = 0.0;
        else if (qp->Sb[v] == -1 && qp->y[v] < 0.0) qp->y[v] = 0.0;
        else if (qp->Sb[v] == 1 && qp->y[v] > 0.0) qp->y[v] = 0.0;
    }
    for (int i = 0; i < nC; i++) {
        double *yi = qp->y + nV + i;
        if (qp->Sc[i] == 0) *yi = 0.0;
        else if (qp->Sc[i] == -1 && *yi < 0.0) *yi = 0.0;
        else if (qp->Sc[i] == 1 && *yi > 0.0) *yi = 0.0;
    }
    /* gradient of the auxiliary QP: g = A'y_C + y_B - Hx */
    AT_times(qp, qp->y + nV, qp->wv1);
    H_times(qp, qp->x, qp->wv2);
    for (int v = 0; v < nV; v++) qp->g[v] = qp->wv1[v] + qp->y[v] - qp->wv2[v];
    /* bounds of the auxiliary QP */
    for (int v = 0; v < nV; v++) {
        double xv = qp->x[v];
        qp->lb[v] = qp->Sb[v] == -1 ? xv : fmin(qp->lbN[v], xv - BOUND_RELAXATION);
        qp->ub[v] = qp->Sb[v] == 1 ? xv : fmax(qp->ubN[v], xv + BOUND_RELAXATION);
    }
    for (int i = 0; i < nC; i++) {
        double ax = qp->Ax[i];
        qp->lbA[i] = qp->Sc[i] == -1 ? ax : fmin(qp->lbAN[i], ax - BOUND_RELAXATION);
        qp->ubA[i] = qp->Sc[i] == 1 ? ax : fmax(qp->ubAN[i], ax + BOUND_RELAXATION);
    }
    if (chol_setup(qp)) return ORC_RET_SETUP_FAILED;
    qp->status = ORC_QPS_AUXILIARYQPSOLVED;
    return ORC_RET_OK;
}

/* ---------------- step direction ---------------- */
static void step_direction(orc_qp *qp) {
    int nV = qp->nV, nC = qp->nC, nFR = qp->nFR, nAC = qp->nAC, nZ = nFR - nAC;
    double *dx = qp->dx, *dy = qp->dy;
    double *tmpg = qp->wv1, *t2 = qp->wv2, *wq = qp->wq, *bA = qp->wc1;
    for (int v = 0; v < nV; v++)
        dx[v] = qp->Sb[v] == -1 ? qp->dlb[v] : (qp->Sb[v] == 1 ? qp->dub[v] : 0.0);
    for (int i = 0; i < nV + nC; i++) dy[i] = 0.0;
    /* constraint right-hand side and gradient shift caused by the fixed variables */
    A_times(qp, dx, qp->wc2);
    for (int i = 0; i < nAC; i++) {
        int r = qp->AC[i];
        bA[i] = (qp->Sc[r] == -1 ? qp->dlbA[r] : qp->dubA[r]) - qp->wc2[r];
    }
    H_times(qp, dx, t2);
    for (int v = 0; v < nV; v++) tmpg[v] = qp->dg[v] + t2[v];
    /* range-space part: T wY = bA */
    for (int c = 0; c < nFR; c++) wq[c] = 0.0;
    for (int i = 0; i < nAC; i++) {
        int c = nFR - 1 - i;
        double s = bA[i];
        for (int cc = c + 1; cc < nFR; cc++) s -= qp->T[(size_t)i * nV + cc] * wq[cc];
        wq[c] = s / qp->T[(size_t)i * nV + c];
    }
    double *xY = qp->wv3;
    for (int v = 0; v < nV; v++) xY[v] = 0.0;
    for (int c = nZ; c < nFR; c++)
        for (int v = 0; v < nV; v++) xY[v] += qp->Q[(size_t)c * nV + v] * wq[c];
    /* null-space part: R'R wZ = -Z'(tmpg + H xY) */
    H_times(qp, xY, t2);
    for (int v = 0; v < nV; v++) t2[v] += tmpg[v];
    for (int j = 0; j < nZ; j++) wq[j] = -dotn(qp->Q + (size_t)j * nV, t2, nV);
    for (int j = 0; j < nZ; j++) { /* R' u = rhs */
        double s = wq[j];
        for (int k = 0; k < j; k++) s -= qp->R[(size_t)j * nV + k] * wq[k];
        wq[j] = s / qp->R[(size_t)j * nV + j];
    }
    for (int j = nZ - 1; j >= 0; j--) { /* R wZ = u */
        double s = wq[j];
        for (int k = j + 1; k < nZ; k++) s -= qp->R[(size_t)k * nV + j] * wq[k];
        wq[j] = s / qp->R[(size_t)j * nV + j];
    }
    for (int v = 0; v < nV; v++) {
        if (qp->Sb[v] != 0) continue;
        double s = xY[v];
        for (int j = 0; j < nZ; j++) s += qp->Q[(size_t)j * nV + v] * wq[j];
        dx[v] = s;
    }
    /* multipliers of the active constraints: T' dyAC = Y'(H dx + dg) */
    H_times(qp, dx, t2);
    for (int v = 0; v < nV; v++) t2[v] += qp->dg[v];
    for (int c = nZ; c < nFR; c++) wq[c] = dotn(qp->Q + (size_t)c * nV, t2, nV);
    for (int m = 0; m < nAC; m++) {
        int i = nAC - 1 - m, c = nZ + m;
        double s = wq[c];
        for (int ii = i + 1; ii < nAC; ii++) s -= qp->T[(size_t)ii * nV + c] * dy[nV + qp->AC[ii]];
        dy[nV + qp->AC[i]] = s / qp->T[(size_t)i * nV + c];
    }
    /* multipliers of the fixed variables */
    AT_times(qp, dy + nV, qp->wv3);
    for (int v = 0; v < nV; v++) dy[v] = qp->Sb[v] != 0 ? t2[v] - qp->wv3[v] : 0.0;
    A_times(qp, dx, qp->dAx);
}

/* ---------------- ratio tests ---------------- */
typedef struct {
    double tau;
    int kind; /* 0 none, 1 remove constraint, 2 remove bound, 3 add constr, 4 add bound */
    int idx, side;
} blocking_t;

static void ratio(double num, double den, double *tau, int *hit) {
    *hit = 0;
    /* qpOASES also demands num >= epsNum (= -1e3*EPS); a tie of two constraints at one
     * homotopy point leaves the second with a numerator of rounding size and EITHER sign,
     * and dropping it there loses the constraint for good -- so a negative numerator is
     * read as "already on the limit" (t = 0). */
    if (den >= EPS_DEN) {
        double t = (num > 0.0 ? num : 0.0) / den;
        if (t < *tau) {
            *tau = t;
            *hit = 1;
        }
    }
}

static blocking_t ratio_tests(const orc_qp *qp) {
    int nV = qp->nV, nC = qp->nC, hit;
    blocking_t b = {1.0, 0, -1, 0};
    for (int i = 0; i < nC; i++) { /* duals of active constraints */
        if (qp->Sc[i] == 0) continue;
        double yi = qp->y[nV + i], d = qp->dy[nV + i];
        if (qp->Sc[i] == -1) ratio(yi, -d, &b.tau, &hit);
        else ratio(-yi, d, &b.tau, &hit);
        if (hit) { b.kind = 1; b.idx = i; b.side = 0; }
    }
    for (int v = 0; v < nV; v++) { /* duals of fixed variables */
        if (qp->Sb[v] == 0) continue;
        double yi = qp->y[v], d = qp->dy[v];
        if (qp->Sb[v] == -1) ratio(yi, -d, &b.tau, &hit);
        else ratio(-yi, d, &b.tau, &hit);
        if (hit) { b.kind = 2; b.idx = v; b.side = 0; }
    }
    for (int i = 0; i < nC; i++) { /* inactive constraints, lower side */
        if (qp->Sc[i] != 0 || qp->lbAN[i] <= -ORC_INFTY) continue;
        ratio(qp->Ax[i] - qp->lbA[i], qp->dlbA[i] - qp->dAx[i], &b.tau, &hit);
        if (hit) { b.kind = 3; b.idx = i; b.side = -1; }
    }
    for (int i = 0; i < nC; i++) { /* inactive constraints, upper side */
        if (qp->Sc[i] != 0 || qp->ubAN[i] >= ORC_INFTY) continue;
        ratio(qp->ubA[i] - qp->Ax[i], qp->dAx[i] - qp->dubA[i], &b.tau, &hit);
        if (hit) { b.kind = 3; b.idx = i; b.side = 1; }
    }
    for (int v = 0; v < nV; v++) { /* free variables, lower side */
        if (qp->Sb[v] != 0 || qp->lbN[v] <= -ORC_INFTY) continue;
        ratio(qp->x[v] - qp->lb[v], qp->dlb[v] - qp->dx[v], &b.tau, &hit);
        if (hit) { b.kind = 4; b.idx = v; b.side = -1; }
    }
    for (int v = 0; v < nV; v++) { /* free variables, upper side */
        if (qp->Sb[v] != 0 || qp->ubN[v] >= ORC_INFTY) continue;
        ratio(qp->ub[v] - qp->x[v], qp->dx[v] - qp->dub[v], &b.tau, &hit);
        if (hit) { b.kind = 4; b.idx = v; b.side = 1; }
    }
    return b;
}

/* ---------------- removal with positive-definiteness guard ---------------- */
/* returns ORC_RET_OK or ORC_RET_UNBOUNDED */
static int remove_with_guard(orc_qp *qp, int is_bound, int idx) {
    int nV = qp->nV;
    if (is_bound) {
        int old = qp->Sb[idx];
        int zc = remove_bound_tq(qp, idx);
        qp->y[idx] = 0.0;
        if (chol_append(qp, zc) == 0) return ORC_RET_OK;
        /* Z'HZ would lose definiteness: put the variable back, on the opposite side
         * ("flipping bounds"); its multiplier is zero, so either sign is admissible */
        if ((old == -1 && qp->ubN[idx] >= ORC_INFTY) || (old == 1 && qp->lbN[idx] <= -ORC_INFTY)) {
            add_bound(qp, idx, old, 0, 1);
            return ORC_RET_UNBOUNDED;
        }
        add_bound(qp, idx, -old, 0, 1);
        if (old == -1) qp->ub[idx] = qp->x[idx];
        else qp->lb[idx] = qp->x[idx];
        qp->nflips++;
        return ORC_RET_OK;
    } else {
        int old = qp->Sc[idx], k = position_in_AC(qp, idx);
        int zc = remove_constraint_tq(qp, k);
        qp->y[nV + idx] = 0.0;
        if (chol_append(qp, zc) == 0) return ORC_RET_OK;
        if ((old == -1 && qp->ubAN[idx] >= ORC_INFTY) || (old == 1 && qp->lbAN[idx] <= -ORC_INFTY)) {
            add_constraint(qp, idx, old, 0, 1);
            return ORC_RET_UNBOUNDED;
        }
        add_constraint(qp, idx, -old, 0, 1);
        if (old == -1) qp->ubA[idx] = qp->Ax[idx];
        else qp->lbA[idx] = qp->Ax[idx];
        qp->nflips++;
        return ORC_RET_OK;
    }
}

/* ---------------- exchange when the incoming row is linearly dependent ---------------- */
/* a_full: full row (all variables) of the incoming constraint / unit vector of the bound.
 * Shifts the multipliers along the dependency until one active quantity reaches zero;
 * that one (pkind 1 = constraint, 2 = bound; pidx) has to leave the working set.
 * returns ORC_RET_OK or ORC_RET_INFEASIBLE */
static int ensure_LI(orc_qp *qp, const double *a_full, int side, double *y_new, int *pkind,
                     int *pidx) {
    int nV = qp->nV, nC = qp->nC, nFR = qp->nFR, nAC = qp->nAC, nZ = nFR - nAC;
    double *afr = qp->wv1, *wq = qp->wq, *xiC = qp->wc1, *xiB = qp->wv2;
    for (int v = 0; v < nV; v++) afr[v] = qp->Sb[v] == 0 ? a_full[v] : 0.0;
    QT_times(qp, afr, wq);
    /* T' xiC = Y' a_FR */
    for (int i = 0; i < nC; i++) qp->wc2[i] = 0.0;
    for (int m = 0; m < nAC; m++) {
        int i = nAC - 1 - m, c = nZ + m;
        double s = wq[c];
        for (int ii = i + 1; ii < nAC; ii++) s -= qp->T[(size_t)ii * nV + c] * xiC[ii];
        xiC[i] = s / qp->T[(size_t)i * nV + c];
    }
    for (int i = 0; i < nAC; i++) qp->wc2[qp->AC[i]] = xiC[i];
    AT_times(qp, qp->wc2, xiB);
    for (int v = 0; v < nV; v++) xiB[v] = qp->Sb[v] != 0 ? a_full[v] - xiB[v] : 0.0;
    /* incoming at its upper side: its multiplier grows in the negative direction */
    double sgn = side == 1 ? -1.0 : 1.0;
    double tmin = ORC_INFTY;
    int kind = 0, idx = -1;
    for (int i = 0; i < nC; i++) {
        if (qp->Sc[i] == 0) continue;
        double xi = sgn * qp->wc2[i], yi = qp->y[nV + i];
        double num = qp->Sc[i] == -1 ? yi : -yi, den = qp->Sc[i] == -1 ? xi : -xi;
        if (den > EPS_DEN) {
            double t = (num > 0.0 ? num : 0.0) / den;
            if (t < tmin) { tmin = t; kind = 1; idx = i; }
        }
    }
    for (int v = 0; v < nV; v++) {
        if (qp->Sb[v] == 0) continue;
        double xi = sgn * xiB[v], yi = qp->y[v];
        double num = qp->Sb[v] == -1 ? yi : -yi, den = qp->Sb[v] == -1 ? xi : -xi;
        if (den > EPS_DEN) {
            double t = (num > 0.0 ? num : 0.0) / den;
            if (t < tmin) { tmin = t; kind = 2; idx = v; }
        }
    }
    if (kind == 0) return ORC_RET_INFEASIBLE;
    for (int i = 0; i < nC; i++)
        if (qp->Sc[i] != 0) qp->y[nV + i] -= tmin * sgn * qp->wc2[i];
    for (int v = 0; v < nV; v++)
        if (qp->Sb[v] != 0) qp->y[v] -= tmin * sgn * xiB[v];
    *y_new = sgn * tmin;
    *pkind = kind;
    *pidx = idx;
    return ORC_RET_OK;
}

/* take the exchange partner out (TQ always; Cholesky when the enlarged Z'HZ is still
 * positive definite). Returns 1 if the Cholesky factor was extended, 0 if it was left
 * as it is -- then the caller adds the incoming row with skipZ (the incoming row is
 * orthogonal to every null-space column but the new one, so R stays valid). */
static int remove_partner(orc_qp *qp, int pkind, int pidx) {
    int zc;
    if (pkind == 1) {
        zc = remove_constraint_tq(qp, position_in_AC(qp, pidx));
        qp->y[qp->nV + pidx] = 0.0;
    } else {
        zc = remove_bound_tq(qp, pidx);
        qp->y[pidx] = 0.0;
    }
    return chol_append(qp, zc) == 0;
}

static int change_active_set(orc_qp *qp, blocking_t b) {
    int nV = qp->nV;
    if (b.kind == 1) return remove_with_guard(qp, 0, b.idx);
    if (b.kind == 2) return remove_with_guard(qp, 1, b.idx);
    if (b.kind == 3 || b.kind == 4) {
        double ynew = 0.0;
        int full = 1;
        int li = b.kind == 3 ? constraint_is_LI(qp, b.idx) : bound_is_LI(qp, b.idx);
        if (!li) {
            int pkind = 0, pidx = -1;
            double *a = (double *)xcalloc((size_t)nV, sizeof(double));
            if (b.kind == 3)
                for (int k = qp->Arp[b.idx]; k < qp->Arp[b.idx + 1]; k++) a[qp->Aci[k]] = qp->Arv[k];
            else a[b.idx] = 1.0;
            int rc = ensure_LI(qp, a, b.side, &ynew, &pkind, &pidx);
            free(a);
            if (rc != ORC_RET_OK) return rc;
            full = remove_partner(qp, pkind, pidx);
        }
        if (b.kind == 3) {
            add_constraint(qp, b.idx, b.side, full, !full);
            qp->y[nV + b.idx] = ynew;
        } else {
            add_bound(qp, b.idx, b.side, full, !full);
            qp->y[b.idx] = ynew;
        }
    }
    return ORC_RET_OK;
}

/* ---------------- homotopy ---------------- */
static double delta_of(double target, double cur) {
    if (fabs(target) >= ORC_INFTY && fabs(cur) >= ORC_INFTY) return 0.0;
    return target - cur;
}

/* make the CURRENT (auxiliary) QP exactly consistent with the iterate, so that rounding
 * errors do not accumulate along the homotopy (qpOASES enableDriftCorrection) */
static void drift_correction(orc_qp *qp) {
    int nV = qp->nV, nC = qp->nC;
    for (int v = 0; v < nV; v++)
        if (qp->Sb[v] != 0) qp->x[v] = qp->Sb[v] == -1 ? qp->lb[v] : qp->ub[v];
    A_times(qp, qp->x, qp->Ax);
    for (int i = 0; i < nC; i++) {
        if (qp->Sc[i] == -1) qp->lbA[i] = qp->Ax[i];
        else if (qp->Sc[i] == 1) qp->ubA[i] = qp->Ax[i];
    }
    AT_times(qp, qp->y + nV, qp->wv1);
    H_times(qp, qp->x, qp->wv2);
    for (int v = 0; v < nV; v++) qp->g[v] = qp->wv1[v] + qp->y[v] - qp->wv2[v];
}

static int homotopy(orc_qp *qp, int *nWSR) {
    int nV = qp->nV, nC = qp->nC, maxit = *nWSR, iter = 0, rc = ORC_RET_OK;
    qp->status = ORC_QPS_PERFORMINGHOMOTOPY;
    /* inactive sides only have to stay clear of the iterate: re-relax them when the
     * current value is infinite but the target is finite */
    for (int v = 0; v < nV; v++) {
        if (qp->Sb[v] != -1 && qp->lb[v] <= -ORC_INFTY && qp->lbN[v] > -ORC_INFTY)
            qp->lb[v] = fmin(qp->lbN[v], qp->x[v] - BOUND_RELAXATION);
        if (qp->Sb[v] != 1 && qp->ub[v] >= ORC_INFTY && qp->ubN[v] < ORC_INFTY)
            qp->ub[v] = fmax(qp->ubN[v], qp->x[v] + BOUND_RELAXATION);
    }
    for (int i = 0; i < nC; i++) {
        if (qp->Sc[i] != -1 && qp->lbA[i] <= -ORC_INFTY && qp->lbAN[i] > -ORC_INFTY)
            qp->lbA[i] = fmin(qp->lbAN[i], qp->Ax[i] - BOUND_RELAXATION);
        if (qp->Sc[i] != 1 && qp->ubA[i] >= ORC_INFTY && qp->ubAN[i] < ORC_INFTY)
            qp->ubA[i] = fmax(qp->ubAN[i], qp->Ax[i] + BOUND_RELAXATION);
    }
    for (;;) {
        for (int v = 0; v < nV; v++) {
            qp->dg[v] = qp->gN[v] - qp->g[v];
            qp->dlb[v] = delta_of(qp->lbN[v], qp->lb[v]);
            qp->dub[v] = delta_of(qp->ubN[v], qp->ub[v]);
        }
        for (int i = 0; i < nC; i++) {
            qp->dlbA[i] = delta_of(qp->lbAN[i], qp->lbA[i]);
            qp->dubA[i] = delta_of(qp->ubAN[i], qp->ubA[i]);
        }
        step_direction(qp);
        blocking_t b = ratio_tests(qp);
        double tau = b.tau;
        for (int v = 0; v < nV; v++) {
            qp->x[v] += tau * qp->dx[v];
            qp->g[v] += tau * qp->dg[v];
            qp->lb[v] += tau * qp->dlb[v];
            qp->ub[v] += tau * qp->dub[v];
        }
        for (int i = 0; i < nV + nC; i++) qp->y[i] += tau * qp->dy[i];
        for (int i = 0; i < nC; i++) {
            qp->lbA[i] += tau * qp->dlbA[i];
            qp->ubA[i] += tau * qp->dubA[i];
        }
        if (b.kind == 0) { /* full step: the data now ARE the targets */
            for (int v = 0; v < nV; v++) {
                qp->g[v] = qp->gN[v];
                qp->lb[v] = qp->lbN[v];
                qp->ub[v] = qp->ubN[v];
                if (qp->Sb[v] != 0) qp->x[v] = qp->Sb[v] == -1 ? qp->lb[v] : qp->ub[v];
            }
            for (int i = 0; i < nC; i++) {
                qp->lbA[i] = qp->lbAN[i];
                qp->ubA[i] = qp->ubAN[i];
            }
            A_times(qp, qp->x, qp->Ax);
            qp->status = ORC_QPS_SOLVED;
            break;
        }
        if (iter >= maxit) {
            A_times(qp, qp->x, qp->Ax);
            rc = ORC_RET_MAX_NWSR;
            break;
        }
        A_times(qp, qp->x, qp->Ax);
        /* the blocking quantity sits exactly on its limit */
        if (b.kind == 3) {
            if (b.side == -1) qp->lbA[b.idx] = qp->Ax[b.idx];
            else qp->ubA[b.idx] = qp->Ax[b.idx];
        } else if (b.kind == 4) {
            if (b.side == -1) qp->lb[b.idx] = qp->x[b.idx];
            else qp->ub[b.idx] = qp->x[b.idx];
        }
        rc = change_active_set(qp, b);
        if (rc == ORC_RET_INFEASIBLE) { qp->infeasible = 1; break; }
        if (rc == ORC_RET_UNBOUNDED) { qp->unbounded = 1; break; }
        iter++;
        drift_correction(qp);
    }
    *nWSR = iter;
    return rc;
}


/* qpOASES rejects inconsistent data up front (areBoundsConsistent): lb > ub or lbA > ubA
 * makes the QP infeasible before any working-set change */
static int bounds_inconsistent(const orc_qp *qp) {
    for (int v = 0; v < qp->nV; v++)
        if (qp->lbN[v] > qp->ubN[v] + ORC_EPS) return 1;
    for (int i = 0; i < qp->nC; i++)
        if (qp->lbAN[i] > qp->ubAN[i] + ORC_EPS) return 1;
    return 0;
}

/* ---------------- public solve entry points ---------------- */
int orc_qp_init(orc_qp *qp, const double *g, const double *lb, const double *ub,
                const double *lbA, const double *ubA, int *nWSR, const double *x0,
                const double *y0, const int *guess_b) {
    store_targets(qp, g, lb, ub, lbA, ubA);
    qp->nflips = 0;
    if (bounds_inconsistent(qp)) {
        qp->infeasible = 1; qp->unbounded = 0;
        *nWSR = 0;
        return ORC_RET_INFEASIBLE;
    }
    int rc = setup_aux(qp, x0, y0, guess_b, NULL);
    if (rc != ORC_RET_OK && (x0 || y0 || guess_b)) rc = setup_aux(qp, NULL, NULL, NULL, NULL);
    if (rc != ORC_RET_OK) {
        *nWSR = 0;
        return rc;
    }
    return homotopy(qp, nWSR);
}

int orc_qp_hotstart(orc_qp *qp, const double *g, const double *lb, const double *ub,
                    const double *lbA, const double *ubA, int *nWSR) {
    if (qp->status == ORC_QPS_NOTINITIALISED) return ORC_RET_SETUP_FAILED;
    store_targets(qp, g, lb, ub, lbA, ubA);
    qp->infeasible = qp->unbounded = 0;
    if (bounds_inconsistent(qp)) {
        qp->infeasible = 1;
        *nWSR = 0;
        return ORC_RET_INFEASIBLE;
    }
    return homotopy(qp, nWSR);
}

int orc_qp_hotstart_matrices(orc_qp *qp, const double *g, const double *lb, const double *ub,
                             const double *lbA, const double *ubA, int *nWSR) {
    if (qp->status == ORC_QPS_NOTINITIALISED) return ORC_RET_SETUP_FAILED;
    int nV = qp->nV, nC = qp->nC;
    store_targets(qp, g, lb, ub, lbA, ubA);
    if (bounds_inconsistent(qp)) {
        qp->infeasible = 1; qp->unbounded = 0;
        *nWSR = 0;
        return ORC_RET_INFEASIBLE;
    }
    double *x0 = (double *)xcalloc((size_t)nV, sizeof(double));
    double *y0 = (double *)xcalloc((size_t)nV + nC, sizeof(double));
    int *gb = (int *)xcalloc((size_t)nV, sizeof(int)), *gc = (int *)xcalloc((size_t)nC, sizeof(int));
    memcpy(x0, qp->x, sizeof(double) * (size_t)nV);
    memcpy(y0, qp->y, sizeof(double) * ((size_t)nV + nC));
    memcpy(gb, qp->Sb, sizeof(int) * (size_t)nV);
    memcpy(gc, qp->Sc, sizeof(int) * (size_t)nC);
    int rc = setup_aux(qp, x0, y0, gb, gc);
    if (rc != ORC_RET_OK) rc = setup_aux(qp, NULL, NULL, NULL, NULL);
    free(x0); free(y0); free(gb); free(gc);
    if (rc != ORC_RET_OK) {
        *nWSR = 0;
        return rc;
    }
    return homotopy(qp, nWSR);
}

void orc_qp_set_regularisation(orc_qp *qp, double reg) { qp->hreg = reg; }
void orc_qp_set_guess_constraints_from_y0(orc_qp *qp, int on) { qp->guess_c_from_y0 = on != 0; }

void orc_qp_get_primal(const orc_qp *qp, double *x) { memcpy(x, qp->x, sizeof(double) * (size_t)qp->nV); }
void orc_qp_get_dual(const orc_qp *qp, double *y) {
    memcpy(y, qp->y, sizeof(double) * ((size_t)qp->nV + qp->nC));
}
double orc_qp_get_objective(const orc_qp *qp) {
    double *Hx = (double *)xcalloc((size_t)qp->nV, sizeof(double));
    H_times(qp, qp->x, Hx);
    double o = 0.5 * (dotn(qp->x, Hx, qp->nV) - qp->hreg * dotn(qp->x, qp->x, qp->nV)) + dotn(qp->gN, qp->x, qp->nV);
    free(Hx);
    return o;
}
void orc_qp_get_working_set_bounds(const orc_qp *qp, int *ws) { memcpy(ws, qp->Sb, sizeof(int) * (size_t)qp->nV); }
void orc_qp_get_working_set_constraints(const orc_qp *qp, int *ws) {
    memcpy(ws, qp->Sc, sizeof(int) * (size_t)qp->nC);
}
int orc_qp_status(const orc_qp *qp) { return qp->status; }
int orc_qp_is_solved(const orc_qp *qp) { return qp->status == ORC_QPS_SOLVED; }
int orc_qp_is_infeasible(const orc_qp *qp) { return qp->infeasible; }
int orc_qp_is_unbounded(const orc_qp *qp) { return qp->unbounded; }
int orc_qp_nflips(const orc_qp *qp) { return qp->nflips; }

int orc_exitflag(const orc_qp *qp) {
    /* include/sqphot/Types.hpp:51-73 via src/qpOASESInterface.cpp:332-357 */
    if (qp->infeasible) return 22;
    if (qp->unbounded) return 23;
    if (qp->status == ORC_QPS_SOLVED) return 20;
    switch (qp->status) {
    case ORC_QPS_NOTINITIALISED: return 25;
    case ORC_QPS_PREPARINGAUXILIARYQP: return 26;
    case ORC_QPS_AUXILIARYQPSOLVED: return 27;
    case ORC_QPS_PERFORMINGHOMOTOPY: return 28;
    case ORC_QPS_HOMOTOPYQPSOLVED: return 29;
    }
    return 30;
}

/* cold init repeated `reps` times inside C (bench.py's cpu_baseline: keeps the interpreter out of the
 * timed loop). Returns the nWSR of the last solve. */
int orc_qp_init_repeat(orc_qp *qp, const double *g, const double *lb, const double *ub, const double *lbA,
                       const double *ubA, int nWSR_max, int reps) {
    int n = 0;
    for (int r = 0; r < reps; r++) {
        n = nWSR_max;
        orc_qp_init(qp, g, lb, ub, lbA, ubA, &n, 0, 0, 0);
    }
    return n;
}

/* QPhandler::solveQP on the CPU, `iters` times inside C (bench.py's hs071_single_qp leg: the interpreter and
 * ctypes stay out of the timed loop): hot start on the vector set A or B in turn (update_delta, FIXED matrices),
 * then the working-set mapping and the KKT certificate of the reference (kkt_oracle.c). vec[k] = {g, lb, ub, lbA,
 * ubA} of set k, already clamped to +-1e20 by the caller. Returns the number of certified solves. */
int orc_qp_solveqp_repeat(orc_qp *qp, const double *const *vecA, const double *const *vecB, int nWSR_max, int iters) {
    int nV = qp->nV, nC = qp->nC, good = 0;
    int *Wb = (int *)xcalloc((size_t)nV, sizeof(int)), *Wc = (int *)xcalloc((size_t)nC, sizeof(int));
    for (int it = 0; it < iters; it++) {
        const double *const *v = (it & 1) ? vecB : vecA;
        int n = nWSR_max;
        orc_qp_hotstart(qp, v[0], v[1], v[2], v[3], v[4], &n);
        orc_optimality_status st;
        if (orc_kkt_get_working_set(nV, nC, qp->Ajc, qp->Air, qp->Aval, qp->x, v[1], v[2], v[3], v[4], qp->Sb, qp->Sc, Wb, Wc) == 0 &&
            orc_kkt_test_optimality(nV, nC, qp->Ajc, qp->Air, qp->Aval, qp->haveH ? qp->Hjc : 0, qp->haveH ? qp->Hir : 0,
                                    qp->haveH ? qp->Hval : 0, v[0], v[1], v[2], v[3], v[4], qp->x, qp->y, Wb, Wc, &st) == 1)
            good++;
    }
    free(Wb); free(Wc);
    return good;
}
