/*
 * kkt_oracle.c -- CPU ORACLE (test infrastructure, not product code).
 * Restates qpOASESInterface::get_working_set (src/qpOASESInterface.cpp:835-895),
 * ::test_optimality (:498-684) and ::get_status (:332-357).
 */
#include "rsqp_oracle.h"

#include <math.h>
#include <stdlib.h>

#define ACTIVE_ABOVE 1      /* include/sqphot/Types.hpp:84-89 */
#define ACTIVE_BELOW (-1)
#define ACTIVE_BOTH_SIDE (-99)
#define INACTIVE 0

static const double sqrt_m_eps = 1.0e-8; /* include/sqphot/Utils.hpp:37 */

int orc_kkt_get_working_set(int nV, int nC, const int *Ajc, const int *Air, const double *Aval,
                            const double *x, const double *lb, const double *ub,
                            const double *lbA, const double *ubA, const int *ws_b,
                            const int *ws_c, int *W_b, int *W_c) {
    /* bounds (:846-868): solver +1 maps to ACTIVE_ABOVE, -1 to ACTIVE_BELOW */
    for (int i = 0; i < nV; i++) {
        switch (ws_b[i]) {
        case 1:
            W_b[i] = fabs(x[i] - lb[i]) < sqrt_m_eps ? ACTIVE_BOTH_SIDE : ACTIVE_ABOVE;
            break;
        case -1:
            W_b[i] = fabs(x[i] - ub[i]) < sqrt_m_eps ? ACTIVE_BOTH_SIDE : ACTIVE_BELOW;
            break;
        case 0:
            W_b[i] = INACTIVE;
            break;
        default:
            return -1; /* INVALID_WORKING_SET */
        }
    }
    double *Ax = (double *)calloc((size_t)(nC > 0 ? nC : 1), sizeof(double));
    orc_sphb_times(nC, nV, 0, Ajc, Air, Aval, x, Ax);
    /* constraints (:871-892). The reference writes fabs(Ax-lbA<sqrt_m_eps): the comparison
     * is inside fabs, so BOTH_SIDE is decided by the SIGNED test Ax - lbA < 1e-8. */
    for (int i = 0; i < nC; i++) {
        switch (ws_c[i]) {
        case 1:
            W_c[i] = (Ax[i] - lbA[i] < sqrt_m_eps) ? ACTIVE_BOTH_SIDE : ACTIVE_ABOVE;
            break;
        case -1:
            W_c[i] = (Ax[i] - ubA[i] < sqrt_m_eps) ? ACTIVE_BOTH_SIDE : ACTIVE_BELOW;
            break;
        case 0:
            W_c[i] = INACTIVE;
            break;
        default:
            free(Ax);
            return -1;
        }
    }
    free(Ax);
    return 0;
}

int orc_kkt_test_optimality(int nV, int nC, const int *Ajc, const int *Air, const double *Aval,
                            const int *Hjc, const int *Hir, const double *Hval,
                            const double *g, const double *lb, const double *ub,
                            const double *lbA, const double *ubA, const double *x,
                            const double *y, const int *W_b, const int *W_c,
                            orc_optimality_status *out) {
    double primal = 0.0, dual = 0.0, compl = 0.0, stat = 0.0;
    double *Ax = (double *)calloc((size_t)(nC > 0 ? nC : 1), sizeof(double));
    double *gap = (double *)calloc((size_t)(nV > 0 ? nV : 1), sizeof(double));
    double *Hx = (double *)calloc((size_t)(nV > 0 ? nV : 1), sizeof(double));
    int rc = 1;

    /* primal feasibility (:518-528) */
    for (int i = 0; i < nV; i++) {
        primal += fmax(0.0, lb[i] - x[i]);
        primal += -fmin(0.0, ub[i] - x[i]);
    }
    orc_sphb_times(nC, nV, 0, Ajc, Air, Aval, x, Ax);
    for (int i = 0; i < nC; i++) {
        primal += fmax(0.0, lbA[i] - Ax[i]);
        primal += -fmin(0.0, ubA[i] - Ax[i]);
    }

    /* dual feasibility (:533-578) */
    for (int i = 0; i < nV + nC; i++) {
        int w = i < nV ? W_b[i] : W_c[i - nV];
        switch (w) {
        case INACTIVE: dual += fabs(y[i]); break;
        case ACTIVE_BELOW: dual += -fmin(0.0, y[i]); break;
        case ACTIVE_ABOVE: dual += fmax(0.0, y[i]); break;
        case ACTIVE_BOTH_SIDE: break;
        default: rc = -1; goto done;
        }
    }

    /* stationarity ||A'y_c + y_b - g - Hx||_1 (:595-604) */
    orc_sphb_transposed_times(nC, nV, 0, Ajc, Air, Aval, y + nV, gap);
    if (Hjc) orc_sphb_times(nV, nV, 0, Hjc, Hir, Hval, x, Hx);
    for (int i = 0; i < nV; i++) {
        gap[i] += y[i];
        gap[i] -= g[i];
        gap[i] -= Hx[i];
    }
    stat = orc_one_norm(gap, nV);

    /* complementarity (:611-658) */
    for (int i = 0; i < nV; i++) {
        switch (W_b[i]) {
        case INACTIVE: compl += fabs(y[i]); break;
        case ACTIVE_BELOW: compl += fabs(y[i] * (x[i] - lb[i])); break;
        case ACTIVE_ABOVE: compl += fabs(y[i] * (ub[i] - x[i])); break;
        case ACTIVE_BOTH_SIDE: break;
        default: rc = -1; goto done;
        }
    }
    for (int i = 0; i < nC; i++) {
        switch (W_c[i]) {
        case INACTIVE: compl += fabs(y[i + nV]); break;
        case ACTIVE_BELOW: compl += fabs(y[i + nV] * (Ax[i] - lbA[i])); break;
        case ACTIVE_ABOVE: compl += fabs(y[i + nV] * (ubA[i] - Ax[i])); break;
        case ACTIVE_BOTH_SIDE: break;
        default: rc = -1; goto done;
        }
    }

    out->primal_violation = primal;
    out->dual_violation = dual;
    out->compl_violation = compl;
    out->stationarity_violation = stat;
    out->KKT_error = compl + stat + dual + primal;
    rc = out->KKT_error > 1.0e-6 ? 0 : 1; /* :673 */
done:
    free(Ax);
    free(gap);
    free(Hx);
    return rc;
}
