/*
 * traj_oracle.c -- CPU ORACLE (test infrastructure, NOT product code): the QP side of a whole hs071 SQP run, in one C loop.
 *
 * What the reference clocks per pass of its while loop (src/Algorithm.cpp:57,138-139) minus NLP evaluation in Ipopt/ASL:
 * Algorithm::setupQP (src/Algorithm.cpp:645-697 -> QPhandler set_A / set_H / set_bounds / set_g at iteration 0, update_* later:
 * src/QPhandler.cpp:167-201,272-297,342-368,430-463,508-531), QPhandler::solveQP (:470-499 = optimizeQP with the dispatch of
 * src/qpOASESInterface.cpp:137-224 + get_working_set :835-895 + test_optimality :498-684) and the getters the loop reads.
 * Used by bench.py as the CPU leg beside `host_replay --trajectory` (the same loop through the HIP boundary) and by the tests
 * to check that leg against the committed trajectory (tests/golden/sqp_traces.json). Parity status of the solver inside:
 * unpinned vs qpOASES (rsqp_oracle.h).
 */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rsqp_oracle.h"

/* hs071 in closed form (test/CUTE_examples/hs071.nl): gradient, constraints, Jacobian (1-based COO, 8 entries), Hessian of the
 * Lagrangian f - lam'c (lower triangle row by row, 10 entries; SQPTNLP::Eval_Hessian negates lambda, src/SQPTNLP.cpp:124-126) */
static void hs071_eval(const double *x, const double *lam, double *grad, double *c, double *Jv, double *Hv) {
    const double x1 = x[0], x2 = x[1], x3 = x[2], x4 = x[3], l1 = lam[0], l2 = lam[1];
    grad[0] = x4 * (2 * x1 + x2 + x3); grad[1] = x1 * x4; grad[2] = x1 * x4 + 1.0; grad[3] = x1 * (x1 + x2 + x3);
    c[0] = x1 * x2 * x3 * x4; c[1] = x1 * x1 + x2 * x2 + x3 * x3 + x4 * x4;
    const double J[8] = {x2 * x3 * x4, x1 * x3 * x4, x1 * x2 * x4, x1 * x2 * x3, 2 * x1, 2 * x2, 2 * x3, 2 * x4};
    memcpy(Jv, J, sizeof(J));
    const double hf[10] = {2 * x4, x4, 0, x4, 0, 0, 2 * x1 + x2 + x3, x1, x1, 0};
    const double hc1[10] = {0, x3 * x4, 0, x2 * x4, x1 * x4, 0, x2 * x3, x1 * x3, x1 * x2, 0};
    const int diag[10] = {1, 0, 1, 0, 0, 1, 0, 0, 0, 1};
    for (int e = 0; e < 10; e++) Hv[e] = hf[e] - l1 * hc1[e] - (diag[e] ? 2.0 * l2 : 0.0);
}

static double now_us(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e6 * (double)ts.tv_sec + 1e-3 * (double)ts.tv_nsec;
}

/* traj: nit rows of {delta, rho, x[4], lam[2]}. Returns 0, or the 1-based iteration whose QP was not solved / not certified.
 * out_us[0] = mean microseconds per SQP iteration over reps x nit, out_us[1] = mean of iteration 0, out_us[2] = mean of the later
 * ones; x_last[8] / y_last[10]: primal / dual of the last QP; qp_iter: working-set changes of one run; modes[nit]: 0 cold,
 * 1 hotstart(vectors), 2 hotstart(matrices), 3 re-init. */
int orc_hs071_trajectory_replay(int nit, const double *traj, int reps, double *out_us, double *x_last, double *y_last,
                                int *qp_iter, int *modes) {
    enum { n = 4, m = 2, nV = 8, nC = 2, NA = 12, NH = 16 };
    static const int Jr[8] = {1, 1, 1, 1, 2, 2, 2, 2}, Jc[8] = {1, 2, 3, 4, 1, 2, 3, 4};
    static const int Hr[10] = {1, 2, 2, 3, 3, 3, 4, 4, 4, 4}, Hc[10] = {1, 1, 2, 1, 2, 3, 1, 2, 3, 4};
    static const int id_irow[2] = {1, 1}, id_jcol[2] = {n + 1, n + m + 1}, id_size[2] = {m, m};
    static const double id_val[2] = {1.0, -1.0};
    static const double x_l[4] = {1, 1, 1, 1}, x_u[4] = {5, 5, 5, 5}, c_l[2] = {25, 40};
    const double c_u[2] = {INFINITY, 40};
    double us_all = 0.0, us_first = 0.0;
    int bad = 0;
    for (int r = 0; r < reps && !bad; r++) {
        orc_qp *qp = orc_qp_create(nV, nC);
        int Ajc[nV + 1], Air[NA], Aord[NA], Hjc[nV + 1], Hir[NH], Hord[NH];
        double Aval[NA], Hval[NH], g[nV], lb[nV], ub[nV], lbA[nC], ubA[nC], x[nV], y[nV + nC];
        int wsb[nV], wsc[nC], Wb[nV], Wc[nC];
        memset(lb, 0, sizeof(lb)); memset(ub, 0, sizeof(ub));
        int first_solved = 0, upd = 0, old = 0, new_ = 0, iters = 0;
        double rho_prev = 0.0;
        for (int k = 0; k < nit && !bad; k++) {
            const double *t = traj + 8 * k, delta = t[0], rho = t[1];
            const double t0 = now_us();
            double grad[n], c[m], Jv[8], Hv[10];
            hs071_eval(t + 2, t + 6, grad, c, Jv, Hv);
            if (k == 0) {
                orc_sphb_set_structure(nC, nV, 8, Jr, Jc, Jv, 2, id_irow, id_jcol, id_size, id_val, 0, Ajc, Air, Aval, Aord);
                orc_sphb_set_structure_sym(nV, nV, 10, Hr, Hc, Hv, 1, 0, Hjc, Hir, Hval, Hord);
                orc_handler_set_bounds(n, m, delta, x_l, x_u, t + 2, c_l, c_u, c, lb, ub, lbA, ubA);
                orc_handler_set_g(n, m, grad, rho, g);
            } else {
                orc_sphb_set_matval(NA, 2 * m, Aord, Jv, Aval);
                orc_sphb_set_matval_sym(10, Hr, Hc, 1, Hord, Hv, Hval);
                orc_handler_update_bounds(n, m, delta, x_l, x_u, t + 2, c_l, c, lb, ub, lbA);
                /* NOT the reference: its qpOASES branch leaves ubA stale (src/QPhandler.cpp:358-360), which turns its own run
                 * infeasible after the first accepted step when a constraint is an equality (hs071's c2); a whole-trajectory
                 * replay needs the value (same statement in restartsqp_amd/handler.py::update_bounds, refresh_ubA) */
                for (int i = 0; i < m; i++) ubA[i] = c_u[i] - c[i];
                if (rho != rho_prev) for (int i = n; i < nV; i++) g[i] = rho;
                for (int i = 0; i < n; i++) g[i] = grad[i];
                if (first_solved) upd = 1;
            }
            rho_prev = rho;
            orc_qp_set_A_csc(qp, Ajc, Air, Aval);
            orc_qp_set_H_csc(qp, Hjc, Hir, Hval);
            /* optimizeQP (src/qpOASESInterface.cpp:137-224) */
            int nWSR = 1000, mode;
            if (!first_solved) {
                orc_qp_init(qp, g, lb, ub, lbA, ubA, &nWSR, NULL, NULL, NULL);
                mode = 0;
                if (orc_qp_is_solved(qp)) first_solved = 1;
            } else {
                const int cur = upd ? 2 : 1;
                if (old == 0) old = cur;
                else { if (new_ != 0) old = new_; new_ = cur; }
                if (new_ == 0 || new_ == old) {
                    const int st = new_ == 0 ? old : new_;
                    if (st == 1) { orc_qp_hotstart(qp, g, lb, ub, lbA, ubA, &nWSR); mode = 1; }
                    else { orc_qp_hotstart_matrices(qp, g, lb, ub, lbA, ubA, &nWSR); mode = 2; }
                } else {
                    orc_qp_get_primal(qp, x); orc_qp_get_dual(qp, y); orc_qp_get_working_set_bounds(qp, wsb);
                    orc_qp_init(qp, g, lb, ub, lbA, ubA, &nWSR, x, y, wsb);
                    mode = 3;
                    new_ = old = 0;
                }
            }
            upd = 0;
            iters += nWSR;
            if (modes && r == 0) modes[k] = mode;
            /* getPrimalSolution / getDualSolution (:221-222), then QPhandler::test_optimality (src/QPhandler.cpp:495, 580) */
            orc_qp_get_primal(qp, x); orc_qp_get_dual(qp, y);
            orc_qp_get_working_set_bounds(qp, wsb); orc_qp_get_working_set_constraints(qp, wsc);
            orc_optimality_status st;
            int ok = orc_kkt_get_working_set(nV, nC, Ajc, Air, Aval, x, lb, ub, lbA, ubA, wsb, wsc, Wb, Wc) == 0 &&
                     orc_kkt_test_optimality(nV, nC, Ajc, Air, Aval, Hjc, Hir, Hval, g, lb, ub, lbA, ubA, x, y, Wb, Wc, &st) == 1;
            if (!orc_qp_is_solved(qp) || !ok) bad = k + 1;
            const double us = now_us() - t0;
            us_all += us;
            if (k == 0) us_first += us;
        }
        if (x_last) memcpy(x_last, x, sizeof(x));
        if (y_last) memcpy(y_last, y, sizeof(y));
        if (qp_iter) *qp_iter = iters;
        orc_qp_destroy(qp);
    }
    if (out_us) {
        out_us[0] = us_all / ((double)reps * nit);
        out_us[1] = us_first / reps;
        out_us[2] = nit > 1 ? (us_all - us_first) / ((double)reps * (nit - 1)) : 0.0;
    }
    return bad;
}
