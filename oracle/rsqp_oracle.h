/*
 * rsqp_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the QP-subproblem hot path of lanl-ansi/RestartSQP
 * (SURVEY.md section 8). Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (restartsqp_amd/) never
 * links, imports or calls it.
 *
 * PARITY STATUS
 *   - containers (orc_sphb_*): restate src/SpHbMat.cpp / src/SpTripletMat.cpp;
 *     pinned by the fixture recorded in SURVEY.md 8(c) and by the properties the
 *     reference's own unit tests check (test/unitTest/test_SpHbMat.cpp:404-479).
 *   - KKT certificate / working-set mapping (orc_kkt_*): restate
 *     src/qpOASESInterface.cpp:498-684, 835-895 line by line.
 *   - active-set solver (orc_qp_*): the arithmetic lives in qpOASES 3.2.1
 *     (CMakeLists.txt:80-94, cmake/ExternalQPOASES.cmake:2-16), which is NOT in
 *     /root/reference and not in this image. It is restated from the published
 *     online active-set (parametric homotopy) algorithm; the reference holds no
 *     expected x / y / working set for any QP.  ==> "parity unpinned" vs qpOASES.
 */
#ifndef RSQP_ORACLE_H
#define RSQP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_INFTY 1.0e20              /* qpOASES INFTY */
#define ORC_EPS 2.221e-16             /* qpOASES EPS   */

/* ---------------------------------------------------------------------- */
/* containers: SpHbMat / SpTripletMat                                      */
/* ---------------------------------------------------------------------- */

/* SpHbMat::setStructure(rhs, I_info)  src/SpHbMat.cpp:196-268 (compressed column)
 * in : 1-based COO (irow,jcol,val)[nnz_t] + identity blocks; out: CSC arrays
 *      jc[ncol+1], ir[nnz], val[nnz], order[nnz] with nnz = nnz_t + sum(id_size). */
int orc_sphb_set_structure(int nrow, int ncol, int nnz_t, const int *irow, const int *jcol,
                           const double *val, int n_ident, const int *id_irow,
                           const int *id_jcol, const int *id_size, const double *id_value,
                           int compressed_row, int *ptr, int *idx, double *out_val, int *order);

/* SpHbMat::setStructure(rhs) for a (possibly symmetric, one-triangle) triplet matrix
 * src/SpHbMat.cpp:284-355. Returns the number of CSC entries (mirrored off-diagonals). */
int orc_sphb_sym_nnz(int nnz_t, const int *irow, const int *jcol, int is_symmetric);
int orc_sphb_set_structure_sym(int nrow, int ncol, int nnz_t, const int *irow, const int *jcol,
                               const double *val, int is_symmetric, int compressed_row,
                               int *ptr, int *idx, double *out_val, int *order);

/* SpHbMat::setMatVal(rhs, I_info) src/SpHbMat.cpp:368-380 */
void orc_sphb_set_matval(int nnz_total, int n_ident_entries, const int *order,
                         const double *triplet_val, double *matval);
/* SpHbMat::setMatVal(rhs) src/SpHbMat.cpp:383-393 */
void orc_sphb_set_matval_sym(int nnz_t, const int *irow, const int *jcol, int is_symmetric,
                             const int *order, const double *triplet_val, double *matval);

/* SpHbMat::times src/SpHbMat.cpp:698-737 and transposed_times :659-696.
 * (ptr,idx) is CSC when compressed_row==0, CSR otherwise. Entry-order accumulation. */
void orc_sphb_times(int nrow, int ncol, int compressed_row, const int *ptr, const int *idx,
                    const double *val, const double *p, double *result);
void orc_sphb_transposed_times(int nrow, int ncol, int compressed_row, const int *ptr,
                               const int *idx, const double *val, const double *p,
                               double *result);

/* SpHbMat dense ctor src/SpHbMat.cpp:59-165 (row_oriented data -> CSC / CSR). */
int orc_sphb_from_dense(const double *data, int nrow, int ncol, int row_oriented,
                        int compressed_row, int *ptr, int *idx, double *val);
/* SpHbMat::get_dense_matrix src/SpHbMat.cpp:588-656 (row oriented output) */
void orc_sphb_to_dense(int nrow, int ncol, int compressed_row, const int *ptr, const int *idx,
                       const double *val, double *dense_row_major);

/* SpTripletMat::times / transposed_times src/SpTripletMat.cpp:237-258, 311-323 */
void orc_triplet_times(int nrow, int ncol, int nnz, const int *irow, const int *jcol,
                       const double *val, int is_symmetric, const double *p, double *result);
void orc_triplet_transposed_times(int nrow, int ncol, int nnz, const int *irow,
                                  const int *jcol, const double *val, int is_symmetric,
                                  const double *p, double *result);

/* oneNorm / infNorm  src/Utils.cpp:65-83 */
double orc_one_norm(const double *x, int n);
double orc_inf_norm(const double *x, int n);

/* ---------------------------------------------------------------------- */
/* QPhandler formulas (caller side of the boundary)                        */
/* ---------------------------------------------------------------------- */
/* QPhandler::set_bounds src/QPhandler.cpp:185-201 (NEW_FORMULATION=false, non-QORE branch).
 * nV_qp = n + 2m; lb/ub have nV_qp entries (slack lb stay as they were: zero-init). */
void orc_handler_set_bounds(int n, int m, double delta, const double *x_l, const double *x_u,
                            const double *x_k, const double *c_l, const double *c_u,
                            const double *c_k, double *lb, double *ub, double *lbA,
                            double *ubA);
/* QPhandler::update_bounds src/QPhandler.cpp:342-368: lbA, lb, ub only (ubA NOT refreshed). */
void orc_handler_update_bounds(int n, int m, double delta, const double *x_l,
                               const double *x_u, const double *x_k, const double *c_l,
                               const double *c_k, double *lb, double *ub, double *lbA);
/* QPhandler::set_g src/QPhandler.cpp:272-297 */
void orc_handler_set_g(int n, int m, const double *grad, double rho, double *g);

/* ---------------------------------------------------------------------- */
/* KKT certificate and working-set mapping                                 */
/* ---------------------------------------------------------------------- */
typedef struct {
    double primal_violation, dual_violation, compl_violation, stationarity_violation, KKT_error;
} orc_optimality_status;

/* qpOASESInterface::get_working_set src/qpOASESInterface.cpp:835-895.
 * ws_b / ws_c: solver convention (+1 upper, -1 lower, 0 inactive).
 * W_b / W_c : ActiveType (1 ABOVE, -1 BELOW, -99 BOTH_SIDE, 0 INACTIVE), quirks kept. */
int orc_kkt_get_working_set(int nV, int nC, const int *Ajc, const int *Air, const double *Aval,
                            const double *x, const double *lb, const double *ub,
                            const double *lbA, const double *ubA, const int *ws_b,
                            const int *ws_c, int *W_b, int *W_c);
/* qpOASESInterface::test_optimality src/qpOASESInterface.cpp:498-684. Returns 1 if
 * KKT_error <= 1e-6, 0 otherwise, -1 on invalid working set. */
int orc_kkt_test_optimality(int nV, int nC, const int *Ajc, const int *Air, const double *Aval,
                            const int *Hjc, const int *Hir, const double *Hval,
                            const double *g, const double *lb, const double *ub,
                            const double *lbA, const double *ubA, const double *x,
                            const double *y, const int *W_b, const int *W_c,
                            orc_optimality_status *out);

/* ---------------------------------------------------------------------- */
/* active-set QP solver (stand-in for qpOASES 3.2.1 SQProblem)             */
/* ---------------------------------------------------------------------- */
typedef struct orc_qp orc_qp;

enum { ORC_QPS_NOTINITIALISED = 0, ORC_QPS_PREPARINGAUXILIARYQP = 1, ORC_QPS_AUXILIARYQPSOLVED = 2,
       ORC_QPS_PERFORMINGHOMOTOPY = 3, ORC_QPS_HOMOTOPYQPSOLVED = 4, ORC_QPS_SOLVED = 5 };

enum { ORC_RET_OK = 0, ORC_RET_MAX_NWSR = 1, ORC_RET_INFEASIBLE = 2, ORC_RET_UNBOUNDED = 3,
       ORC_RET_SETUP_FAILED = 4 };

orc_qp *orc_qp_create(int nV, int nC);
void orc_qp_destroy(orc_qp *qp);
/* matrices are copied; H may be NULL (LP). H is the FULL symmetric matrix in CSC. */
int orc_qp_set_A_csc(orc_qp *qp, const int *jc, const int *ir, const double *val);
int orc_qp_set_H_csc(orc_qp *qp, const int *jc, const int *ir, const double *val);

/* SQProblem::init(H,g,A,lb,ub,lbA,ubA,nWSR [,xOpt,yOpt,guessedBounds]) -- call sites
 * src/qpOASESInterface.cpp:155, 204-206, 700-702, 726-728, 747-749.
 * x0/y0/guess_b may be NULL (cold start when all three are NULL). */
int orc_qp_init(orc_qp *qp, const double *g, const double *lb, const double *ub,
                const double *lbA, const double *ubA, int *nWSR, const double *x0,
                const double *y0, const int *guess_b);
/* the same cold init `reps` times in a C loop (timing aid of bench.py's cpu_baseline) */
int orc_qp_init_repeat(orc_qp *qp, const double *g, const double *lb, const double *ub, const double *lbA,
                       const double *ubA, int nWSR_max, int reps);
/* SQProblem::hotstart(g,lb,ub,lbA,ubA,nWSR) -- src/qpOASESInterface.cpp:180, 191 */
int orc_qp_hotstart(orc_qp *qp, const double *g, const double *lb, const double *ub,
                    const double *lbA, const double *ubA, int *nWSR);
/* SQProblem::hotstart(H,g,A,lb,ub,lbA,ubA,nWSR) -- src/qpOASESInterface.cpp:184, 197.
 * The matrices must have been replaced with orc_qp_set_{A,H}_csc beforehand. */
int orc_qp_hotstart_matrices(orc_qp *qp, const double *g, const double *lb, const double *ub,
                             const double *lbA, const double *ubA, int *nWSR);

/* H := H + reg*I for every later solve (qpOASES regularises an all-zero Hessian, i.e. the LP of
 * optimizeLP, src/qpOASESInterface.cpp:227-284); the objective excludes the reg term */
/* bench.py only: `iters` x (hot start on vector set A / B in turn + working-set mapping + KKT certificate) in C */
int orc_qp_solveqp_repeat(orc_qp *qp, const double *const *vecA, const double *const *vecB, int nWSR_max, int iters);
void orc_qp_set_regularisation(orc_qp *qp, double reg);
/* init(.., x0, y0, guessedBounds) without guessed constraints (the FIXED <-> VARIED flip of
 * src/qpOASESInterface.cpp:199-207). 0 (default) = the reference's path: qpOASES derives no constraint
 * from y0 when x0 is given as well, the working set restarts from the constraints x0 happens to sit on.
 * 1 = what the HIP engine does: constraint sides from the signs of y0 (the working set of the previous
 * solve). Same KKT point on a strictly convex QP, far fewer working-set changes. */
void orc_qp_set_guess_constraints_from_y0(orc_qp *qp, int on);
void orc_qp_get_primal(const orc_qp *qp, double *x);
void orc_qp_get_dual(const orc_qp *qp, double *y);       /* nV bound mult., then nC */
double orc_qp_get_objective(const orc_qp *qp);
void orc_qp_get_working_set_bounds(const orc_qp *qp, int *ws_b);
void orc_qp_get_working_set_constraints(const orc_qp *qp, int *ws_c);
int orc_qp_status(const orc_qp *qp);
int orc_qp_is_solved(const orc_qp *qp);
int orc_qp_is_infeasible(const orc_qp *qp);
int orc_qp_is_unbounded(const orc_qp *qp);
int orc_qp_nflips(const orc_qp *qp);

/* qpOASESInterface::get_status src/qpOASESInterface.cpp:332-357 -> Exitflag value */
int orc_exitflag(const orc_qp *qp);

/* bench.py / tests only (traj_oracle.c): the QP side of every iteration of an hs071 SQP run in one C loop -- setupQP
 * (src/Algorithm.cpp:645-697), solveQP (src/QPhandler.cpp:470-499), getters; traj = nit rows {delta, rho, x[4], lam[2]} */
int orc_hs071_trajectory_replay(int nit, const double *traj, int reps, double *out_us, double *x_last, double *y_last,
                                int *qp_iter, int *modes);

#ifdef __cplusplus
}
#endif
#endif
