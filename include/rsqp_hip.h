/*
 * rsqp_hip.h -- C ABI of the MI355X-native QP-subproblem engine (librsqp_hip.so).
 *
 * Drop-in boundary for lanl-ansi/RestartSQP: these are the entry points a
 * `QPSolverInterface` subclass (include/sqphot/QPsolverInterface.hpp:43-194) binds in place
 * of its qpOASES calls. Plain pointers and sizes only; every pointer is a HOST pointer
 * unless the name says `_dev`. All floating point data is fp64, indices are 32-bit int.
 *
 * QP convention (QPsolverInterface.hpp:37-41):
 *      min 1/2 x'Hx + g'x   s.t.  lbA <= Ax <= ubA,  lb <= x <= ub
 * Multipliers: y[0..nV) for the bounds, y[nV..nV+nC) for the constraints
 * (qpOASESInterface.cpp:290-305); y >= 0 at a lower side, y <= 0 at an upper side.
 *
 * Return value of every int function: RSQP_OK (0) or a negative RSQP_ERR_* code; solver
 * outcomes (infeasible, iteration limit ...) are NOT errors -- read rsqp_get_status().
 * The host adapter turns them into the reference's exceptions.
 */
#ifndef RSQP_HIP_H
#define RSQP_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define RSQP_OK 0
#define RSQP_ERR_ARG (-1)       /* bad argument / call order                         */
#define RSQP_ERR_DEVICE (-2)    /* HIP runtime failure (no GPU, launch error, OOM)    */
#define RSQP_ERR_TOO_LARGE (-3) /* problem does not fit the selected engine           */
#define RSQP_ERR_WORKING_SET (-4) /* INVALID_WORKING_SET (qpOASESInterface.cpp:552,866) */

/* include/sqphot/Types.hpp:51-73 (subset produced on this path) */
#define RSQP_QP_OPTIMAL 20
#define RSQP_QPERROR_INTERNAL_ERROR 21
#define RSQP_QPERROR_INFEASIBLE 22
#define RSQP_QPERROR_UNBOUNDED 23
#define RSQP_QPERROR_EXCEED_MAX_ITER 24
#define RSQP_QPERROR_NOTINITIALISED 25
#define RSQP_QPERROR_PREPARINGAUXILIARYQP 26
#define RSQP_QPERROR_AUXILIARYQPSOLVED 27
#define RSQP_QPERROR_PERFORMINGHOMOTOPY 28
#define RSQP_QPERROR_HOMOTOPYQPSOLVED 29
#define RSQP_QPERROR_UNKNOWN 30

/* include/sqphot/Types.hpp:84-89 */
#define RSQP_ACTIVE_ABOVE 1
#define RSQP_ACTIVE_BELOW (-1)
#define RSQP_ACTIVE_BOTH_SIDE (-99)
#define RSQP_INACTIVE 0

/* which vector (setters of QPsolverInterface.hpp:144-173) */
enum { RSQP_VEC_G = 0, RSQP_VEC_LB = 1, RSQP_VEC_UB = 2, RSQP_VEC_LBA = 3, RSQP_VEC_UBA = 4 };

/* how a solve starts -- the four qpOASES call shapes of qpOASESInterface.cpp:137-224 */
enum {
    RSQP_MODE_COLD = 0,         /* init(H,g,A,lb,ub,lbA,ubA,nWSR)                :155       */
    RSQP_MODE_HOT_VECTORS = 1,  /* hotstart(g,lb,ub,lbA,ubA,nWSR)                :180,191   */
    RSQP_MODE_HOT_MATRICES = 2, /* hotstart(H,g,A,lb,ub,lbA,ubA,nWSR)            :184,197   */
    RSQP_MODE_WARM_REINIT = 3   /* init(..., nWSR,0,x_qp,y_qp,&bounds)           :204-206   */
};

/* include/sqphot/Types.hpp:107-119 (the numeric part) */
typedef struct {
    double primal_violation, dual_violation, compl_violation, stationarity_violation, KKT_error;
} rsqp_optimality_status;

typedef struct rsqp_solver rsqp_solver;
typedef struct rsqp_batch rsqp_batch;

/* ------------------------------------------------------------------------------------ */
/* library                                                                               */
/* ------------------------------------------------------------------------------------ */
const char *rsqp_version(void);
/* sha256 of the sources, headers and flags this library was built from (restartsqp_amd/build.py rebuilds on a mismatch) */
const char *rsqp_build_hash(void);
/* number of visible HIP devices (0 when there is none); never initialises a context */
int rsqp_device_count(void);
const char *rsqp_last_error(void);

/* ------------------------------------------------------------------------------------ */
/* one QP: replaces qpOASESInterface (+ the SQProblem object it owns)                    */
/* ------------------------------------------------------------------------------------ */
/* qpOASESInterface ctor + allocate_memory (qpOASESInterface.cpp:35-50, 106-128) and the
 * plain-QP ctor (:54-94). device < 0 selects the current device. */
int rsqp_create(int nV, int nC, int device, rsqp_solver **out);
void rsqp_destroy(rsqp_solver *s);
/* measurement only (bench.py): per-kernel-class accounting of the HBM-resident engine. Profiling brackets
 * every launch of a class with HIP events (serialises the stream). rsqp_get_engine_profile returns the number
 * of classes n (0: the handle has no HBM engine) and fills out4n[4k..4k+3] = {calls, ms, algorithmic bytes, 0}. */
int rsqp_set_engine_profiling(rsqp_solver *s, int on);
int rsqp_get_engine_profile(const rsqp_solver *s, double *out4n, int n);
int rsqp_engine_profile_names(const char **names, int n);
/* measurement only: the last blocked (matrix-core) set-up of a non-empty working set on the HBM-resident engine --
 * what SQProblem::hotstart(H, g, A, ..) / init(.., x0, y0, guessedBounds) re-factorise (call sites
 * qpOASESInterface.cpp:184,197,204-206). out8 = {nFR, nAC, nZ, ms of QR + explicit Q + R^-1, ms of Z'HZ + Cholesky +
 * inverse (HIP events on the engine's stream), algorithmic flops of the first part, of the second, 0}.
 * Returns 1, or 0 when no blocked set-up has run on this handle. */
int rsqp_get_setup_profile(const rsqp_solver *s, double *out8);
/* seconds of the one-off structure analysis behind the first set_A / set_H (SpHbMat::setStructure, SpHbMat.cpp:196-355:
 * sort, CSC + CSR copy + SpMV plan, upload), timed apart from the per-solve cost; which: 0 = A, 1 = H;
 * < 0 when the matrix has not been set. */
double rsqp_get_structure_seconds(const rsqp_solver *s, int which);
/* RSQP_MODE_* the dispatch of the last rsqp_optimize_qp / _lp (or the caller of rsqp_solve) used: which of the four call
 * shapes of qpOASESInterface.cpp:155,180-206 ran; -1 before the first solve */
int rsqp_get_last_mode(const rsqp_solver *s);
/* which formulation of the HBM-resident engine holds the factors of this handle (the stand-in for qpOASES' TQ / Cholesky factors
 * behind SQProblem::init / hotstart, call sites qpOASESInterface.cpp:155,180-206): 0 = null-space (any Hessian), 1 = range-space,
 * diagonal positive Hessian, 2 = general range-space with a banded H^-1 operator, 3 = the same with the dense inverse, 4 = 3 plus
 * the static tableau [I; A] H^-1 [I A'] of small dense problems (nV + nC <= 8192);
 * -1: the handle has no HBM-resident engine (or has not solved yet) */
int rsqp_get_large_path(const rsqp_solver *s);
int rsqp_get_nV(const rsqp_solver *s);
int rsqp_get_nC(const rsqp_solver *s);
/* engine selection: 0 = automatic (the LDS-resident kernel when the problem image fits the
 * 160 KiB of one CU, the HBM-resident engine otherwise), 1 = LDS-resident, 2 = HBM-resident.
 * Both run entirely on the GPU; there is no CPU path. Call before the first solve. */
int rsqp_set_engine(rsqp_solver *s, int engine);
int rsqp_get_engine(const rsqp_solver *s);
/* The FIXED <-> VARIED flip of optimizeQP re-initialises with init(.., x_qp, y_qp, &bounds) and NO guessed
 * constraints (qpOASESInterface.cpp:199-207); qpOASES then restarts from the constraints A x_qp happens to
 * sit on -- with perturbed data: none -- and re-adds the active set one change at a time (7 292 changes on the
 * sparse 10k x 20k sequence). from_y0 == 0 (DEFAULT): exactly that path (what the CPU oracle restates).
 * from_y0 != 0 (opt-in shortcut, NOT the reference's behaviour): the constraint sides are taken from the signs
 * of y_qp, i.e. the working set of the previous solve; same KKT point on a strictly convex QP, ~20x fewer
 * changes, but nWSR and -- where the solution is not unique -- the point may differ from the reference's. */
int rsqp_set_reinit_guess(rsqp_solver *s, int from_y0);
/* Options fields the adapter reads: qp_maxiter, lp_maxiter (Options.cpp:45,54) */
int rsqp_set_options(rsqp_solver *s, int qp_maxiter, int lp_maxiter);

/* set_A (qpOASESInterface.cpp:426-442): first call = SpHbMat::setStructure(rhs, I_info)
 * (SpHbMat.cpp:196-268) -- 1-based COO + identity blocks -> CSC on the device; later
 * calls = SpHbMat::setMatVal (:368-380), a device scatter through `order`. */
int rsqp_set_A_triplet(rsqp_solver *s, int nnz, const int *irow, const int *jcol, const double *val,
                       int n_ident, const int *id_irow, const int *id_jcol, const int *id_size,
                       const double *id_value);
/* set_H (qpOASESInterface.cpp:400-423): SpHbMat::setStructure(rhs) (:284-355, symmetric
 * triangle mirrored) on the first call, setMatVal (:383-393) afterwards. */
int rsqp_set_H_triplet(rsqp_solver *s, int nnz, const int *irow, const int *jcol, const double *val,
                       int is_symmetric);
/* plain-QP ctor path (qpOASESInterface.cpp:54-94): matrices arrive as CSC (0-based).
 * H is the full symmetric matrix. A second call with the same pattern refreshes values. */
int rsqp_set_A_csc(rsqp_solver *s, const int *jc, const int *ir, const double *val);
int rsqp_set_H_csc(rsqp_solver *s, const int *jc, const int *ir, const double *val);
/* read the device CSC back (getA()/getH() of QPsolverInterface.hpp:47-59); any output
 * pointer may be NULL. `order` is SpHbMat::order_. */
int rsqp_get_A_nnz(const rsqp_solver *s);
int rsqp_get_H_nnz(const rsqp_solver *s);
int rsqp_get_A_csc(const rsqp_solver *s, int *jc, int *ir, double *val, int *order);
int rsqp_get_H_csc(const rsqp_solver *s, int *jc, int *ir, double *val, int *order);

/* vector setters (qpOASESInterface.cpp:361-395, 445-484). Scalar setters are staged on
 * the host and flushed to the device by the next solve / product / certificate call. */
int rsqp_set_vector(rsqp_solver *s, int which, const double *v);
int rsqp_set_entry(rsqp_solver *s, int which, int location, double value);
int rsqp_get_vector(const rsqp_solver *s, int which, double *v);
/* device time (ms per launch, HIP events) of the two kernels behind a value refresh of A --
 * SpHbMat::setMatVal (SpHbMat.cpp:368-380): the scatter through `order` (20 B per entry) and the
 * refresh of the CSR copy -- on the values staged by the last rsqp_set_A_triplet */
int rsqp_time_value_refresh(rsqp_solver *s, int repeats, float *ms_scatter, float *ms_gather);
/* the product path since round 4: ONE launch that writes every refreshed triplet value to its CSC slot and to its slot of
 * the CSR copy (set_A on a known pattern, SpHbMat.cpp:368-380); average ms per launch */
int rsqp_time_value_refresh_fused(rsqp_solver *s, int repeats, float *ms);
/* tuning aid for the HBM-resident engine (no reference counterpart): device ms per call of one of its streaming
 * kernel classes -- kind 0: y = M w (column-major), 1: y = M'x, 2: rank-1 update -- on an nrows x ncols block of an
 * n x n buffer (leading dimension n), exactly as the engine launches them */
int rsqp_time_large_kernel(int device, int n, int kind, int nrows, int ncols, int repeats, float *ms);
/* reset_constraints (qpOASESInterface.cpp:897-902) */
int rsqp_reset_constraints(rsqp_solver *s);

/* optimizeQP (qpOASESInterface.cpp:137-224) including the FIXED/VARIED dispatch of
 * get_Matrix_change_status (:817-833), reset_flags (:488-496) and handle_error
 * (:686-758). *nWSR_used receives what the adapter adds to Stats::qp_iter. */
int rsqp_optimize_qp(rsqp_solver *s, int *nWSR_used);
/* optimizeLP (qpOASESInterface.cpp:227-284): the same engine with H = 0 and the lp_maxiter
 * budget. As qpOASES does for an all-zero Hessian, the LP is solved as the QP with
 * H = regVal*I, regVal = |g|_2 * 1e3*EPS, followed by one regularisation (proximal) step.
 * Any H set on this handle is ignored. handle_error's LP branch (:688-717) included. */
int rsqp_optimize_lp(rsqp_solver *s, int *nWSR_used);
/* low-level: one SQProblem::init / hotstart call. nWSR: in = limit, out = used.
 * x0, y0, guess_b (qpOASES convention -1/0/+1) may be NULL. */
int rsqp_solve(rsqp_solver *s, int mode, int *nWSR, const double *x0, const double *y0,
               const int *guess_b);

/* getters (qpOASESInterface.cpp:290-357) */
int rsqp_get_primal(const rsqp_solver *s, double *x);            /* nV            */
int rsqp_get_dual(const rsqp_solver *s, double *y);              /* nV + nC       */
double rsqp_get_objective(const rsqp_solver *s);
int rsqp_get_status(const rsqp_solver *s);                       /* Exitflag      */
int rsqp_is_solved(const rsqp_solver *s);
/* solver convention -1 lower / 0 / +1 upper (getWorkingSetBounds/Constraints) */
int rsqp_get_working_set_raw(const rsqp_solver *s, int *ws_b, int *ws_c);
/* get_working_set (qpOASESInterface.cpp:835-895): ActiveType per bound / constraint,
 * computed on the device (A*x product + mapping), reference quirks preserved. */
int rsqp_get_working_set(rsqp_solver *s, int *W_c, int *W_b);
/* test_optimality (qpOASESInterface.cpp:498-684): fused device certificate. Returns 1
 * (KKT_error <= 1e-6), 0, or RSQP_ERR_WORKING_SET. W_c / W_b may be NULL. */
int rsqp_test_optimality(rsqp_solver *s, int *W_c, int *W_b, rsqp_optimality_status *out);

/* SpHbMat::times / transposed_times on the device copy (SpHbMat.cpp:659-737) */
int rsqp_A_times(rsqp_solver *s, const double *p, double *result);            /* nC <- nV */
int rsqp_A_transposed_times(rsqp_solver *s, const double *p, double *result); /* nV <- nC */
int rsqp_H_times(rsqp_solver *s, const double *p, double *result);

/* ------------------------------------------------------------------------------------ */
/* on-disk QP formats of the reference (host only, no GPU needed)                         */
/* ------------------------------------------------------------------------------------ */
/* WriteQPDataToFile (QPsolverInterface.hpp:182-184; called for every failed QP, QPhandler.cpp:569-571):
 *   RSQP_DUMP_QPOASES  qpOASESInterface.cpp:791-814 + SpHbMat.cpp:568-578:
 *                      lb[nV] lbA[nC] ub[nV] ubA[nC] g[nV]; A: ir[nnz] jc[nV+1] val[nnz]; H likewise
 *   RSQP_DUMP_QORE     QOREInterface.cpp:582-598 + SpHbMat.cpp:556-567 (the layout of test/unsolved_QP_data):
 *                      nV nC nnzA nnzH; lb[nV+nC] ub[nV+nC] g[nV]; A, H as CSR: rowptr, col, val
 * one number per line, doubles "%23.16e" (Vector.cpp:212-215). Matrices are given / kept as CSC. */
enum { RSQP_DUMP_QPOASES = 0, RSQP_DUMP_QORE = 1 };
int rsqp_write_qp_dump(const char *path, int layout, int nV, int nC, const double *lb, const double *ub,
                       const double *lbA, const double *ubA, const double *g, const int *A_jc, const int *A_ir,
                       const double *A_val, const int *H_jc, const int *H_ir, const double *H_val);
/* the same for the data currently held by a solver handle (matrix values are read back from the device) */
int rsqp_write_qp_data(const rsqp_solver *s, const char *path, int layout);
/* reader of the QORE layout (test/QPsolvers_testers.cpp:48-150); matrices come back as CSC
 * (convert_csr_to_csc, :18-29). First call the _sizes function, then pass arrays of
 * lb, ub, g: nV; lbA, ubA: nC; A_jc, H_jc: nV+1; A_ir, A_val: nnzA; H_ir, H_val: nnzH. */
int rsqp_read_qore_dump_sizes(const char *path, int *nV, int *nC, int *nnzA, int *nnzH);
int rsqp_read_qore_dump(const char *path, double *lb, double *ub, double *lbA, double *ubA, double *g,
                        int *A_jc, int *A_ir, double *A_val, int *H_jc, int *H_ir, double *H_val);

/* ------------------------------------------------------------------------------------ */
/* a batch of independent QPs (north_star: CUTEst sweeps / parameter scans)              */
/* ------------------------------------------------------------------------------------ */
/* nq problems of individual size; nV[q], nC[q], and per problem CSC matrices given as
 * concatenated arrays with offsets: Ajc_off[q] indexes into Ajc (length sum(nV+1)),
 * Annz_off[q] into Air/Aval; likewise H. Everything is copied to the device once. */
int rsqp_batch_create(int nq, const int *nV, const int *nC, const int *Ajc, const int *Air,
                      const double *Aval, const int *Hjc, const int *Hir, const double *Hval,
                      int device, rsqp_batch **out);
void rsqp_batch_destroy(rsqp_batch *b);
/* vectors concatenated over the batch: g, lb, ub have sum(nV) entries; lbA, ubA sum(nC) */
int rsqp_batch_set_vectors(rsqp_batch *b, const double *g, const double *lb, const double *ub,
                           const double *lbA, const double *ubA);
/* refresh the matrix values (same patterns) */
int rsqp_batch_set_matrix_values(rsqp_batch *b, const double *Aval, const double *Hval);
/* solve all problems with one launch; data already resident on the device.
 * mode as above (WARM_REINIT not available for batches). Asynchronous on the batch's
 * stream; rsqp_batch_sync() waits. */
int rsqp_batch_solve(rsqp_batch *b, int mode, int max_nWSR);
int rsqp_batch_sync(rsqp_batch *b);
/* keep != 0 (default): every solve writes the state a hot start needs (factors, iterate, multipliers,
 * working set: what a qpOASES SQProblem object keeps between init / hotstart calls) back to HBM.
 * keep == 0: batches that are only ever solved from a cold start (parameter scans) skip that write
 * (1.7 KB per hs071-scale QP); a hot start after such a solve silently becomes a cold start. */
int rsqp_batch_set_keep_state(rsqp_batch *b, int keep);
/* which kernel the last rsqp_batch_solve launched (diagnostics; all of them stand in for the SQProblem::init / hotstart calls of
 * qpOASESInterface.cpp:155,180-206): 0 = the LDS-resident null-space kernels (+ the mid-size tableau kernel), 1 = the hs071-scale
 * tableau kernel with 8 lanes per problem, 2 = the lane-per-problem kernel (cold starts of one-shape batches of at most 8 x 2
 * with more than 16 384 members; RSQP_LANE); -1 before the first solve */
int rsqp_batch_get_last_kernel(const rsqp_batch *b);
/* device time of the last rsqp_batch_solve in milliseconds (HIP events on its stream) */
float rsqp_batch_last_solve_ms(rsqp_batch *b);
/* HIP-event stopwatch on the batch's stream: start records an event, stop records a
 * second one, waits for it and returns the elapsed device time in ms (covers every
 * launch enqueued in between -- what bench.py divides by the step count). While the stopwatch
 * runs, rsqp_batch_solve does not record its own per-launch events (rsqp_batch_last_solve_ms keeps
 * the value of the last solve outside a stopwatch interval). */
int rsqp_batch_timer_start(rsqp_batch *b);
float rsqp_batch_timer_stop_ms(rsqp_batch *b);
/* results, concatenated like the inputs; any pointer may be NULL */
int rsqp_batch_get_results(rsqp_batch *b, double *x, double *y, int *ws_b, int *ws_c, int *status,
                           int *nWSR, double *obj);
/* fused KKT certificate for every problem of the batch (one launch) */
int rsqp_batch_test_optimality(rsqp_batch *b, rsqp_optimality_status *out /* nq */, int *ok /* nq */);
/* fixed-stride result records written to DEVICE memory, for the gather of a sharded batch over RCCL
 * (the path has no other exchange). Per problem, stride = 4 + 3 nVmax + 2 nCmax doubles:
 *   {Exitflag, nWSR, objective, KKT_error (0 before rsqp_batch_test_optimality), x[nVmax],
 *    y_bounds[nVmax], y_constraints[nCmax], ws_b[nVmax], ws_c[nCmax]}   (qpOASESInterface.cpp:290-357).
 * rec_dev: caller-owned device buffer of nq * stride doubles; enqueued on the batch's stream. */
int rsqp_batch_record_stride(const rsqp_batch *b);
int rsqp_batch_pack_records_dev(rsqp_batch *b, double *rec_dev);
/* the same records packed on the device and copied to a host buffer (gathers that run over host memory) */
int rsqp_batch_pack_records_host(rsqp_batch *b, double *rec_host);
/* Sharding of a batch of independent QPs over the ranks of a multi-GPU job, for C++ hosts that drive one
 * process per GPU themselves: every rank calls these with the same arguments, builds an rsqp_batch from ITS members on
 * its device, solves, packs the fixed-stride records (rsqp_batch_pack_records_dev) and all-gathers them with its own RCCL
 * communicator (ncclAllGather of count * stride doubles; ranks with one member less pad). Host-only, no device needed.
 *   rsqp_shard_range:     contiguous block [*lo, *hi) of rank `rank` of `world`; sizes differ by at most one (512 QPs on 8
 *                         GPUs: 64 each).
 *   rsqp_balanced_shard:  heterogeneous sizes -- members sorted by nV * max(nC, 1), largest first, and dealt snake-wise
 *                         (0..W-1, W-1..0, ..), so that every rank gets its share of the expensive members; idx receives the
 *                         member indices of rank `rank` (largest first), *count their number (<= ceil(nq / world)).
 * Both are the partitions restartsqp_amd/parallel.py uses (shard_range, balanced_shards). */
int rsqp_shard_range(int nq, int rank, int world, int *lo, int *hi);
int rsqp_balanced_shard(int nq, const int *nV, const int *nC, int rank, int world, int *idx, int *count);
/* The two exchange steps of the sharded batch as NATIVE RCCL calls (the reference itself has no collective:
 * it is single-threaded, test/simple_test.cpp:72). librccl.so is bound at first use, so hosts that never shard need none.
 *   rsqp_rccl_unique_id / rsqp_rccl_comm_create / _destroy: ncclGetUniqueId on one rank (the host ships the 128 bytes to
 *       the others by whatever it has: MPI, a file, torch.distributed), ncclCommInitRank on every rank with its device.
 *       A host that already owns an ncclComm_t passes it instead -- `comm` is an ncclComm_t everywhere below.
 *   rsqp_rccl_broadcast_dev: ncclBroadcast of `bytes` bytes of device memory from rank `root` (shared structure of a
 *       parameter scan: patterns, values, vectors) on `hip_stream` (NULL = the null stream); returns when it has completed.
 *   rsqp_batch_allgather_records: packs this rank's fixed-stride records (rsqp_batch_pack_records_dev) straight into its
 *       slot of all_dev and all-gathers in place over RCCL / xGMI. all_dev: DEVICE buffer of world * count_per_rank * stride
 *       doubles; count_per_rank >= the member count of every rank, the same on all ranks (ranks with fewer members pad
 *       with zero records: Exitflag 0). Enqueued on the batch's stream behind the solve; returns when it has completed. */
#define RSQP_RCCL_UNIQUE_ID_BYTES 128
int rsqp_rccl_unique_id(char id[RSQP_RCCL_UNIQUE_ID_BYTES]);
int rsqp_rccl_comm_create(const char id[RSQP_RCCL_UNIQUE_ID_BYTES], int rank, int world, int device, void **comm);
int rsqp_rccl_comm_destroy(void *comm);
int rsqp_rccl_broadcast_dev(void *comm, void *buf_dev, long long bytes, int root, void *hip_stream);
int rsqp_batch_allgather_records(rsqp_batch *b, void *comm, int count_per_rank, double *all_dev);

/* ------------------------------------------------------------------------------------ */
/* batched sparse products, device resident -- the SpMV the roofline target names        */
/* ------------------------------------------------------------------------------------ */
/* y_k = A_k x_k (transposed == 0) or x_k = A_k' y_k for nbatch matrices that share one
 * CSC pattern (jc, ir) and have their own values / vectors, all in device memory. */
typedef struct rsqp_spmv_plan rsqp_spmv_plan;
int rsqp_spmv_plan_create(int nrow, int ncol, const int *jc, const int *ir, int nbatch, int device,
                          rsqp_spmv_plan **out);
void rsqp_spmv_plan_destroy(rsqp_spmv_plan *p);
/* host -> device staging of values (nbatch*nnz) and input vectors */
int rsqp_spmv_plan_upload(rsqp_spmv_plan *p, const double *vals, const double *xin, int transposed);
int rsqp_spmv_plan_run(rsqp_spmv_plan *p, int transposed, int repeats, float *ms_per_launch);
int rsqp_spmv_plan_download(rsqp_spmv_plan *p, double *out, int transposed);
/* which kernel rsqp_spmv_plan_run launches for this product: 0 = csx_stream_spmv (entry-order sums,
 * bit-exact vs SpHbMat.cpp:659-737), 35 = csx_ldsvec_spmv_pipe2<4,3>, 38 = csx_ldsvec_spmv_pipe2<2,4>
 * (other codes: tuning variants of sparse.hip). *idx16 (may be NULL) = 1 when the 16-bit index copies
 * (instantiation <.., unsigned short>) are used, 0 for <.., int>. */
int rsqp_spmv_plan_variant(const rsqp_spmv_plan *p, int transposed, int *idx16);

/* ------------------------------------------------------------------------------------ */
/* dense f64 building blocks of the HBM-resident engine (MFMA GEMM, blocked QR, Cholesky) */
/* ------------------------------------------------------------------------------------ */
/* They rebuild the TQ factorisation and the projected Hessian when a solve starts from a
 * non-empty working set -- qpOASES setupTQfactorisation / computeProjectedCholesky inside
 * SQProblem::hotstart(H, g, A, ...) and init(..., xOpt, yOpt, guessedBounds), reference call
 * sites src/qpOASESInterface.cpp:184,197,204-206. Host-pointer entries (column-major), for
 * verification and benchmarking; the engine calls the device versions directly.
 * C = alpha op(A) op(B) + beta C; *ms (may be NULL) = device time of `repeats` launches / repeats */
int rsqp_dense_gemm(int transA, int transB, int m, int n, int k, double alpha, const double *A, int lda,
                    const double *B, int ldb, double beta, double *C, int ldc, int repeats, float *ms);
/* Householder QR of B (m x n, m >= n): Q (m x m) and Rinv = R^-1 (n x n) are returned, B is
 * overwritten by R / the reflectors; *ndep = columns found linearly dependent (remaining norm
 * <= eps_li * original norm), the outputs are only valid when it is 0 */
int rsqp_dense_qr(int m, int n, double *B, double *Q, double *Rinv, double eps_li, int *ndep, float *ms);
/* Cholesky G = U'U (upper) of a symmetric n x n matrix, then Ginv = G^-1; *not_pd != 0 when a
 * pivot fails d > pd_rel (|g_jj| + sum) + pd_abs */
int rsqp_dense_chol_inverse(int n, double *G, double *Ginv, double pd_rel, double pd_abs, int *not_pd, float *ms);

#ifdef __cplusplus
}
#endif
#endif
