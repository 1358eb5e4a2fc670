"""Dense 2048 x 4096 cold solve on the HBM-resident engine, twice in one process (the first solve pays the code-object
load); RSQP_LIB selects the library, so two builds can be compared inside one gpurun call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from restartsqp_amd import capi, problems
q = problems.dense_qp()
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    s = capi.Solver(q.nV, q.nC)
    s.set_options(qp_maxiter=400000)
    s.set_A_csc(q.A_jc, q.A_ir, q.A_val); s.set_H_csc(q.H_jc, q.H_ir, q.H_val)
    for w, v in zip(range(5), (q.g, q.lb, q.ub, q.lbA, q.ubA)):
        s.set_vector(w, v)
    c0 = time.process_time(); t = time.perf_counter(); n = s.solve(capi.MODE_COLD, 200000); t = time.perf_counter() - t; c0 = time.process_time() - c0
    ok, st, _, _ = s.test_optimality()
    print("%s rep %d: %.3f s (process CPU %.3f s), nWSR %d, %.1f us per change, KKT %.2e certified %d" % (os.environ.get("RSQP_LIB", "default"), rep, t, c0, n, 1e6 * t / max(n, 1), st.KKT_error, ok), flush=True)
    s.close()
