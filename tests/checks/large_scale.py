"""HBM-resident engine at growing sizes: time, iterations, certificate; oracle comparison while cheap."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O
from restartsqp_amd import capi, problems

def run(q, nWSR):
    s = capi.Solver(q.nV, q.nC)
    s.set_A_csc(q.A_jc, q.A_ir, q.A_val); s.set_H_csc(q.H_jc, q.H_ir, q.H_val)
    for w, v in zip(range(5), (q.g, q.lb, q.ub, q.lbA, q.ubA)):
        s.set_vector(w, v)
    t = time.time(); n = s.solve(capi.MODE_COLD, nWSR); t = time.time() - t
    return s, n, t

sizes = [(300, 600), (600, 1200), (1024, 2048)]
if len(sys.argv) > 1:
    sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for n, m in sizes:
    q = problems.dense_qp(n, m, seed=20260101)
    s, it, t = run(q, 100000)
    ok, st, _, _ = s.test_optimality()
    wb, wc = s.working_set_raw()
    print("dense %dx%d engine %d: nWSR %d in %.2f s (%.3f ms/iter) status %d KKT %.2e ok %s nFX %d nAC %d" % (
        n, m, s.engine, it, t, 1e3 * t / max(it, 1), s.status, st.KKT_error, ok, int((wb != 0).sum()), int((wc != 0).sum())), flush=True)
    if n <= 700:
        qp = O.OracleQP(q.nV, q.nC); qp.set_A_csc(q.A_jc, q.A_ir, q.A_val); qp.set_H_csc(q.H_jc, q.H_ir, q.H_val)
        t0 = time.time(); rc, n2 = qp.init(q.g, q.lb, q.ub, q.lbA, q.ubA, 100000); t0 = time.time() - t0
        print("   oracle: nWSR %d in %.2f s; same ws %s; dx %.1e" % (n2, t0, np.array_equal(wb, qp.ws_bounds) and np.array_equal(wc, qp.ws_constraints), np.abs(s.x - qp.x).max()), flush=True)
