"""BASELINE configs 3 and 4 on the HBM-resident engine: cold solve + certificate; sparse: a few hot starts."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from restartsqp_amd import capi, problems

def load(q):
    s = capi.Solver(q.nV, q.nC)
    s.set_A_csc(q.A_jc, q.A_ir, q.A_val); s.set_H_csc(q.H_jc, q.H_ir, q.H_val)
    for w, v in zip(range(5), (q.g, q.lb, q.ub, q.lbA, q.ubA)):
        s.set_vector(w, v)
    return s

which = sys.argv[1] if len(sys.argv) > 1 else "dense"
if which == "dense":
    q = problems.dense_qp()
else:
    q = problems.sparse_qp(box=float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)
t = time.time(); s = load(q); print("setup %.2f s" % (time.time() - t), flush=True)
t = time.time(); n = s.solve(capi.MODE_COLD, 200000); t = time.time() - t
ok, st, _, _ = s.test_optimality()
wb, wc = s.working_set_raw()
print("%s %dx%d: cold nWSR %d in %.2f s (%.3f ms/iter) status %d KKT %.2e ok %s nFX %d nAC %d" % (
    q.name, q.nV, q.nC, n, t, 1e3 * t / max(n, 1), s.status, st.KKT_error, ok, int((wb != 0).sum()), int((wc != 0).sum())), flush=True)
if which != "dense":
    for k, (qk, changed) in enumerate(problems.sparse_sequence(q, nsteps=6)):
        if changed:
            continue   # HOT_MATRICES re-factorises from scratch: timed separately
        for w, v in zip(range(5), (qk.g, qk.lb, qk.ub, qk.lbA, qk.ubA)):
            s.set_vector(w, v)
        t = time.time(); n = s.solve(capi.MODE_HOT_VECTORS, 200000); t = time.time() - t
        ok, st, _, _ = s.test_optimality()
        print("  hot step %d: nWSR %d in %.3f s status %d KKT %.2e" % (k, n, t, s.status, st.KKT_error), flush=True)
if which != "dense" and len(sys.argv) > 3:
    qk, changed = None, False
    for qk, changed in problems.sparse_sequence(q, nsteps=2):
        pass
    s.set_A_csc(qk.A_jc, qk.A_ir, qk.A_val)
    for w, v in zip(range(5), (qk.g, qk.lb, qk.ub, qk.lbA, qk.ubA)):
        s.set_vector(w, v)
    t = time.time(); n = s.solve(capi.MODE_HOT_MATRICES, 200000); t = time.time() - t
    ok, st, _, _ = s.test_optimality()
    print("  hot MATRICES step: nWSR %d in %.3f s status %d KKT %.2e" % (n, t, s.status, st.KKT_error), flush=True)
