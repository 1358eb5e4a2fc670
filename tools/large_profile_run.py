"""Workload for `rocprofv3 --kernel-trace --stats`: the HBM-resident engine on BASELINE configs 3 and 4.
    python3 tools/large_profile_run.py dense            dense 2048 x 4096, cold start
    python3 tools/large_profile_run.py sparse [steps]   sparse 10k x 20k: cold start + `steps` QPs of the warm-started sequence
    python3 tools/large_profile_run.py band5 [steps]    the same with the 5-band Hessian (problems.sparse_qp(band=5))
Prints wall time, nWSR and the KKT certificate. Build first (python3 __graft_entry__.py): nothing is compiled here."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from restartsqp_amd import capi, problems

which = sys.argv[1] if len(sys.argv) > 1 else "dense"
q = problems.dense_qp() if which == "dense" else problems.sparse_qp(band=5 if which == "band5" else 0)
s = capi.Solver(q.nV, q.nC)
s.set_options(qp_maxiter=400000)
s.set_A_csc(q.A_jc, q.A_ir, q.A_val); s.set_H_csc(q.H_jc, q.H_ir, q.H_val)
for w, v in zip(range(5), (q.g, q.lb, q.ub, q.lbA, q.ubA)):
    s.set_vector(w, v)
t = time.perf_counter(); n = s.optimize_qp(); t = time.perf_counter() - t
ok, st, _, _ = s.test_optimality()
print("%s cold: %.3f s, nWSR %d, %.1f us per working-set change, KKT %.2e certified %d, path: %s" % (q.name, t, n, 1e6 * t / max(n, 1), st.KKT_error, ok, capi.Solver.LARGE_PATHS[s.large_path()]), flush=True)
if which != "dense":
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    for k, (qk, changed) in enumerate(problems.sparse_sequence(q, nsteps=steps)):
        t = time.perf_counter()
        for w, v in zip(range(5), (qk.g, qk.lb, qk.ub, qk.lbA, qk.ubA)):
            s.set_vector(w, v)
        if changed:
            s.set_A_csc(qk.A_jc, qk.A_ir, qk.A_val)
        nk = s.optimize_qp()
        okk, stk, _, _ = s.test_optimality()
        t = time.perf_counter() - t
        print("step %d %s: %.3f s, nWSR %d, certified %d" % (k + 1, "VARIED" if changed else "FIXED", t, nk, okk), flush=True)
