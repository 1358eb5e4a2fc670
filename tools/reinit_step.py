"""Sparse 10 000 x 20 000: cold start, then FIXED / VARIED steps through rsqp_optimize_qp under the reference's re-initialisation
rule (no guessed constraints) -- the time of the re-initialisation dominated VARIED step.   python3 tools/reinit_step.py [nsteps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from restartsqp_amd import capi, problems
q = problems.sparse_qp()
s = capi.Solver(q.nV, q.nC)
s.set_options(qp_maxiter=400000)
s.set_A_csc(q.A_jc, q.A_ir, q.A_val); s.set_H_csc(q.H_jc, q.H_ir, q.H_val)
for w, v in zip(range(5), (q.g, q.lb, q.ub, q.lbA, q.ubA)):
    s.set_vector(w, v)
t = time.perf_counter(); n = s.optimize_qp(); t = time.perf_counter() - t
print("cold: nWSR %d in %.2f s" % (n, t), flush=True)
s.set_reinit_guess(False)
for qk, changed in problems.sparse_sequence(q, nsteps=int(sys.argv[1]) if len(sys.argv) > 1 else 2, seed=20260150):
    t = time.perf_counter()
    for w, v in zip(range(5), (qk.g, qk.lb, qk.ub, qk.lbA, qk.ubA)):
        s.set_vector(w, v)
    if changed:
        s.set_A_csc(qk.A_jc, qk.A_ir, qk.A_val)
    nk = s.optimize_qp()
    ok, st, _, _ = s.test_optimality()
    t = time.perf_counter() - t
    print("%s step: nWSR %d in %.3f s (%.3f ms per change) certified %s" % ("VARIED" if changed else "FIXED", nk, t, 1e3 * t / max(nk, 1), ok), flush=True)
