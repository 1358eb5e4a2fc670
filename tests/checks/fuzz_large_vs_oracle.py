"""Randomised check of the HBM-resident engine against the CPU oracle: dense and sparse problems of
50-160 variables, cold start, hot start on vectors, hot start with new matrices (blocked QR /
Cholesky set-up), warm re-initialisation. Usage (GPU box): python tests/checks/fuzz_large_vs_oracle.py [seed] [count] [band]
(band: any third argument -> 5-band Hessians, i.e. the banded H^-1 operator of the general range-space path, some with free variables)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from restartsqp_amd import capi, problems
import oracle as O
O.build()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 8
band = len(sys.argv) > 3
rng = np.random.default_rng(seed)


def same(s, n, qp, n_or, what, q):
    wb, wc = s.working_set_raw()
    if qp.exitflag() in (22, 23):
        # infeasible / unbounded: the verdict is what is compared -- the multipliers grow without bound on the way there, the count of
        # changes in front of the stop differs between formulations (null-space 525, oracle 527 on the 142 x 208 member of seed 14),
        # and a range-space attempt that lost its pivots is re-run on the null-space path (both attempts are counted)
        ok = s.status == qp.exitflag()
        if not ok:
            print("MISMATCH", what, q.name, q.nV, q.nC, "status", s.status, qp.exitflag())
        return ok
    ok = (s.status == qp.exitflag() and n == n_or and np.array_equal(wb, qp.ws_bounds) and np.array_equal(wc, qp.ws_constraints)
          and np.abs(s.x - qp.x).max() <= 1e-9 * max(1.0, np.abs(qp.x).max()) and np.abs(s.y - qp.y).max() <= 1e-9 * max(1.0, np.abs(qp.y).max()))
    if not ok:
        print("MISMATCH", what, q.name, q.nV, q.nC, "nWSR", n, n_or, "status", s.status, qp.exitflag())
    return ok


bad = 0
for k in range(count):
    nV = int(rng.integers(50, 160)); nC = int(rng.integers(30, 220))
    dens = float(rng.choice([0.05, 0.3, 1.0]))
    q = problems.banded_qp(rng, nV, nC, density=dens, hb=int(rng.integers(1, 3)), free=bool(k % 3 == 0)) if band else problems.random_qp(rng, nV, nC, density=dens)
    s = capi.Solver(q.nV, q.nC); s.set_engine(2); s.set_options(100000, 100)
    s.set_A_csc(q.A_jc, q.A_ir, q.A_val); s.set_H_csc(q.H_jc, q.H_ir, q.H_val)
    for w, v in zip(range(5), (q.g, q.lb, q.ub, q.lbA, q.ubA)):
        s.set_vector(w, v)
    qp = O.OracleQP(q.nV, q.nC); qp.set_A_csc(q.A_jc, q.A_ir, q.A_val); qp.set_H_csc(q.H_jc, q.H_ir, q.H_val)
    n = s.solve(capi.MODE_COLD, 100000); rc, n_or = qp.init(q.g, q.lb, q.ub, q.lbA, q.ubA, 100000)
    bad += not same(s, n, qp, n_or, "cold", q)
    q2 = problems.perturb(rng, q, 0.03)
    for w, v in zip(range(5), (q2.g, q2.lb, q2.ub, q2.lbA, q2.ubA)):
        s.set_vector(w, v)
    n = s.solve(capi.MODE_HOT_VECTORS, 100000); rc, n_or = qp.hotstart(q2.g, q2.lb, q2.ub, q2.lbA, q2.ubA, 100000)
    bad += not same(s, n, qp, n_or, "hot vectors", q)
    A2 = q2.A_val * (1.0 + 0.01 * rng.normal(size=q2.A_val.shape))
    s.set_A_csc(q2.A_jc, q2.A_ir, A2); qp.set_A_csc(q2.A_jc, q2.A_ir, A2)
    n = s.solve(capi.MODE_HOT_MATRICES, 100000); rc, n_or = qp.hotstart_matrices(q2.g, q2.lb, q2.ub, q2.lbA, q2.ubA, 100000)
    bad += not same(s, n, qp, n_or, "hot matrices", q)
    x0, y0, gb = s.x, s.y, s.working_set_raw()[0]
    q3 = problems.perturb(rng, q2, 0.03)
    for w, v in zip(range(5), (q3.g, q3.lb, q3.ub, q3.lbA, q3.ubA)):
        s.set_vector(w, v)
    rule = bool(k & 1)                       # even problems: the reference's rule (default), odd: sides from sign(y0)
    s.set_reinit_guess(rule); qp.set_guess_constraints_from_y0(rule)
    n = s.solve(capi.MODE_WARM_REINIT, 100000, x0, y0, gb)
    rc, n_or = qp.init(q3.g, q3.lb, q3.ub, q3.lbA, q3.ubA, 100000, x0=x0, y0=y0, guess_b=gb)
    bad += not same(s, n, qp, n_or, "warm", q)
    s.close()
    print("problem %d (%d x %d): checked, mismatches so far %d" % (k, nV, nC, bad), flush=True)
print("FUZZ", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
