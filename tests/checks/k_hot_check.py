"""Hot starts on the KKT-tableau kernel (restartsqp_amd/csrc/qp_small_g.h) against the CPU oracle: batches whose members have 20..60
variables, cold start, then three hot starts on perturbed vectors; every member every step: nWSR, working sets, x / y.
With RSQP_K_DEBUG_BAIL=n in the environment every hot start of that kernel bails out before its n-th change and the
null-space kernel takes the member over from the stored state (factors rebuilt for the stored working set, homotopy data
kept) -- the answers must not change. A third of the members are non-convex (always solved by the null-space kernel) to
cover states of both kinds in one batch. Usage (GPU box): python tests/checks/k_hot_check.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from restartsqp_amd import capi, problems
from restartsqp_amd.qpdump import QPData, dense_to_csc
import oracle as O
O.build()
rng = np.random.default_rng(21)
probs = []
for k in range(36):
    q = problems.random_qp(rng, int(rng.integers(20, 60)), int(rng.integers(5, 30)))
    if k % 3 == 2:      # an indefinite Hessian: flipping bounds, the tableau kernel bails out of the cold start
        H = q.dense_H(); H[0, 0] = -abs(H[0, 0]); H[1, 1] = 0.0; H[1, :] = 0.0; H[:, 1] = 0.0
        q = QPData(q.nV, q.nC, *dense_to_csc(H), q.A_jc, q.A_ir, q.A_val, q.g, q.lb, q.ub, q.lbA, q.ubA, name="indefinite")
    probs.append(q)
b = capi.Batch(probs)
b.solve(capi.MODE_COLD, 1000)
orc, bad = [], 0
for q, r in zip(probs, b.results()):
    qp = O.OracleQP(q.nV, q.nC); qp.set_A_csc(q.A_jc, q.A_ir, q.A_val); qp.set_H_csc(q.H_jc, q.H_ir, q.H_val)
    rc, n = qp.init(q.g, q.lb, q.ub, q.lbA, q.ubA, 1000)
    orc.append(qp)
    same = r["status"] == qp.exitflag() and (qp.nflips() > 0 or n == r["nWSR"])
    if same and qp.nflips() == 0:
        same = np.array_equal(r["ws_b"], qp.ws_bounds) and np.array_equal(r["ws_c"], qp.ws_constraints)
    if not same:
        bad += 1; print("cold MISMATCH", q.name, q.nV, q.nC, r["status"], qp.exitflag(), r["nWSR"], n)
flipped = [qp.nflips() > 0 for qp in orc]
for step in range(3):
    probs = [problems.perturb(rng, q, 0.05) for q in probs]
    b.set_vectors_from(probs)
    b.solve(capi.MODE_HOT_VECTORS, 1000)
    for k, (q, qp, r) in enumerate(zip(probs, orc, b.results())):
        rc, n = qp.hotstart(q.g, q.lb, q.ub, q.lbA, q.ubA, 1000)
        flipped[k] = flipped[k] or qp.nflips() > 0
        if flipped[k]:       # non-convex members: the path is not unique under rounding once a bound was flipped
            same = r["status"] == qp.exitflag()
        else:
            same = (r["status"] == qp.exitflag() and n == r["nWSR"] and np.array_equal(r["ws_b"], qp.ws_bounds) and np.array_equal(r["ws_c"], qp.ws_constraints)
                    and np.abs(r["x"] - qp.x).max() <= 1e-9 * max(1.0, np.abs(qp.x).max()) and np.abs(r["y"] - qp.y).max() <= 1e-9 * max(1.0, np.abs(qp.y).max()))
        if not same:
            bad += 1; print("step", step, "MISMATCH", q.name, q.nV, q.nC, "nWSR", r["nWSR"], n, "status", r["status"], qp.exitflag(), flush=True)
print("K HOT", "FAILED" if bad else "OK", "hook", os.environ.get("RSQP_K_DEBUG_BAIL"), "convex members", sum(1 for f in flipped if not f))
sys.exit(1 if bad else 0)
