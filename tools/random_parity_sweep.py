"""Extended randomized parity sweep (not part of the test suite): random and degenerate QPs of mid size through the
batch API (explicit-inverse LDS kernel for nV > 8) against the CPU oracle -- status, working sets, nWSR bit-exact,
x / y to 1e-9. Usage: python tools/random_parity_sweep.py [count] [seed] [--degenerate]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from restartsqp_amd import capi, problems
import oracle as O
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 1500
rng = np.random.default_rng(int(args[1]) if len(args) > 1 else 77)
probs = []
only_degenerate = "--degenerate" in sys.argv
for k in range(n):
    if only_degenerate or k % 7 == 6:
        probs.append(problems.degenerate_qp(rng, k % 5))
    else:
        probs.append(problems.random_qp(rng, int(rng.integers(9, 70)), int(rng.integers(1, 40)), density=float(rng.uniform(0.2, 0.9))))
bad = 0
for lo in range(0, n, 250):
    chunk = probs[lo:lo + 250]
    b = capi.Batch(chunk)
    b.solve(capi.MODE_COLD, 2000)
    res = b.results()
    for q, r in zip(chunk, res):
        qp = O.OracleQP(q.nV, q.nC)
        qp.set_A_csc(q.A_jc, q.A_ir, q.A_val); qp.set_H_csc(q.H_jc, q.H_ir, q.H_val)
        rc, nw = qp.init(q.g, q.lb, q.ub, q.lbA, q.ubA, 2000)
        same = bool(r["status"] == 20) == qp.is_solved() and np.array_equal(r["ws_b"], qp.ws_bounds) and np.array_equal(r["ws_c"], qp.ws_constraints) and r["nWSR"] == nw
        sc = max(1.0, float(np.abs(qp.x).max()), float(np.abs(qp.y).max()))
        close = (not qp.is_solved()) or (np.abs(r["x"] - qp.x).max() <= 1e-9 * sc and np.abs(r["y"] - qp.y).max() <= 1e-9 * sc)
        if not (same and close):
            bad += 1
            if bad <= 5:
                print("MISMATCH nV %d nC %d: nWSR %d vs %d, ws same %s, x/y close %s" % (q.nV, q.nC, r["nWSR"], nw, same, close), flush=True)
    print("checked %d, mismatches so far %d" % (min(lo + 250, n), bad), flush=True)
print("DONE: %d problems, %d mismatches" % (n, bad))
