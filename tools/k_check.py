"""GPU check of the KKT-tableau kernel (qp_small_g.h) on the 512-QP hs0xx batch and on random convex QPs:
every member against the oracle (status, working sets, nWSR, x / y to 1e-9), and the batch time with and without it.
Usage (GPU box): python tools/k_check.py [nrandom]"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(tag, probs, reps=20):
    import oracle as O
    from restartsqp_amd import capi
    b = capi.Batch(probs)
    b.set_keep_state(False)
    b.solve(capi.MODE_COLD, 1000)
    res = b.results()
    ok, kkt = b.test_optimality()
    bad = 0
    for k, (q, r) in enumerate(zip(probs, res)):
        qp = O.OracleQP(q.nV, q.nC)
        qp.set_A_csc(q.A_jc, q.A_ir, q.A_val); qp.set_H_csc(q.H_jc, q.H_ir, q.H_val)
        rc, n = qp.init(q.g, q.lb, q.ub, q.lbA, q.ubA, 1000)
        same = (r["status"] == qp.exitflag() and r["nWSR"] == n and np.array_equal(r["ws_b"], qp.ws_bounds)
                and np.array_equal(r["ws_c"], qp.ws_constraints))
        if same and rc == 0:
            same = (np.abs(r["x"] - qp.x).max() <= 1e-9 * max(1.0, np.abs(qp.x).max())
                    and np.abs(r["y"] - qp.y).max() <= 1e-9 * max(1.0, np.abs(qp.y).max()))
        if not same:
            bad += 1
            if bad <= 10:
                print("  MISMATCH", k, q.name, q.nV, q.nC, "status", r["status"], qp.exitflag(), "nWSR", r["nWSR"], n, flush=True)
    ms = []
    for _ in range(reps):
        b.solve(capi.MODE_COLD, 1000, sync=True)
        ms.append(b.last_solve_ms())
    print("%s: %d QPs, %d mismatches, certificate failures %d, median %.3f ms (min %.3f)" %
          (tag, len(probs), bad, sum(1 for o in ok if o != 1), float(np.median(ms)), float(np.min(ms))), flush=True)
    b.close()
    return bad


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        from restartsqp_amd import parallel, problems
        allp = problems.hs_batch(512)
        order = parallel.balanced_order(allp)
        bad = run("hs0xx batch 512", [allp[k] for k in order])
        shard = parallel.balanced_shards(allp, 8)[0]
        bad += run("64-QP shard", [allp[k] for k in shard])
        nr = int(sys.argv[2])
        if nr > 0:
            rng = np.random.default_rng(99)
            probs = [problems.random_qp(rng, int(rng.integers(9, 70)), int(rng.integers(1, 29)), density=float(rng.choice([0.2, 0.5, 1.0])))
                     for _ in range(nr)]
            bad += run("random convex", probs, reps=3)
            probs = [problems.random_qp(rng, int(rng.integers(33, 64)), int(rng.integers(20, 64)), density=float(rng.choice([0.2, 0.5, 1.0])))
                     for _ in range(nr // 2)]
            bad += run("random convex 33-63 x 20-63 (64 x 64 build)", probs, reps=3)
        sys.exit(1 if bad else 0)
    nr = sys.argv[1] if len(sys.argv) > 1 else "300"
    rc = 0
    for env in ({}, {"RSQP_SMALL_NO_KKT": "1"}):
        print("== env", env, flush=True)
        rc |= subprocess.call([sys.executable, os.path.abspath(__file__), "--child", nr], env=dict(os.environ, **env))
    sys.exit(rc)
