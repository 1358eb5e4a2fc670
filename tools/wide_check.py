"""Tuning check for the mid-size LDS kernels: the 512-QP hs0xx batch (BASELINE configs[4]) and its 69 x 28 members
alone -- time per launch and a digest of the answers (nWSR, working sets, x) against the CPU oracle.
Run once with the product library and once with RSQP_LIB=<tuning build>."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from restartsqp_amd import capi, parallel, problems
import oracle as O

allp = problems.hs_batch(512)
order = parallel.balanced_order(allp)
sets = {"512": [allp[k] for k in order], "69x28": [p for p in allp if p.nV == 69]}
check = "--check" in sys.argv
for name, probs in sets.items():
    b = capi.Batch(probs)
    b.solve(capi.MODE_COLD, 1000)
    ms = []
    for _ in range(10):
        b.solve(capi.MODE_COLD, 1000)
        ms.append(b.last_solve_ms())
    res = b.results()
    h = hashlib.sha1()
    for r in res:
        h.update(r["ws_b"].tobytes()); h.update(r["ws_c"].tobytes()); h.update(np.int64(r["nWSR"]).tobytes())
    out = {"set": name, "n": len(probs), "ms_median": float(np.median(ms)), "nWSR_sum": int(sum(r["nWSR"] for r in res)),
           "nWSR_max": int(max(r["nWSR"] for r in res)), "ws_digest": h.hexdigest()[:12],
           "solved": int(sum(r["status"] == 20 for r in res))}
    if check:
        bad = 0
        for q, r in zip(probs, res):
            qp = O.OracleQP(q.nV, q.nC)
            qp.set_A_csc(q.A_jc, q.A_ir, q.A_val); qp.set_H_csc(q.H_jc, q.H_ir, q.H_val)
            rc, n = qp.init(q.g, q.lb, q.ub, q.lbA, q.ubA, 1000)
            same = (r["status"] == qp.exitflag() and np.array_equal(r["ws_b"], qp.ws_bounds) and
                    np.array_equal(r["ws_c"], qp.ws_constraints) and
                    np.abs(r["x"] - qp.x).max() <= 1e-9 * max(1.0, np.abs(qp.x).max()) and
                    (r["nWSR"] == n or q.name == "hs071_first_qp"))
            bad += not same
        out["oracle_mismatches"] = bad
    print(json.dumps(out), flush=True)
