"""Regenerates tests/golden/oracle_sparse_sequence_2500_reference_rule.json: the ORACLE's answers for the sparse
configuration at n = 2 500, m = 5 000 (50 000 non-zeros) driven through the restated optimizeQP dispatch
(oracle.OracleInterface) under the REFERENCE's re-initialisation rule (qpOASESInterface.cpp:199-207: init(.., x_qp,
y_qp, &bounds), no guessed constraints): the cold start, then 4 steps of problems.sparse_sequence (FIXED, VARIED =
flip, FIXED, VARIED = flip). Comparison data for the HIP engine -- not outputs of the reference (qpOASES is not
available: "parity unpinned"). Minutes of CPU; the tests only read the JSON.
`python make_sequence_golden.py 2500 4 5` writes the same for the 5-band Hessian of problems.sparse_qp(band=5)
(oracle_sparse_band5_sequence_2500_reference_rule.json)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402
from restartsqp_amd import problems  # noqa: E402


def main(n=2500, nsteps=4, band=0):
    q = problems.sparse_qp(n, 2 * n, 20 * n, band=band)
    oi = O.OracleInterface(q.nV, q.nC, qp_maxiter=400000, from_y0=False)
    oi.set_A_csc(q.A_jc, q.A_ir, q.A_val); oi.qp.set_H_csc(q.H_jc, q.H_ir, q.H_val)
    steps = []

    def record(used, t):
        qp = oi.qp
        steps.append(dict(mode=oi.modes[-1], nWSR=used, exitflag=qp.exitflag(), seconds=t, x=qp.x.tolist(), y=qp.y.tolist(),
                          ws_b=qp.ws_bounds.tolist(), ws_c=qp.ws_constraints.tolist()))
        print(oi.modes[-1], used, qp.exitflag(), "%.1f s" % t, flush=True)

    t = time.time(); used = oi.optimize_qp(q.g, q.lb, q.ub, q.lbA, q.ubA); record(used, time.time() - t)
    for qk, changed in problems.sparse_sequence(q, nsteps=nsteps):
        if changed:
            oi.set_A_csc(qk.A_jc, qk.A_ir, qk.A_val)
        t = time.time(); used = oi.optimize_qp(qk.g, qk.lb, qk.ub, qk.lbA, qk.ubA); record(used, time.time() - t)
    path = os.path.join(ROOT, "tests/golden/oracle_sparse%s_sequence_%d_reference_rule.json" % ("_band5" if band else "", n))
    with open(path, "w") as f:
        json.dump(dict(n=n, band=band, rule="reference (constraints from A x0 only)", steps=steps), f)
    print("wrote", path)


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
