"""Regenerates tests/golden/oracle_qp_solutions.json.

The file holds the ORACLE's answers (oracle/qp_oracle.c) for the reference's 18 QP dumps and
the derived hs071 first QP. They are regression vectors for the oracle and comparison data
for the HIP engine -- NOT outputs of the reference: qpOASES is not available, and the
reference commits no expected solutions ("parity unpinned")."""
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402
from restartsqp_amd import problems  # noqa: E402
from restartsqp_amd.qpdump import read_qore_dump  # noqa: E402


def main():
    qps = [problems.hs071_first_qp()]
    qps += [read_qore_dump(p) for p in sorted(glob.glob(os.path.join(ROOT, "tests/golden/qore_dumps/*.log")))]
    out = {}
    for q in qps:
        qp = O.OracleQP(q.nV, q.nC)
        qp.set_A_csc(q.A_jc, q.A_ir, q.A_val)
        qp.set_H_csc(q.H_jc, q.H_ir, q.H_val)
        rc, n = qp.init(q.g, q.lb, q.ub, q.lbA, q.ubA, 1000)
        out[q.name] = dict(rc=rc, nWSR=n, exitflag=qp.exitflag(), objective=qp.objective, nflips=qp.nflips(),
                           x=qp.x.tolist(), y=qp.y.tolist(), ws_b=qp.ws_bounds.tolist(), ws_c=qp.ws_constraints.tolist())
    with open(os.path.join(ROOT, "tests/golden/oracle_qp_solutions.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote", len(out), "solutions")


if __name__ == "__main__":
    main()
