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


def large(which):
    """Full-size oracle answers for BASELINE configs 3 / 4 (minutes to hours of CPU; run in the build
    container, the tests only read the JSON): dense 2048 x 4096 cold start, sparse cold start at the
    size given on the command line. Inputs are regenerated from their seeds by the tests."""
    import time
    if which == "dense":
        q = problems.dense_qp()
    else:
        n = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
        q = problems.sparse_qp(n, 2 * n, 20 * n)
    qp = O.OracleQP(q.nV, q.nC)
    qp.set_A_csc(q.A_jc, q.A_ir, q.A_val)
    qp.set_H_csc(q.H_jc, q.H_ir, q.H_val)
    t = time.time()
    rc, n = qp.init(q.g, q.lb, q.ub, q.lbA, q.ubA, 400000)
    t = time.time() - t
    out = dict(name=q.name, rc=rc, nWSR=n, exitflag=qp.exitflag(), objective=qp.objective, nflips=qp.nflips(),
               oracle_seconds_build_container=t, x=qp.x.tolist(), y=qp.y.tolist(),
               ws_b=qp.ws_bounds.tolist(), ws_c=qp.ws_constraints.tolist())
    path = os.path.join(ROOT, "tests/golden/oracle_large_%s.json" % q.name)
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, "nWSR", n, "seconds", t)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--large":
        large(sys.argv[2])
    else:
        main()
