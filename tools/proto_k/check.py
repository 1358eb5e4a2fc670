"""Validation of the KKT-tableau formulation (tools/proto_k/proto_g.cpp, the CPU prototype of restartsqp_amd/csrc/qp_small_g.h)
against the oracle: status, working sets, nWSR identical, x / y to 1e-9 -- or a clean BAIL. Usage: python tools/proto_k/check.py [what]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402
from restartsqp_amd import problems  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
WHICH = "g"      # the tableau formulation (proto_g.cpp; round 3's explicit-KKT-inverse prototype went with its kernel)
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", os.path.join(HERE, "libproto%s.so" % WHICH), os.path.join(HERE, "proto_%s.cpp" % WHICH)])
L = C.CDLL(os.path.join(HERE, "libproto%s.so" % WHICH))
dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
L.protok_solve.argtypes = [C.c_int, C.c_int] + [dp] * 7 + [C.c_int, dp, dp, ip, ip, ip, ip]


def solve_k(q, maxit=1000):
    A = np.asfortranarray(q.dense_A()) if q.nC else np.zeros((0, q.nV), order="F")
    H = np.asfortranarray(q.dense_H())
    x, y = np.zeros(q.nV), np.zeros(q.nV + q.nC)
    Sb, Sc = np.zeros(q.nV, np.int32), np.zeros(max(q.nC, 1), np.int32)
    n, info = C.c_int(0), (C.c_int * 4)()
    P = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(dp)
    Af, Hf = A.ravel(order="F").copy(), H.ravel(order="F").copy()
    v = [np.ascontiguousarray(t, dtype=np.float64) for t in (q.g, q.lb, q.ub, q.lbA if q.nC else np.zeros(1), q.ubA if q.nC else np.zeros(1))]
    rc = L.protok_solve(q.nV, q.nC, P(Af) if q.nC else P(np.zeros(1)), P(Hf), *[t.ctypes.data_as(dp) for t in v], maxit, x.ctypes.data_as(dp),
                        y.ctypes.data_as(dp), Sb.ctypes.data_as(ip), Sc.ctypes.data_as(ip), C.byref(n), info)
    return rc, n.value, x, y, Sb, Sc[:q.nC], list(info)


def compare(q):
    """returns 'same' / 'bail<reason>' / 'DIFF...'"""
    rc, n, x, y, Sb, Sc, info = solve_k(q)
    if rc == 9:
        return "bail%d" % info[0]
    qp = O.OracleQP(q.nV, q.nC)
    qp.set_A_csc(q.A_jc, q.A_ir, q.A_val); qp.set_H_csc(q.H_jc, q.H_ir, q.H_val)
    rco, no = qp.init(q.g, q.lb, q.ub, q.lbA, q.ubA, 1000)
    if rc != rco:
        return "DIFF rc %d vs %d" % (rc, rco)
    if n != no:
        return "DIFF nWSR %d vs %d" % (n, no)
    if not (np.array_equal(Sb, qp.ws_bounds) and np.array_equal(Sc, qp.ws_constraints)):
        return "DIFF working set"
    if rc == 0:
        ex = np.abs(x - qp.x).max() / max(1.0, np.abs(qp.x).max()); ey = np.abs(y - qp.y).max() / max(1.0, np.abs(qp.y).max())
        if ex > 1e-9 or ey > 1e-9:
            return "DIFF x %.1e y %.1e" % (ex, ey)
    return "same"


def tally(name, probs):
    from collections import Counter
    c = Counter()
    for q in probs:
        r = compare(q)
        c[r if not r.startswith("DIFF") else "DIFF"] += 1
        if r.startswith("DIFF"):
            print("   ", q.name, q.nV, q.nC, r)
    print(name, dict(c), flush=True)
    return c


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "batch"):
        tally("hs_batch(512)", problems.hs_batch(512))
    if what in ("all", "random"):
        rng = np.random.default_rng(77)
        tally("random convex 2000", [problems.random_qp(rng, int(rng.integers(9, 70)), int(rng.integers(1, 40)), density=float(rng.choice([0.2, 0.5, 1.0])))
                                     for _ in range(2000)])
    if what in ("all", "degenerate"):
        rng = np.random.default_rng(78)
        tally("degenerate 3000", [problems.degenerate_qp(rng, int(rng.integers(0, 5))) for _ in range(3000)])
    if what in ("all", "dumps"):
        import glob
        from restartsqp_amd.qpdump import read_qore_dump
        tally("reference dumps", [read_qore_dump(p) for p in sorted(glob.glob(os.path.join(ROOT, "tests/golden/qore_dumps/*.log")))])
