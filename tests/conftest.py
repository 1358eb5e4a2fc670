import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def dump_paths():
    return sorted(glob.glob(os.path.join(GOLDEN, "qore_dumps", "QORE_hs*qpdata.log")))


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def capi():
    """The product library. GPU tests must run the HIP path: fail (not skip) if it is missing."""
    from restartsqp_amd import build, capi as c
    build.build_lib()
    c.lib()
    return c


def clamp(v):
    return np.clip(np.asarray(v, float), -1.0e20, 1.0e20)


def oracle_cold(O, q, nWSR=1000):
    qp = O.OracleQP(q.nV, q.nC)
    qp.set_A_csc(q.A_jc, q.A_ir, q.A_val)
    qp.set_H_csc(q.H_jc, q.H_ir, q.H_val)
    rc, n = qp.init(q.g, q.lb, q.ub, q.lbA, q.ubA, nWSR)
    return qp, rc, n


def oracle_certificate(O, q, x, y, ws_b, ws_c):
    A = (q.A_jc, q.A_ir, q.A_val)
    H = (q.H_jc, q.H_ir, q.H_val)
    Wb, Wc = O.kkt_get_working_set(q.nV, q.nC, A, x, clamp(q.lb), clamp(q.ub), clamp(q.lbA), clamp(q.ubA), ws_b, ws_c)
    ok, st = O.kkt_test_optimality(q.nV, q.nC, A, H, q.g, clamp(q.lb), clamp(q.ub), clamp(q.lbA), clamp(q.ubA), x, y,
                                   Wb, Wc)
    return ok, st, Wb, Wc
