"""The C++ host adapter (restartsqp_amd/csrc/host: the QPSolverInterface subclass of
INTEGRATION.md) replaying QPhandler's call sequence for hs071 through the C ABI."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, oracle_cold
from restartsqp_amd import problems

pytestmark = pytest.mark.gpu
HOST = os.path.join(ROOT, "restartsqp_amd", "csrc", "host")


def test_cpp_adapter_replays_hs071(capi, oracle):
    subprocess.check_call(["make", "-s", "-C", HOST, "host_replay"])
    out = subprocess.run([os.path.join(HOST, "host_replay")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l.split() for l in out.stdout.strip().splitlines()]
    assert len(lines) == 2
    q = problems.hs071_first_qp()
    qp, rc, n0 = oracle_cold(oracle, q)
    nlp = problems.hs071_nlp()
    q2 = problems.handler_qp(nlp, delta=0.5)
    expected = [(n0, qp.x.copy(), qp.objective)]
    rc, n1 = qp.hotstart(q2.g, q2.lb, q2.ub, q2.lbA, q2.ubA, 1000)
    expected.append((n0 + n1, qp.x.copy(), qp.objective))
    for l, (it, x, obj) in zip(lines, expected):
        d = {l[i]: l[i + 1] for i in range(0, 12, 2)}
        assert d["status"] == "20" and int(d["qp_iter"]) == it and d["kkt_ok"] == "1"
        xs = np.array([float(v) for v in l[13:21]])
        assert np.abs(xs - x).max() < 1e-12 and abs(float(d["obj"]) - obj) < 1e-12
        assert l[21:] == ["Wc", "-99", "-99"]
