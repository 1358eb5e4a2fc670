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


def _parse(line):
    tok = line.split()
    k = tok.index("status")
    d = {tok[i]: tok[i + 1] for i in range(k, tok.index("x"), 2)}
    d["tag"] = " ".join(tok[:k])
    d["x"] = np.array([float(v) for v in tok[tok.index("x") + 1:]])
    return d


def test_cpp_penalty_update_and_soc_trace(capi, oracle):
    """Algorithm::update_penalty_parameter + second_order_correction (reference src/Algorithm.cpp:886-1028,
    1144-1211) replayed by host_replay --penalty: an LP and a QP solver object side by side, solveLP, then
    update_penalty + solveQP twice, the SOC solve, the restore, and a second LP on the same LP object (hot start
    with a new gradient: the regularisation of the first init must stay in force). Expected values: the oracle
    driven by a restatement of the dispatch decisions (qpOASESInterface.cpp:137-284)."""
    subprocess.check_call(["make", "-s", "-C", HOST, "host_replay"])
    out = subprocess.run([os.path.join(HOST, "host_replay"), "--penalty"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    got = [_parse(l) for l in out.stdout.strip().splitlines()]
    assert [g["tag"] for g in got] == ["qp rho=1", "lp rho=1", "qp rho=10", "qp rho=100", "qp soc", "qp restored", "lp rho=100"]

    def nlp_at(x):
        d = problems.hs071_nlp(x, lam=np.zeros(2))
        d["c_u"] = np.array([np.inf, np.inf])
        return d

    nlp = nlp_at([1.0, 2.0, 2.0, 1.0])
    delta, EPS = 0.25, 2.221e-16
    q = problems.handler_qp(nlp, delta=delta, rho=1.0)
    exp, total = [], 0
    # QP object
    qp = oracle.OracleQP(q.nV, q.nC)
    qp.set_A_csc(q.A_jc, q.A_ir, q.A_val); qp.set_H_csc(q.H_jc, q.H_ir, q.H_val)
    rc, n = qp.init(q.g, q.lb, q.ub, q.lbA, q.ubA, 1000); total += n
    exp.append((total, qp.x.copy(), qp.objective))
    assert np.abs(qp.x[4:]).sum() > 1e-3          # the slacks are positive: the penalty update has work to do
    # LP object: H = 0 -> regVal*I, one regularisation step (optimizeLP)
    g_lp = np.concatenate([np.zeros(4), np.ones(4)])
    lp = oracle.OracleQP(q.nV, q.nC)
    lp.set_A_csc(q.A_jc, q.A_ir, q.A_val); lp.set_H_csc(None, None, None)
    reg = np.linalg.norm(g_lp) * 1e3 * EPS
    lp.set_regularisation(reg)
    rc, n1 = lp.init(g_lp, q.lb, q.ub, q.lbA, q.ubA, 100)
    rc2, n2 = lp.hotstart(g_lp - reg * lp.x, q.lb, q.ub, q.lbA, q.ubA, 100); total += n1 + n2
    assert rc == 0 and rc2 == 0
    exp.append((total, lp.x.copy(), float(g_lp @ lp.x)))
    # update_penalty + solveQP: matrices FIXED -> hotstart on vectors
    g = q.g.copy()
    for rho in (10.0, 100.0):
        g[4:] = rho
        rc, n = qp.hotstart(g, q.lb, q.ub, q.lbA, q.ubA, 1000); total += n
        assert rc == 0
        exp.append((total, qp.x.copy(), qp.objective))
    # SOC: g[:4] = H p + grad, bounds at x_trial; ubA stays (QPhandler::update_bounds)
    p = qp.x[:4].copy()
    Hk = q.dense_H()[:4, :4]
    trial = nlp_at(nlp["x"] + p)
    g_soc = g.copy(); g_soc[:4] = Hk @ p + nlp["grad"]
    lb, ub = q.lb.copy(), q.ub.copy()
    lb[:4] = np.maximum(nlp["x_l"] - trial["x"], -delta); ub[:4] = np.minimum(nlp["x_u"] - trial["x"], delta)
    lbA = nlp["c_l"] - trial["c"]
    rc, n = qp.hotstart(g_soc, lb, ub, lbA, q.ubA, 1000); total += n
    assert rc == 0
    exp.append((total, qp.x.copy(), qp.objective))
    rc, n = qp.hotstart(g, q.lb, q.ub, q.lbA, q.ubA, 1000); total += n
    exp.append((total, qp.x.copy(), qp.objective))
    # second LP on the same object: set_A marked the Jacobian as updated -> old status VARIED, new UNDEFINED ->
    # hotstart with matrices (:246-250); regVal of the first init is kept
    g_lp2 = np.concatenate([np.zeros(4), 100.0 * np.ones(4)])
    rc, n1 = lp.hotstart_matrices(g_lp2, q.lb, q.ub, q.lbA, q.ubA, 100)
    rc2, n2 = lp.hotstart(g_lp2 - reg * lp.x, q.lb, q.ub, q.lbA, q.ubA, 100); total += n1 + n2
    assert rc == 0 and rc2 == 0
    exp.append((total, lp.x.copy(), float(g_lp2 @ lp.x)))
    for gline, (it, x, obj) in zip(got, exp):
        assert gline["status"] == "20", gline
        assert int(gline["qp_iter"]) == it, (gline["tag"], gline["qp_iter"], it)
        assert np.abs(gline["x"] - x).max() <= 1e-9 * max(1.0, np.abs(x).max()), gline["tag"]
        assert abs(float(gline["obj"]) - obj) <= 1e-9 * max(1.0, abs(obj)), gline["tag"]


@pytest.mark.parametrize("name", ["QORE_hs015qpdata.log", "QORE_hs074qpdata.log", "QORE_hs116qpdata.log"])
def test_cpp_data_ctor_getters_and_dump_writer(capi, oracle, name, tmp_path):
    """Plain-QP ctor with data (qpOASESInterface.cpp:54-94) on a reference dump, the data getters as
    QPhandler::get_active_set uses them (QPhandler.cpp:596-650) and WriteQPDataToFile (qpOASESInterface.cpp:791-814,
    QOREInterface.cpp:582-598): the QORE-layout file equals the dump it came from token for token."""
    from restartsqp_amd import qpdump
    subprocess.check_call(["make", "-s", "-C", HOST, "host_replay"])
    src = os.path.join(ROOT, "tests", "golden", "qore_dumps", name)
    out = subprocess.run([os.path.join(HOST, "host_replay"), "--dump", src, "rt.log"], capture_output=True, text=True,
                         timeout=120, cwd=str(tmp_path))
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    q = qpdump.read_qore_dump(src)
    s = capi.Solver(q.nV, q.nC)
    s.set_A_csc(q.A_jc, q.A_ir, q.A_val); s.set_H_csc(q.H_jc, q.H_ir, q.H_val)
    for w, v in zip(range(5), (q.g, q.lb, q.ub, q.lbA, q.ubA)):
        s.set_vector(w, v)
    n = s.optimize_qp()
    head = lines[0].split()
    assert head[:6] == ["dump", "status", str(s.status), "qp_iter", str(n), "solved"]
    x = s.x
    Ax = oracle.sphb_times(q.nC, q.nV, q.A_jc, q.A_ir, q.A_val, x)
    tol = 1.4901161193847656e-08

    def act(v, lo, hi):
        return [(-99 if abs(h - a) < tol else -1) if abs(a - l) < tol else (1 if abs(h - a) < tol else 0)
                for a, l, h in zip(v, lo, hi)]
    assert [int(t) for t in lines[1].split()[1:]] == act(x, q.lb, q.ub)
    assert [int(t) for t in lines[2].split()[1:]] == act(Ax, q.ubA, q.ubA)       # the reference reads getUbA() twice (:641-642)
    got_Ax = np.array([float(t) for t in lines[3].split()[1:]])
    assert np.abs(got_Ax - Ax).max() <= 1e-12 * max(1.0, np.abs(Ax).max())
    assert lines[4].split() == ["getG", str(q.nV), "getH_nnz", str(len(q.H_val)), "getA_nnz", str(len(q.A_val))]
    assert open(tmp_path / "QORE_rt.log").read().split() == open(src).read().split()
    ref = str(tmp_path / "expected_qpoases.log")
    qpdump.write_qpoases_dump(ref, q)
    assert open(tmp_path / "qpOASESrt.log").read() == open(ref).read()
