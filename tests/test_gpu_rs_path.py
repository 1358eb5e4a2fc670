"""The GENERAL range-space path of the HBM-resident engine (qp_rs_path.h / qp_rs_kernels.h): any symmetric positive
definite Hessian -- active bounds and constraints as rows of one matrix C, the explicit inverse of C H^-1 C', H^-1 as a banded
LDL' operator or a dense inverse -- against the oracle (working sets, status, nWSR bit-exact; x / y to 1e-9) and against the
null-space path of the same engine, in all four call shapes of optimizeQP (src/qpOASESInterface.cpp:155,180-206)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, oracle_cold
from restartsqp_amd import problems
from restartsqp_amd.qpdump import QPData, dense_to_csc
from test_gpu_large_engine import load, same_as_oracle

pytestmark = pytest.mark.gpu


banded_qp = problems.banded_qp


def all_call_shapes(capi, oracle, rng, q, want_path):
    s = load(capi, q)
    n = s.solve(capi.MODE_COLD, 20000)
    assert s.large_path() == want_path, capi.Solver.LARGE_PATHS[s.large_path()]
    qp, rc, n_or = oracle_cold(oracle, q, 20000)
    same_as_oracle(s, n, qp, n_or)
    ok, st, _, _ = s.test_optimality()
    assert ok and st.KKT_error < 1e-9
    q2 = problems.perturb(rng, q, 0.05)                                   # hotstart(g, lb, ub, lbA, ubA)
    for w, v in zip(range(5), (q2.g, q2.lb, q2.ub, q2.lbA, q2.ubA)):
        s.set_vector(w, v)
    n = s.solve(capi.MODE_HOT_VECTORS, 20000)
    rc, n_or = qp.hotstart(q2.g, q2.lb, q2.ub, q2.lbA, q2.ubA, 20000)
    same_as_oracle(s, n, qp, n_or)
    A2 = q2.A_val * (1.0 + 0.01 * rng.normal(size=q2.A_val.shape))        # hotstart(H, g, A, ...): blocked set-up of the guess
    s.set_A_csc(q2.A_jc, q2.A_ir, A2); s.set_H_csc(q2.H_jc, q2.H_ir, q2.H_val * 1.05)
    n = s.solve(capi.MODE_HOT_MATRICES, 20000)
    assert s.large_path() == want_path
    qp.set_A_csc(q2.A_jc, q2.A_ir, A2); qp.set_H_csc(q2.H_jc, q2.H_ir, q2.H_val * 1.05)
    rc, n_or = qp.hotstart_matrices(q2.g, q2.lb, q2.ub, q2.lbA, q2.ubA, 20000)
    same_as_oracle(s, n, qp, n_or)
    q3 = problems.perturb(rng, q2, 0.05)                                  # init(.., x0, y0, guessed bounds)
    for w, v in zip(range(5), (q3.g, q3.lb, q3.ub, q3.lbA, q3.ubA)):
        s.set_vector(w, v)
    x0, y0 = s.x.copy(), s.y.copy()
    wb, _ = s.working_set_raw()
    n = s.solve(capi.MODE_WARM_REINIT, 20000, x0=x0, y0=y0, guess_b=wb)
    rc, n_or = qp.init(q3.g, q3.lb, q3.ub, q3.lbA, q3.ubA, 20000, x0=x0, y0=y0, guess_b=wb)
    same_as_oracle(s, n, qp, n_or)
    ok, st, _, _ = s.test_optimality()
    assert ok and st.KKT_error < 1e-9


@pytest.mark.parametrize("knob", [None, "RSQP_LARGE_RSH_DENSE", "RSQP_LARGE_RSH_DENSE+RSQP_LARGE_NO_TABLEAU", "RSQP_NO_BLOCKED_SETUP", "RSQP_LARGE_BAND_1WG",
                                  "RSQP_LARGE_NO_CARRY", "RSQP_LARGE_NO_LAZY"])
def test_banded_hessian_all_call_shapes(capi, oracle, monkeypatch, knob):
    for k in (knob or "").split("+"):
        if k:
            monkeypatch.setenv(k, "1")
    rng = np.random.default_rng(501)
    want = {"RSQP_LARGE_RSH_DENSE": 4, "RSQP_LARGE_RSH_DENSE+RSQP_LARGE_NO_TABLEAU": 3}.get(knob, 2)
    for nV, nC, hb, free in ((40, 30, 2, False), (97, 140, 2, False), (150, 90, 1, False), (64, 80, 2, True), (33, 0, 2, False)):
        all_call_shapes(capi, oracle, rng, banded_qp(rng, nV, nC, hb=hb, free=free), want)


@pytest.mark.parametrize("tableau", [True, False])
def test_dense_hessian_all_call_shapes(capi, oracle, monkeypatch, tableau):
    """dense H^-1; with the static tableau [I; A] H^-1 [I A'] (the default for nV + nC <= 8192: products of a row, step direction
    and set-up matrix are gathers / one gathered-column product) and without it"""
    if not tableau:
        monkeypatch.setenv("RSQP_LARGE_NO_TABLEAU", "1")
    rng = np.random.default_rng(502)
    for nV, nC, dens in ((30, 25, 0.3), (90, 60, 0.3), (50, 130, 1.0), (70, 0, 0.3)):
        all_call_shapes(capi, oracle, rng, problems.random_qp(rng, nV, nC, dens), 4 if tableau else 3)


def test_same_path_as_the_null_space_formulation(capi, monkeypatch):
    """sparse configuration with the 5-band Hessian at n = 600: the general range-space path and the null-space path of the same
    engine walk the same homotopy (nWSR, working sets) to the same point"""
    q = problems.sparse_qp(600, 1200, 12000, band=5)
    s = load(capi, q)
    n = s.solve(capi.MODE_COLD, 100000)
    assert s.large_path() == 2
    monkeypatch.setenv("RSQP_LARGE_NO_RSH", "1")
    t = load(capi, q)
    m = t.solve(capi.MODE_COLD, 100000)
    assert t.large_path() == 0
    assert n == m and s.status == t.status == 20
    wb, wc = s.working_set_raw(); vb, vc = t.working_set_raw()
    assert np.array_equal(wb, vb) and np.array_equal(wc, vc)
    assert np.abs(s.x - t.x).max() <= 1e-9 and np.abs(s.y - t.y).max() <= 1e-9 * max(1.0, np.abs(t.y).max())


def test_indefinite_or_unsymmetric_hessian_keeps_the_null_space_path(capi, oracle):
    rng = np.random.default_rng(503)
    q = banded_qp(rng, 30, 20)
    H = np.zeros((30, 30))
    for c in range(30):
        H[q.H_ir[q.H_jc[c]:q.H_jc[c + 1]], c] = q.H_val[q.H_jc[c]:q.H_jc[c + 1]]
    Hi = H.copy(); Hi[5, 5] = -1.0                                        # indefinite
    Hu = H.copy(); Hu[3, 4] += 0.1                                        # not symmetric
    for Hm, convex in ((Hi, False), (Hu, False)):
        q2 = QPData(q.nV, q.nC, *dense_to_csc(Hm), q.A_jc, q.A_ir, q.A_val, q.g, q.lb, q.ub, q.lbA, q.ubA)
        s = load(capi, q2)
        s.solve(capi.MODE_COLD, 5000)
        assert s.large_path() == 0


def test_degenerate_inputs_on_the_general_path(capi, oracle):
    """duplicate rows, constraints parallel to bounds, dependent equalities: the exchange rule on the general path"""
    rng = np.random.default_rng(504)
    bad = 0
    for trial in range(24):
        q = banded_qp(rng, int(rng.integers(12, 40)), int(rng.integers(8, 30)), density=0.4)
        A = np.zeros((q.nC, q.nV))
        for c in range(q.nV):
            A[q.A_ir[q.A_jc[c]:q.A_jc[c + 1]], c] = q.A_val[q.A_jc[c]:q.A_jc[c + 1]]
        k = trial % 3
        if k == 0 and q.nC > 2:
            A[1] = A[0]; q.lbA[1] = q.lbA[0]; q.ubA[1] = q.ubA[0]        # duplicate row
        elif k == 1:
            A[0] = 0.0; A[0, 2] = 1.0; q.lbA[0] = q.lb[2]; q.ubA[0] = q.ub[2]   # constraint parallel to a bound
        else:
            A[2] = A[0] + A[1]; q.lbA[2] = q.lbA[0] + q.lbA[1]; q.ubA[2] = q.ubA[0] + q.ubA[1]
        q2 = QPData(q.nV, q.nC, q.H_jc, q.H_ir, q.H_val, *dense_to_csc(A), q.g, q.lb, q.ub, q.lbA, q.ubA)
        s = load(capi, q2)
        n = s.solve(capi.MODE_COLD, 5000)
        assert s.large_path() == 2
        qp, rc, n_or = oracle_cold(oracle, q2, 5000)
        assert s.status == qp.exitflag()
        if s.status != 20:
            continue
        ok, st, _, _ = s.test_optimality()
        assert ok and st.KKT_error < 1e-8
        assert abs(s.objective - qp.objective) <= 1e-8 * max(1.0, abs(qp.objective))
        wb, wc = s.working_set_raw()
        bad += not (np.array_equal(wb, qp.ws_bounds) and np.array_equal(wc, qp.ws_constraints))
    assert bad <= 3          # (exact ties may be broken by the last bits of differently ordered sums: DESIGN 5)


def test_band5_sequence_reference_rule_matches_oracle_at_2500(capi):
    """BASELINE configs[3] with SURVEY 8(d)'s "optional 5-band SPD" Hessian at n = 2 500 through rsqp_optimize_qp under the reference's
    re-initialisation rule: cold, FIXED, VARIED (= flip), FIXED, VARIED against the committed oracle answers
    (tests/golden/make_sequence_golden.py 2500 4 5)"""
    path = os.path.join(GOLDEN, "oracle_sparse_band5_sequence_2500_reference_rule.json")
    gold = json.load(open(path))
    n = gold["n"]
    q = problems.sparse_qp(n, 2 * n, 20 * n, band=5)
    s = load(capi, q, engine=0, nWSR=400000)
    steps = [(q, False)] + list(problems.sparse_sequence(q, nsteps=len(gold["steps"]) - 1))
    for (qk, changed), gs in zip(steps, gold["steps"]):
        if changed:
            s.set_A_csc(qk.A_jc, qk.A_ir, qk.A_val)
        for w, v in zip(range(5), (qk.g, qk.lb, qk.ub, qk.lbA, qk.ubA)):
            s.set_vector(w, v)
        used = s.optimize_qp()
        assert s.large_path() == 2
        assert s.status == gs["exitflag"] == 20 and used == gs["nWSR"], (gs["mode"], used, gs["nWSR"])
        wb, wc = s.working_set_raw()
        assert np.array_equal(wb, np.array(gs["ws_b"])) and np.array_equal(wc, np.array(gs["ws_c"]))
        x, y = np.array(gs["x"]), np.array(gs["y"])
        assert np.abs(s.x - x).max() <= 1e-9 * max(1.0, np.abs(x).max())
        assert np.abs(s.y - y).max() <= 1e-9 * max(1.0, np.abs(y).max())
        ok, st, _, _ = s.test_optimality()
        assert ok


def test_infeasible_qp_is_reported_infeasible_on_every_path(capi, oracle, monkeypatch):
    """A 142 x 208 QP whose perturbed limits are inconsistent (found by tests/checks/fuzz_large_vs_oracle.py 14: HiGHS confirms the
    infeasibility): on the way to the vertex where the homotopy stops, the multipliers grow without bound and the explicit inverse
    of a range-space path loses its pivots -- the engine then lets the null-space path take the solve over (RsqpLargeEngine::
    solve) and must end with the oracle's verdict, 22, from a hot start and from a cold start, whatever formulation began."""
    rng = np.random.default_rng(14)
    for k in range(11):
        nV = int(rng.integers(50, 160)); nC = int(rng.integers(30, 220))
        q = problems.random_qp(rng, nV, nC, density=float(rng.choice([0.05, 0.3, 1.0])))
        q2 = problems.perturb(rng, q, 0.03)
        if k < 10:
            rng.normal(size=q2.A_val.shape); problems.perturb(rng, q2, 0.03)
    assert (q.nV, q.nC) == (142, 208)
    qp = oracle.OracleQP(q.nV, q.nC); qp.set_A_csc(q.A_jc, q.A_ir, q.A_val); qp.set_H_csc(q.H_jc, q.H_ir, q.H_val)
    qp.init(q.g, q.lb, q.ub, q.lbA, q.ubA, 100000)
    qp.hotstart(q2.g, q2.lb, q2.ub, q2.lbA, q2.ubA, 100000)
    assert qp.exitflag() == 22
    for knob in (None, "RSQP_LARGE_NO_TABLEAU", "RSQP_LARGE_NO_LAZY", "RSQP_LARGE_NO_RSH"):
        if knob:
            monkeypatch.setenv(knob, "1")
        s = load(capi, q)
        s.solve(capi.MODE_COLD, 100000)
        assert s.status == 20
        for w, v in zip(range(5), (q2.g, q2.lb, q2.ub, q2.lbA, q2.ubA)):
            s.set_vector(w, v)
        s.solve(capi.MODE_HOT_VECTORS, 100000)
        assert s.status == 22, (knob, s.status)
        t = load(capi, q2)
        t.solve(capi.MODE_COLD, 100000)
        assert t.status == 22, (knob, t.status)
        if knob:
            monkeypatch.delenv(knob)
