"""HBM-resident engine (qp_large.hip: Householder / explicit-inverse null-space updates; for diagonal positive
Hessians -- every sparse configuration below -- the range-space path with the explicit inverse of the Schur
complement) against the oracle -- same bar as the LDS-resident kernel: working sets, status and nWSR bit-exact,
x / y to 1e-9 relative -- and BASELINE configs 3 and 4 at full size through the reference's KKT certificate."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, dump_paths, oracle_cold
from restartsqp_amd import problems
from restartsqp_amd.qpdump import QPData, dense_to_csc, read_qore_dump

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["default_path", "null_space_path"])
def engine_path(request, monkeypatch):
    """Every test of this file runs twice: with the engine's own choice of formulation (a positive definite Hessian takes the
    general range-space path of DESIGN 4.5 since round 5) and with that path switched off (RSQP_LARGE_NO_RSH=1: the null-space
    path -- or, for a diagonal Hessian, the range-space path of DESIGN 4.4 -- as in rounds 1-4)."""
    if request.param == "null_space_path":
        monkeypatch.setenv("RSQP_LARGE_NO_RSH", "1")
    return request.param


def load(capi, q, engine=2, nWSR=100000):
    s = capi.Solver(q.nV, q.nC)
    s.set_engine(engine)
    s.set_options(nWSR, 100)
    s.set_A_csc(q.A_jc, q.A_ir, q.A_val); s.set_H_csc(q.H_jc, q.H_ir, q.H_val)
    for w, v in zip(range(5), (q.g, q.lb, q.ub, q.lbA, q.ubA)):
        s.set_vector(w, v)
    return s


def same_as_oracle(s, n, qp, n_or, check_nwsr=True):
    wb, wc = s.working_set_raw()
    assert s.status == qp.exitflag()
    assert np.array_equal(wb, qp.ws_bounds) and np.array_equal(wc, qp.ws_constraints)
    if check_nwsr:
        assert n == n_or
    assert np.abs(s.x - qp.x).max() <= 1e-9 * max(1.0, np.abs(qp.x).max())
    assert np.abs(s.y - qp.y).max() <= 1e-9 * max(1.0, np.abs(qp.y).max())


def test_random_convex_against_oracle(capi, oracle):
    rng = np.random.default_rng(41)
    cases = [problems.hs071_first_qp()]
    cases += [problems.random_qp(rng, int(rng.integers(1, 40)), int(rng.integers(0, 45))) for _ in range(14)]
    cases += [problems.random_qp(rng, 90, 60, 0.3), problems.random_qp(rng, 50, 130, 0.3)]
    for q in cases:
        s = load(capi, q)
        n = s.solve(capi.MODE_COLD, 5000)
        qp, rc, n_or = oracle_cold(oracle, q, 5000)
        same_as_oracle(s, n, qp, n_or)
        ok, st, _, _ = s.test_optimality()
        assert ok and st.KKT_error < 1e-9


def test_edge_cases(capi, oracle):
    rng = np.random.default_rng(43)
    cases = [problems.random_qp(rng, 6, 0), problems.random_qp(rng, 1, 3)]
    q = problems.random_qp(rng, 7, 4); q.lb[:] = -np.inf; q.ub[:] = 1e20; cases.append(q)        # free variables
    q = problems.random_qp(rng, 9, 3); q.ubA[:] = q.lbA; cases.append(q)                           # equalities
    A = np.array([[1.0, 1.0]])
    cases.append(QPData(2, 1, *dense_to_csc(np.eye(2)), *dense_to_csc(A), np.zeros(2), -np.ones(2), np.ones(2),
                        np.array([3.0]), np.array([np.inf])))                                      # infeasible
    H = np.diag([1.0, -1.0])
    cases.append(QPData(2, 0, *dense_to_csc(H), *dense_to_csc(np.zeros((0, 2))), np.array([0.0, -1.0]),
                        np.array([-1.0, 0.0]), np.array([1.0, np.inf]), np.zeros(0), np.zeros(0)))  # unbounded
    for q in cases:
        s = load(capi, q)
        n = s.solve(capi.MODE_COLD, 1000)
        qp, rc, n_or = oracle_cold(oracle, q)
        same_as_oracle(s, n, qp, n_or)
    q = problems.random_qp(np.random.default_rng(3), 30, 40)
    s = load(capi, q)
    n = s.solve(capi.MODE_COLD, 3)
    qp, rc, n_or = oracle_cold(oracle, q, nWSR=3)
    assert n == n_or == 3 and s.status == qp.exitflag() == 28


def test_nonconvex_dumps(capi, oracle):
    """Flipping bounds on the reference's (indefinite) dumps. Without a flip the run is compared
    like a convex one. With flips the homotopy path of a non-convex QP is not unique under
    rounding (an equality held "at its lower side" by one run is held "at its upper side" by the
    other), so the comparison is: same primal point, same objective, same set of active
    quantities, and the reference's certificate with the same verdict."""
    from conftest import oracle_certificate
    for p in dump_paths():
        q = read_qore_dump(p)
        s = load(capi, q)
        n = s.solve(capi.MODE_COLD, 1000)
        qp, rc, n_or = oracle_cold(oracle, q)
        if qp.nflips() == 0:
            same_as_oracle(s, n, qp, n_or)
            continue
        wb, wc = s.working_set_raw()
        assert s.status == qp.exitflag() == 20
        assert np.abs(s.x - qp.x).max() <= 1e-9 * max(1.0, np.abs(qp.x).max())
        assert abs(s.objective - qp.objective) <= 1e-9 * max(1.0, abs(qp.objective))
        eq_b, eq_c = q.lb == q.ub, q.lbA == q.ubA
        assert np.array_equal(wb[~eq_b], qp.ws_bounds[~eq_b]) and np.array_equal(wc[~eq_c], qp.ws_constraints[~eq_c])
        assert np.array_equal(wb != 0, qp.ws_bounds != 0) and np.array_equal(wc != 0, qp.ws_constraints != 0)
        ok, st, _, _ = s.test_optimality()
        ok_o, st_o, _, _ = oracle_certificate(oracle, q, qp.x, qp.y, qp.ws_bounds, qp.ws_constraints)
        scale = max(1.0, np.abs(q.g).max(), np.abs(q.H_val).max() * max(1.0, np.abs(qp.x).max()), np.abs(qp.y).max())
        assert abs(st.KKT_error - st_o.KKT_error) <= 1e-12 * scale


def test_hot_start_modes(capi, oracle):
    rng = np.random.default_rng(47)
    for trial in range(4):
        q = problems.random_qp(rng, int(rng.integers(8, 40)), int(rng.integers(3, 35)))
        s = load(capi, q)
        n = s.solve(capi.MODE_COLD, 5000)
        qp, rc, n_or = oracle_cold(oracle, q, 5000)
        same_as_oracle(s, n, qp, n_or)
        q2 = problems.perturb(rng, q, 0.05)                      # hotstart(g, lb, ub, lbA, ubA)
        for w, v in zip(range(5), (q2.g, q2.lb, q2.ub, q2.lbA, q2.ubA)):
            s.set_vector(w, v)
        n = s.solve(capi.MODE_HOT_VECTORS, 5000)
        rc, n_or = qp.hotstart(q2.g, q2.lb, q2.ub, q2.lbA, q2.ubA, 5000)
        same_as_oracle(s, n, qp, n_or)
        A2 = q2.A_val * (1.0 + 0.01 * rng.normal(size=q2.A_val.shape))   # hotstart(H, g, A, ...)
        s.set_A_csc(q2.A_jc, q2.A_ir, A2); s.set_H_csc(q2.H_jc, q2.H_ir, q2.H_val * 1.05)
        n = s.solve(capi.MODE_HOT_MATRICES, 5000)
        qp.set_A_csc(q2.A_jc, q2.A_ir, A2); qp.set_H_csc(q2.H_jc, q2.H_ir, q2.H_val * 1.05)
        rc, n_or = qp.hotstart_matrices(q2.g, q2.lb, q2.ub, q2.lbA, q2.ubA, 5000)
        same_as_oracle(s, n, qp, n_or)
        x0, y0, gb = s.x, s.y, s.working_set_raw()[0]               # init(..., x0, y0, bounds)
        q3 = problems.perturb(rng, q2, 0.05)
        for w, v in zip(range(5), (q3.g, q3.lb, q3.ub, q3.lbA, q3.ubA)):
            s.set_vector(w, v)
        rule = bool(trial & 1)                      # the reference's rule (default) and the opt-in sign(y0) rule in turn
        s.set_reinit_guess(rule); qp.set_guess_constraints_from_y0(rule)
        n = s.solve(capi.MODE_WARM_REINIT, 5000, x0, y0, gb)
        rc, n_or = qp.init(q3.g, q3.lb, q3.ub, q3.lbA, q3.ubA, 5000, x0=x0, y0=y0, guess_b=gb)
        same_as_oracle(s, n, qp, n_or)


def test_deferred_reflection_kernels_match_oracle(capi, oracle, monkeypatch):
    """An incoming constraint's reflection of Z and the shrinking of Wz ride on the step direction's products (k_ger_gemv_t,
    k_wz_shrink_gemv); the second one only from 3 072 null-space columns on. Forced on for every size here (the engine reads
    the knob when a solver is created), and switched off altogether: the same decisions as the oracle either way, cold and
    hot start. Third knob: the range-space part of the step direction recomputed at every step instead of carried over an
    added constraint (k_carry_wY / k_carry_xY, the default)."""
    rng = np.random.default_rng(4711)
    cases = [problems.random_qp(rng, 64, 40, 0.5), problems.random_qp(rng, 150, 120, 0.3), problems.random_qp(rng, 96, 200, 0.4)]
    # (RSQP_LARGE_WZ_SYM_MIN=0: Wz in its upper triangle only, the storage of problems from 4 096 variables on, forced for every size)
    for knob, val in (("RSQP_LARGE_FUSE_WZ_MIN", "0"), ("RSQP_LARGE_NO_FUSE", "1"), ("RSQP_LARGE_NO_CARRY", "1"),
                      ("RSQP_LARGE_NO_CARRY_NULL", "1"), ("RSQP_LARGE_WZ_SYM_MIN", "0")):
        monkeypatch.setenv(knob, val)
        for q in cases:
            s = load(capi, q)
            n = s.solve(capi.MODE_COLD, 20000)
            qp, rc, n_or = oracle_cold(oracle, q, 20000)
            same_as_oracle(s, n, qp, n_or)
            q2 = problems.perturb(np.random.default_rng(5), q)
            for w, v in zip(range(5), (q2.g, q2.lb, q2.ub, q2.lbA, q2.ubA)):
                s.set_vector(w, v)
            n = s.solve(capi.MODE_HOT_VECTORS, 20000)
            rc, n_or = qp.hotstart(q2.g, q2.lb, q2.ub, q2.lbA, q2.ubA, 20000)
            same_as_oracle(s, n, qp, n_or)
            s.close()
        monkeypatch.delenv(knob)


def _diag_h_qp(rng, nV, nC, density, hmin=0.5):
    """random_qp with a DIAGONAL Hessian stored as one entry per column (what the engine's range-space path is for)."""
    q = problems.random_qp(rng, nV, nC, density)
    h = hmin + np.abs(rng.normal(size=nV))
    return QPData(nV, nC, np.arange(nV + 1, dtype=np.int32), np.arange(nV, dtype=np.int32), h, q.A_jc, q.A_ir, q.A_val, q.g, q.lb, q.ub,
                  q.lbA, q.ubA, name="diagH_%dx%d" % (nV, nC))


@pytest.mark.parametrize("knob", [None, "RSQP_LARGE_NO_DUAL", "RSQP_LARGE_NO_CARRY", "RSQP_LARGE_NO_FUSE", "RSQP_NO_BLOCKED_SETUP", "RSQP_LARGE_NO_SYM", "RSQP_LARGE_NO_LAZY"])
def test_range_space_path_all_call_shapes(capi, oracle, monkeypatch, knob):
    """Diagonal positive Hessian: the HBM-resident engine keeps the explicit inverse of A_AC,FR D^-1 A_AC,FR' instead of the null-space
    factors (qp_large.hip, Impl::dual). Every call shape against the oracle -- cold, hot start on new vectors, hot start with new
    matrices (blocked Gram + Cholesky set-up from 32 active constraints on), warm re-initialisation under both rules -- with the
    path's knobs in turn: switched off (null-space path: the same answers), multiplier step exact at every change, rank-1 updates
    not deferred, sequential set-up."""
    if knob:
        monkeypatch.setenv(knob, "1")
    rng = np.random.default_rng(8800)
    for trial, (nV, nC, dens) in enumerate(((60, 90, 0.4), (150, 260, 0.25), (96, 40, 0.6))):
        q = _diag_h_qp(rng, nV, nC, dens)
        s = load(capi, q)
        n = s.solve(capi.MODE_COLD, 20000)
        qp, rc, n_or = oracle_cold(oracle, q, 20000)
        same_as_oracle(s, n, qp, n_or)
        q2 = problems.perturb(rng, q, 0.05)
        for w, v in zip(range(5), (q2.g, q2.lb, q2.ub, q2.lbA, q2.ubA)):
            s.set_vector(w, v)
        n = s.solve(capi.MODE_HOT_VECTORS, 20000)
        rc, n_or = qp.hotstart(q2.g, q2.lb, q2.ub, q2.lbA, q2.ubA, 20000)
        same_as_oracle(s, n, qp, n_or)
        A2 = q2.A_val * (1.0 + 0.01 * rng.normal(size=q2.A_val.shape))
        s.set_A_csc(q2.A_jc, q2.A_ir, A2); s.set_H_csc(q2.H_jc, q2.H_ir, q2.H_val * 1.05)
        n = s.solve(capi.MODE_HOT_MATRICES, 20000)
        qp.set_A_csc(q2.A_jc, q2.A_ir, A2); qp.set_H_csc(q2.H_jc, q2.H_ir, q2.H_val * 1.05)
        rc, n_or = qp.hotstart_matrices(q2.g, q2.lb, q2.ub, q2.lbA, q2.ubA, 20000)
        same_as_oracle(s, n, qp, n_or)
        if knob is None and trial == 1:
            # guard against a refactoring that silently switches the path off: the blocked set-up of this hot start was the
            # range-space one (Gram matrix + Cholesky), not QR + Q + R^-1
            sp = s.setup_profile()
            assert sp is not None and sp["range_space"], sp
        x0, y0, gb = s.x, s.y, s.working_set_raw()[0]
        q3 = problems.perturb(rng, q2, 0.05)
        for w, v in zip(range(5), (q3.g, q3.lb, q3.ub, q3.lbA, q3.ubA)):
            s.set_vector(w, v)
        rule = bool(trial & 1)
        s.set_reinit_guess(rule); qp.set_guess_constraints_from_y0(rule)
        n = s.solve(capi.MODE_WARM_REINIT, 20000, x0, y0, gb)
        rc, n_or = qp.init(q3.g, q3.lb, q3.ub, q3.lbA, q3.ubA, 20000, x0=x0, y0=y0, guess_b=gb)
        same_as_oracle(s, n, qp, n_or)
        s.close()


def test_range_space_path_degenerate_inputs(capi, oracle):
    """Degenerate inputs (duplicate constraint, zero row, constraint parallel to a bound, integer data) with a diagonal Hessian on the
    range-space path: exchanges and the two-stage independence test decide as the oracle does -- status, nWSR, working sets identical;
    where an exact tie is broken by the last bits (at most 2 of the 60), the answer must still be a certified KKT point with the
    oracle's objective (the policy of test_degenerate_sweep_mismatch_rate_is_bounded)."""
    rng = np.random.default_rng(8802)
    other = 0
    for t in range(60):
        q0 = problems.degenerate_qp(rng, t % 4)
        h = 0.5 + np.abs(rng.normal(size=q0.nV))
        if t % 4 == 3:
            h = np.round(h) + 1.0                                   # integer data all the way
        q = QPData(q0.nV, q0.nC, np.arange(q0.nV + 1, dtype=np.int32), np.arange(q0.nV, dtype=np.int32), h, q0.A_jc, q0.A_ir, q0.A_val,
                   q0.g, q0.lb, q0.ub, q0.lbA, q0.ubA, name=q0.name)
        s = load(capi, q)
        n = s.solve(capi.MODE_COLD, 2000)
        qp, rc, n_or = oracle_cold(oracle, q, 2000)
        wb, wc = s.working_set_raw()
        same = (s.status == qp.exitflag() and n == n_or and np.array_equal(wb, qp.ws_bounds) and np.array_equal(wc, qp.ws_constraints))
        if same:
            assert np.abs(s.x - qp.x).max() <= 1e-9 * max(1.0, np.abs(qp.x).max())
        else:
            other += 1
            assert s.status == qp.exitflag(), (t, s.status, qp.exitflag())
            if s.status == 20:
                ok, st, _, _ = s.test_optimality()
                assert ok and st.KKT_error < 1e-8, (t, st.KKT_error)
                obj = 0.5 * float(s.x @ (h * s.x)) + float(q.g @ s.x)
                assert abs(obj - qp.objective) <= 1e-8 * max(1.0, abs(qp.objective)), (t, obj, qp.objective)
        s.close()
    assert other <= 2, other


def test_range_space_path_needs_a_positive_diagonal(capi, oracle):
    """A diagonal Hessian with a zero or negative entry is not the range-space path's (D^-1 does not exist / the reduced Hessian can be
    indefinite): the engine keeps the null-space path with its definiteness guard, and the answer is the oracle's."""
    rng = np.random.default_rng(8801)
    q = _diag_h_qp(rng, 40, 30, 0.5)
    for bad in (0.0, -0.3):
        h = q.H_val.copy(); h[7] = bad
        qb = QPData(q.nV, q.nC, q.H_jc, q.H_ir, h, q.A_jc, q.A_ir, q.A_val, q.g, q.lb, q.ub, q.lbA, q.ubA)
        s = load(capi, qb)
        n = s.solve(capi.MODE_COLD, 20000)
        qp, rc, n_or = oracle_cold(oracle, qb, 20000)
        same_as_oracle(s, n, qp, n_or, check_nwsr=False)
        s.close()


def test_deferred_and_carried_paths_are_taken(capi, engine_path):
    """Guard against a refactoring that silently switches the round-3 paths off: with the engine's own accounting on, a cold
    start of a mid-size problem (even leading dimension) must show launches of both fused kernels of the deferred reflections."""
    if engine_path != "null_space_path":
        pytest.skip("kernels of the null-space path")
    q = problems.random_qp(np.random.default_rng(4712), 150, 120, 0.3)
    s = load(capi, q)
    s.set_engine_profiling(True)
    s.solve(capi.MODE_COLD, 20000)
    prof = s.engine_profile()
    s.close()
    assert prof is not None
    assert prof.get("ger_gemv_n", {}).get("calls", 0) > 0        # reflection of Z riding on Z wZ (null-space part carried)
    assert prof.get("ger_gemv_t", {}).get("calls", 0) > 0        # reflection of Z riding on Z'w, or of Y / Minv behind a removal


def test_mid_size_dense_matches_oracle(capi, oracle):
    q = problems.dense_qp(300, 600, seed=20260101)
    s = load(capi, q, engine=0)
    assert s.engine == 2                      # does not fit LDS: automatic choice
    n = s.solve(capi.MODE_COLD, 100000)
    qp, rc, n_or = oracle_cold(oracle, q, 100000)
    same_as_oracle(s, n, qp, n_or)


def test_baseline_dense_2048x4096_certificate(capi):
    """BASELINE config 3 at full size: cold start; checked through the reference's acceptance
    test (KKT_error <= 1e-6) and size-independent properties of the answer."""
    q = problems.dense_qp()
    s = load(capi, q, engine=0)
    n = s.solve(capi.MODE_COLD, 200000)
    ok, st, Wc, Wb = s.test_optimality()
    assert s.status == 20 and ok and st.KKT_error < 1e-8
    x, y = s.x, s.y
    wb, wc = s.working_set_raw()
    assert np.all(x >= q.lb - 1e-9) and np.all(x <= q.ub + 1e-9)
    assert np.all(y[:q.nV][wb == 0] == 0.0) and np.all(y[q.nV:][wc == 0] == 0.0)          # complementarity
    assert np.all(y[q.nV:][wc == -1] >= 0.0) and np.all(y[q.nV:][wc == 1] <= 0.0)           # dual signs
    assert (wc != 0).sum() + (wb != 0).sum() <= q.nV and n >= (wc != 0).sum()
    # the oracle's answer for the same seeded input at FULL size (tests/golden/make_oracle_golden.py --large dense:
    # 10 715 working-set changes, 27 min of CPU in the build container): same count, same working set, same point
    gold = json.load(open(os.path.join(GOLDEN, "oracle_large_dense_2048x4096.json")))
    assert gold["exitflag"] == 20 and n == gold["nWSR"]
    assert np.array_equal(wb, gold["ws_b"]) and np.array_equal(wc, gold["ws_c"])
    gx, gy = np.array(gold["x"]), np.array(gold["y"])
    assert np.abs(x - gx).max() <= 1e-9 * max(1.0, np.abs(gx).max())
    assert np.abs(y - gy).max() <= 1e-9 * max(1.0, np.abs(gy).max())
    assert abs(s.objective - gold["objective"]) <= 1e-9 * max(1.0, abs(gold["objective"]))
    # idempotence: a hot start on unchanged data takes no working-set change
    assert s.solve(capi.MODE_HOT_VECTORS, 1000) == 0 and np.array_equal(s.x, x)


def test_baseline_sparse_10k_sequence(capi):
    """BASELINE config 4 as specified: n = 10 000, m = 20 000, 200 000 Jacobian non-zeros; cold start, then the
    warm-started sequence of 50 QPs (odd steps: new vectors, even steps: new Jacobian values as well) driven through
    rsqp_optimize_qp, i.e. through the FIXED / VARIED dispatch of qpOASESInterface.cpp:137-224 (a FIXED <-> VARIED
    flip re-initialises from (x, y, bounds), :199-207); every answer carries the reference's KKT certificate.
    All 50 steps under the OPT-IN shortcut rsqp_set_reinit_guess(1) (constraint sides of the re-init from sign(y_qp));
    the reference's own rule -- the default -- at full size: test_baseline_sparse_10k_reference_rule."""
    q = problems.sparse_qp()
    s = load(capi, q, engine=0)
    s.set_options(qp_maxiter=200000)
    s.set_reinit_guess(True)
    n = s.optimize_qp()
    ok, st, _, _ = s.test_optimality()
    assert s.status == 20 and ok and st.KKT_error < 1e-8 and n > 1000
    steps = 0
    for qk, changed in problems.sparse_sequence(q, nsteps=50):
        for w, v in zip(range(5), (qk.g, qk.lb, qk.ub, qk.lbA, qk.ubA)):
            s.set_vector(w, v)
        if changed:                                   # VARIED: new Jacobian values
            s.set_A_csc(qk.A_jc, qk.A_ir, qk.A_val)
        nk = s.optimize_qp()
        ok, st, _, _ = s.test_optimality()
        assert s.status == 20 and ok and nk < n // 10, (steps, nk, st.KKT_error)
        steps += 1
    assert steps == 50


def test_baseline_sparse_10k_reference_rule(capi):
    """The same configuration under the REFERENCE's re-initialisation rule (the library default): a FIXED <-> VARIED
    flip calls init(.., x_qp, y_qp, &bounds) with no guessed constraints (qpOASESInterface.cpp:199-207), so the working
    set of the constraints is rebuilt one change at a time. Cold start + 4 steps (2 of them VARIED = flips) at full size,
    every answer to the reference's KKT certificate; the flips must take thousands of changes -- most of the active
    constraints are added again one by one (that is what the rule costs) --, the FIXED steps in between are plain hot starts."""
    q = problems.sparse_qp()
    s = load(capi, q, engine=0)
    s.set_options(qp_maxiter=200000)
    n = s.optimize_qp()
    assert s.status == 20 and n > 1000
    for k, (qk, changed) in enumerate(problems.sparse_sequence(q, nsteps=4)):
        for w, v in zip(range(5), (qk.g, qk.lb, qk.ub, qk.lbA, qk.ubA)):
            s.set_vector(w, v)
        if changed:
            s.set_A_csc(qk.A_jc, qk.A_ir, qk.A_val)
        nk = s.optimize_qp()
        ok, st, _, _ = s.test_optimality()
        nact = int((s.working_set_raw()[1] != 0).sum())
        assert s.status == 20 and ok and st.KKT_error < 1e-8, (k, nk, st.KKT_error)
        if changed:
            assert nact > 1000 and nk > nact // 2, (k, nk, nact)   # (most of) the active constraints were added again
        else:
            assert nk < n // 10, (k, nk)


def test_sparse_sequence_reference_rule_matches_oracle_at_2500(capi):
    """n = 2 500, m = 5 000 through rsqp_optimize_qp under the reference's rule against the committed oracle answers
    (tests/golden/oracle_sparse_sequence_2500_reference_rule.json, made by tests/golden/make_sequence_golden.py: the
    oracle driven by the restated dispatch oracle.OracleInterface): cold start, FIXED, VARIED (flip), FIXED, VARIED
    (flip) -- nWSR and working sets bit-exact at every step, x / y to 1e-9."""
    gold = json.load(open(os.path.join(GOLDEN, "oracle_sparse_sequence_2500_reference_rule.json")))
    q = problems.sparse_qp(2500, 5000, 50000)
    s = load(capi, q, engine=0)
    s.set_options(qp_maxiter=400000)
    seq = [(q, False)] + list(problems.sparse_sequence(q, nsteps=len(gold["steps"]) - 1))
    for k, ((qk, changed), g) in enumerate(zip(seq, gold["steps"])):
        if k > 0:
            for w, v in zip(range(5), (qk.g, qk.lb, qk.ub, qk.lbA, qk.ubA)):
                s.set_vector(w, v)
            if changed:
                s.set_A_csc(qk.A_jc, qk.A_ir, qk.A_val)
        nk = s.optimize_qp()
        wb, wc = s.working_set_raw()
        assert g["exitflag"] == s.status == 20 and nk == g["nWSR"], (k, g["mode"], nk, g["nWSR"])
        assert np.array_equal(wb, g["ws_b"]) and np.array_equal(wc, g["ws_c"]), (k, g["mode"])
        gx, gy = np.array(g["x"]), np.array(g["y"])
        assert np.abs(s.x - gx).max() <= 1e-9 * max(1.0, np.abs(gx).max())
        assert np.abs(s.y - gy).max() <= 1e-9 * max(1.0, np.abs(gy).max())
    assert [g["mode"] for g in gold["steps"]] == ["cold", "hot_vectors", "reinit", "hot_vectors", "reinit"]


def test_sparse_sequence_matches_oracle_at_2500(capi, oracle):
    """The same configuration at the largest size the oracle finishes in minutes (n = 2 500, m = 5 000, 50 000
    non-zeros): the cold start against the committed oracle answer (tests/golden/oracle_large_sparse_2500x5000.json,
    made by tests/golden/make_oracle_golden.py --large sparse 2500): nWSR and working set bit-exact, x to 1e-9.
    (Warm-started sequences against the oracle: test_hot_start_modes, test_blocked_setup_matches_oracle.)"""
    path = os.path.join(GOLDEN, "oracle_large_sparse_2500x5000.json")
    q = problems.sparse_qp(2500, 5000, 50000)
    s = load(capi, q, engine=0)
    s.set_options(qp_maxiter=200000)
    n = s.optimize_qp()
    gold = json.load(open(path))
    wb, wc = s.working_set_raw()
    assert gold["exitflag"] == 20 and n == gold["nWSR"]
    assert np.array_equal(wb, gold["ws_b"]) and np.array_equal(wc, gold["ws_c"])
    gx = np.array(gold["x"])
    assert np.abs(s.x - gx).max() <= 1e-9 * max(1.0, np.abs(gx).max())


@pytest.mark.parametrize("kind", ["dense", "sparse"])
def test_blocked_setup_matches_oracle(capi, oracle, kind):
    """A solve that starts from a large working set (hotstart with new matrices, init with
    x0 / y0 / guessed bounds) builds Y, Z, (A_AC Y)^-1 by blocked Householder QR and (Z'HZ)^-1 by
    blocked Cholesky on the matrix cores (dense_la.hip) instead of one reflection per constraint.
    Same bar as everywhere: working sets, status, nWSR identical to the oracle; x, y to 1e-9."""
    rng = np.random.default_rng(5)
    q = problems.dense_qp(300, 600, seed=11) if kind == "dense" else problems.sparse_qp(n=500, m=1000, nnz=10000, seed=12)
    s = load(capi, q, engine=2)
    n = s.solve(capi.MODE_COLD, 100000)
    qp, rc, n_or = oracle_cold(oracle, q, 100000)
    same_as_oracle(s, n, qp, n_or)
    assert (s.working_set_raw()[1] != 0).sum() >= 32          # enough active constraints for the blocked path
    q2 = problems.perturb(rng, q, 0.02)
    A2 = q2.A_val * (1.0 + 0.01 * rng.normal(size=q2.A_val.shape))
    for w, v in zip(range(5), (q2.g, q2.lb, q2.ub, q2.lbA, q2.ubA)):
        s.set_vector(w, v)
    s.set_A_csc(q2.A_jc, q2.A_ir, A2); s.set_H_csc(q2.H_jc, q2.H_ir, q2.H_val * 1.02)
    n = s.solve(capi.MODE_HOT_MATRICES, 100000)
    qp.set_A_csc(q2.A_jc, q2.A_ir, A2); qp.set_H_csc(q2.H_jc, q2.H_ir, q2.H_val * 1.02)
    rc, n_or = qp.hotstart_matrices(q2.g, q2.lb, q2.ub, q2.lbA, q2.ubA, 100000)
    same_as_oracle(s, n, qp, n_or)
    ok, st, _, _ = s.test_optimality()
    assert ok and st.KKT_error < 1e-8
    x0, y0, gb = s.x, s.y, s.working_set_raw()[0]
    q3 = problems.perturb(rng, q2, 0.02)
    for w, v in zip(range(5), (q3.g, q3.lb, q3.ub, q3.lbA, q3.ubA)):
        s.set_vector(w, v)
    s.set_reinit_guess(True)                    # the opt-in rule (sides from sign(y0)): the one that starts from a large working set
    n = s.solve(capi.MODE_WARM_REINIT, 100000, x0, y0, gb)
    qp.set_guess_constraints_from_y0(True)
    rc, n_or = qp.init(q3.g, q3.lb, q3.ub, q3.lbA, q3.ubA, 100000, x0=x0, y0=y0, guess_b=gb)
    same_as_oracle(s, n, qp, n_or)


def test_blocked_setup_falls_back_on_dependent_guess(capi, oracle):
    """Duplicate constraint rows in the guessed working set: the blocked QR reports the
    dependence and the engine rebuilds the factorisation one constraint at a time (which skips
    the dependent row) -- the result still matches the oracle."""
    rng = np.random.default_rng(8)
    q = problems.dense_qp(120, 200, seed=3)
    A = q.dense_A(); A[150:200] = A[100:150]          # 50 duplicated rows
    lbA, ubA = q.lbA.copy(), q.ubA.copy(); lbA[150:200] = lbA[100:150]; ubA[150:200] = ubA[100:150]
    jc, ir, val = dense_to_csc(A)
    q = QPData(q.nV, q.nC, q.H_jc, q.H_ir, q.H_val, jc, ir, val, q.g, q.lb, q.ub, lbA, ubA, name="dup")
    s = load(capi, q, engine=2)
    n = s.solve(capi.MODE_COLD, 100000)
    qp, rc, n_or = oracle_cold(oracle, q, 100000)
    same_as_oracle(s, n, qp, n_or)
    x0, y0, gb = s.x, s.y, s.working_set_raw()[0]
    s.set_reinit_guess(False)                    # the reference's rule: constraint sides from A x0
    n = s.solve(capi.MODE_WARM_REINIT, 100000, x0, y0, gb)    # constraints guessed from A x0: both copies look active
    rc, n_or = qp.init(q.g, q.lb, q.ub, q.lbA, q.ubA, 100000, x0=x0, y0=y0, guess_b=gb)
    same_as_oracle(s, n, qp, n_or)


def test_fuzz_against_oracle():
    """Random dense and sparse problems of 50-160 variables on the HBM-resident engine: cold start, hot
    start on vectors, hot start with new matrices (blocked set-up), warm re-initialisation -- working
    sets, status, nWSR identical to the oracle, x / y to 1e-9 (tests/checks/fuzz_large_vs_oracle.py)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "checks", "fuzz_large_vs_oracle.py"), "5", "6"],
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0 and "FUZZ OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
