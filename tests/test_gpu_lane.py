"""The lane-per-problem kernel (csrc/qp_lane.hip) against the CPU oracle and against the 8-lanes-per-problem kernel it relieves.

It serves cold starts of one-pattern batches of at most 8 x 2 (bench.py's headline workload), with or without the state written back; by default
only for more than 16 384 members -- RSQP_LANE=1 (read per batch) sends every eligible batch to it. Bar as everywhere: working sets,
statuses and iteration counts exact, x / y / objective within 1e-9 relative."""
import numpy as np
import pytest

from conftest import oracle_cold
from restartsqp_amd import problems
from test_gpu_parity import assert_same_solution

pytestmark = pytest.mark.gpu


def solve_cold(capi, probs, lane, monkeypatch):
    monkeypatch.setenv("RSQP_LANE", lane)            # (read by rsqp_batch_create)
    b = capi.Batch(probs)
    b.set_keep_state(False)
    b.solve(capi.MODE_COLD, 1000)
    assert b.last_kernel() == (2 if lane != "0" else 1)
    res = b.results()
    ok, kkt = b.test_optimality()
    b.close()
    return res, ok, kkt


def one_pattern_batch(rng, nV, nC, n, free=False, density=0.7, rel=0.05):
    base = problems.random_qp(rng, nV, nC, density=density)
    if free:
        base.lb[::2] = -np.inf; base.ub[1::3] = np.inf
        base.lb[1] = -np.inf; base.ub[1] = np.inf                  # a variable with no bound at all: it enters S in the set-up
    out = []
    for _ in range(n):
        q = problems.perturb(rng, base, rel)
        q.A_val = q.A_val * (1.0 + rel * rng.normal(size=q.A_val.shape))
        out.append(q)
    return out


def test_hs071_batch_on_the_lane_kernel(capi, oracle, monkeypatch):
    probs = problems.hs071_scale_batch(1000)
    res, ok, kkt = solve_cold(capi, probs, "1", monkeypatch)
    for q, r, o in zip(probs, res, ok):
        qp, rc, n = oracle_cold(oracle, q)
        assert_same_solution(qp, r, n)
        assert rc == 0 and o == 1 and r["nWSR"] == 2
        assert abs(r["obj"] - qp.objective) <= 1e-9 * max(1.0, abs(qp.objective))


@pytest.mark.parametrize("shape", [(8, 2), (8, 1), (8, 0), (5, 2), (3, 1), (1, 2), (2, 0)])
@pytest.mark.parametrize("free", [False, True])
def test_one_pattern_batches_match_the_oracle_and_the_eight_lane_kernel(capi, oracle, monkeypatch, shape, free):
    nV, nC = shape
    rng = np.random.default_rng(1000 + 10 * nV + nC + (100 if free else 0))
    probs = one_pattern_batch(rng, nV, nC, 200, free=free and nV >= 2)
    lane, ok, kkt = solve_cold(capi, probs, "1", monkeypatch)
    tiny, _, _ = solve_cold(capi, probs, "0", monkeypatch)
    for q, r, t, o, k in zip(probs, lane, tiny, ok, kkt):
        qp, rc, n = oracle_cold(oracle, q)
        assert_same_solution(qp, r, n)
        assert r["status"] == t["status"] and r["nWSR"] == t["nWSR"] and np.array_equal(r["ws_b"], t["ws_b"]) and np.array_equal(r["ws_c"], t["ws_c"])
        assert np.abs(r["x"] - t["x"]).max() <= 1e-9 * max(1.0, np.abs(t["x"]).max())
        if rc == 0:
            assert o == 1 and k < 1e-9
            assert abs(r["obj"] - qp.objective) <= 1e-9 * max(1.0, abs(qp.objective))


def test_infeasible_unbounded_and_inconsistent_members(capi, oracle, monkeypatch):
    """Members of one pattern that end differently: solved, infeasible constraints, inconsistent bounds (lb > ub), a ragged tail
    (the batch size is no multiple of 64)."""
    rng = np.random.default_rng(4242)
    probs = one_pattern_batch(rng, 6, 2, 131)
    for k in range(0, 131, 7):
        probs[k].lbA = probs[k].ubA + 50.0; probs[k].ubA = probs[k].lbA + 1.0          # far from the box: infeasible
    for k in range(3, 131, 11):
        probs[k].lb[2] = probs[k].ub[2] + 1.0                                            # inconsistent bounds
    lane, _, _ = solve_cold(capi, probs, "1", monkeypatch)
    tiny, _, _ = solve_cold(capi, probs, "0", monkeypatch)
    seen = set()
    for q, r, t in zip(probs, lane, tiny):
        qp, rc, n = oracle_cold(oracle, q)
        seen.add(r["status"])
        assert r["status"] == qp.exitflag() == t["status"]
        if rc == 0:
            assert_same_solution(qp, r, n)
    assert len(seen) >= 2


def test_default_threshold_and_the_calls_the_lane_kernel_does_not_take(capi, oracle, monkeypatch):
    """Default: batches of at most 16 384 members stay on the 8-lane kernel; a hot start that follows a cold start which kept nothing
    runs cold, as the handle promises; a warm re-initialisation from (x0, y0) goes to the 8-lane kernel."""
    monkeypatch.setenv("RSQP_LANE", "1")
    rng = np.random.default_rng(9)
    probs = one_pattern_batch(rng, 8, 2, 70)
    b = capi.Batch(probs)
    b.set_keep_state(False)
    b.solve(capi.MODE_COLD, 1000)
    assert b.last_kernel() == 2
    p2 = [problems.perturb(rng, q, 0.05) for q in probs]
    b.set_vectors_from(p2)
    b.solve(capi.MODE_HOT_VECTORS, 1000)              # no state was kept: a cold start on the new vectors
    assert b.last_kernel() == 2
    for q, r in zip(p2, b.results()):
        qp, rc, n = oracle_cold(oracle, q)
        assert_same_solution(qp, r, n)
    b.close()
    monkeypatch.delenv("RSQP_LANE")
    b = capi.Batch(probs)
    b.set_keep_state(False)
    b.solve(capi.MODE_COLD, 1000)
    assert b.last_kernel() == 1                       # 70 members: below the default threshold
    b.close()


@pytest.mark.parametrize("shape", [(8, 2), (6, 1), (4, 2), (8, 0)])
def test_hot_starts_continue_from_the_state_the_lane_kernel_wrote(capi, oracle, monkeypatch, shape):
    """A batch that keeps its state: the cold start runs on the lane kernel, which writes every member's state block in the layout of
    the 8-lane kernel; the hot starts that follow (new vectors, new matrices, new vectors) run on the 8-lane kernel FROM that state.
    Every member against the oracle's init / hotstart / hotstart(H, g, A, ...) sequence: working sets, statuses and nWSR exact --
    a hot start from a wrong tableau or slot state would take another path."""
    monkeypatch.setenv("RSQP_LANE", "1")
    nV, nC = shape
    rng = np.random.default_rng(300 + 10 * nV + nC)
    probs = one_pattern_batch(rng, nV, nC, 150, free=(nV == 6))
    b = capi.Batch(probs)
    b.solve(capi.MODE_COLD, 1000)
    assert b.last_kernel() == 2
    orcs = []
    for q, r in zip(probs, b.results()):
        qp, rc, n = oracle_cold(oracle, q)
        assert_same_solution(qp, r, n)
        orcs.append(qp)
    cur = probs
    hot_changes = 0
    for step in range(3):
        nxt = [problems.perturb(rng, q, 0.3) for q in cur]
        new_matrices = step == 1
        if new_matrices:
            for q in nxt:
                q.A_val = q.A_val * (1.0 + 0.02 * rng.normal(size=q.A_val.shape))
        b.set_vectors_from(nxt)
        if new_matrices:
            b.set_matrix_values(np.concatenate([q.A_val for q in nxt]), np.concatenate([q.H_val for q in nxt]))
        b.solve(capi.MODE_HOT_MATRICES if new_matrices else capi.MODE_HOT_VECTORS, 1000)
        assert b.last_kernel() == 1
        for q, qp, r in zip(nxt, orcs, b.results()):
            if new_matrices:
                qp.set_A_csc(q.A_jc, q.A_ir, q.A_val); qp.set_H_csc(q.H_jc, q.H_ir, q.H_val)
                rc, n = qp.hotstart_matrices(q.g, q.lb, q.ub, q.lbA, q.ubA, 1000)
            else:
                rc, n = qp.hotstart(q.g, q.lb, q.ub, q.lbA, q.ubA, 1000)
            assert_same_solution(qp, r, n)
            hot_changes += r["nWSR"]
        cur = nxt
    assert hot_changes > 0
    b.close()


def test_a_sequence_of_calls_gives_the_same_answers_with_and_without_the_lane_kernel(capi, oracle, monkeypatch):
    """cold start (state kept), hot start on new vectors, hot start with new matrices: the same answers, state-dependent results
    included (nWSR of every hot start), whether the cold start ran on the lane kernel or on the 8-lane kernel."""
    rng = np.random.default_rng(77)
    probs = one_pattern_batch(rng, 8, 2, 90)
    p2 = [problems.perturb(rng, q, 0.3) for q in probs]
    p3 = [problems.perturb(rng, q, 0.3) for q in p2]
    out = {}
    for lane in ("0", "1"):
        monkeypatch.setenv("RSQP_LANE", lane)
        b = capi.Batch(probs)
        b.solve(capi.MODE_COLD, 1000)
        assert b.last_kernel() == (2 if lane == "1" else 1)
        seq = [b.results()]
        b.set_vectors_from(p2); b.solve(capi.MODE_HOT_VECTORS, 1000); seq.append(b.results())
        b.set_vectors_from(p3); b.solve(capi.MODE_HOT_MATRICES, 1000); seq.append(b.results())
        assert b.last_kernel() == 1
        out[lane] = seq
        b.close()
    for ra, rb in zip(out["0"], out["1"]):
        for a, c in zip(ra, rb):
            assert a["status"] == c["status"] and a["nWSR"] == c["nWSR"] and np.array_equal(a["ws_b"], c["ws_b"]) and np.array_equal(a["ws_c"], c["ws_c"])
            assert np.abs(a["x"] - c["x"]).max() <= 1e-9 * max(1.0, np.abs(a["x"]).max())


@pytest.mark.parametrize("shape", [(8, 2), (5, 1), (7, 0)])
def test_one_shape_with_patterns_of_their_own(capi, oracle, monkeypatch, shape):
    """Members of one SHAPE whose sparsity patterns differ (three base problems of different density, interleaved member by member;
    entry counts differ as well): the vectors still travel through the wave's block, every lane walks the CSC arrays of its own
    problem. Cold start with the state kept, then a hot start on new vectors (8-lane kernel, from that state): the oracle's sequence."""
    monkeypatch.setenv("RSQP_LANE", "1")
    nV, nC = shape
    rng = np.random.default_rng(900 + 10 * nV + nC)
    bases = [problems.random_qp(rng, nV, nC, density=d) for d in (0.3, 0.6, 1.0)]
    # different Hessian patterns too: thin out the off-diagonal part of the first two (symmetrically)
    for k, b0 in enumerate(bases[:2]):
        H = b0.dense_H()
        M = np.triu(rng.random((nV, nV)) < (0.3 + 0.3 * k), 1)
        H = H * (M | M.T | np.eye(nV, dtype=bool))
        H += np.diag(np.abs(H).sum(axis=1))                      # (stays positive definite)
        from restartsqp_amd.qpdump import dense_to_csc
        b0.H_jc, b0.H_ir, b0.H_val = dense_to_csc(H)
    probs = []
    for k in range(201):
        q = problems.perturb(rng, bases[k % 3], 0.05)
        q.A_val = q.A_val * (1.0 + 0.05 * rng.normal(size=q.A_val.shape))
        probs.append(q)
    assert len({(len(q.A_val), len(q.H_val)) for q in probs}) >= 2
    b = capi.Batch(probs)
    b.solve(capi.MODE_COLD, 1000)
    assert b.last_kernel() == 2
    orcs = []
    for q, r in zip(probs, b.results()):
        qp, rc, n = oracle_cold(oracle, q)
        assert_same_solution(qp, r, n)
        orcs.append(qp)
    nxt = [problems.perturb(rng, q, 0.3) for q in probs]
    b.set_vectors_from(nxt)
    b.solve(capi.MODE_HOT_VECTORS, 1000)
    assert b.last_kernel() == 1
    for q, qp, r in zip(nxt, orcs, b.results()):
        rc, n = qp.hotstart(q.g, q.lb, q.ub, q.lbA, q.ubA, 1000)
        assert_same_solution(qp, r, n)
    b.close()


def test_members_that_take_different_paths(capi, oracle, monkeypatch):
    """The QPs of the hs071 SQP trajectory that share one sparsity pattern (tests/golden/sqp_traces.json: 5 of its 6 iterates), each
    with seeded perturbations, interleaved: neighbouring lanes of a wave take different paths of 5 changes each (entering and leaving
    bounds and constraints, exchanges) -- the wave executes their union, every lane must still end where the oracle does."""
    import json, os
    from collections import Counter
    from conftest import GOLDEN
    tr = json.load(open(os.path.join(GOLDEN, "sqp_traces.json")))["hs071"]["qps"]
    base = [problems.handler_qp(problems.hs071_nlp(np.array(g["x"]), np.array(g["lam"])), delta=g["delta"], rho=g["rho"]) for g in tr]
    key = lambda q: (tuple(q.A_jc), tuple(q.A_ir), tuple(q.H_jc), tuple(q.H_ir))
    best = Counter(key(q) for q in base).most_common(1)[0][0]
    base = [q for q in base if key(q) == best]
    assert len(base) >= 3
    rng = np.random.default_rng(20260104)
    probs = [problems.perturb(rng, base[k % len(base)]) for k in range(640)]
    res, ok, kkt = solve_cold(capi, probs, "1", monkeypatch)
    paths = set()
    for q, r, o in zip(probs, res, ok):
        qp, rc, n = oracle_cold(oracle, q)
        assert_same_solution(qp, r, n)
        assert rc == 0 and o == 1
        paths.add((r["nWSR"], tuple(r["ws_b"]), tuple(r["ws_c"])))
    assert len(paths) >= 2


@pytest.mark.parametrize("limit", [0, 1, 2])
def test_iteration_limit(capi, oracle, monkeypatch, limit):
    """nWSR limits below / at what the members need (the hs071 QP takes 2 changes): status, return code and the iterate where the
    homotopy was stopped as the oracle has them; the 8-lane kernel agrees."""
    monkeypatch.setenv("RSQP_LANE", "1")
    probs = problems.hs071_scale_batch(100)
    b = capi.Batch(probs)
    b.set_keep_state(False)
    b.solve(capi.MODE_COLD, limit)
    assert b.last_kernel() == 2
    lane = b.results()
    b.close()
    monkeypatch.setenv("RSQP_LANE", "0")
    b = capi.Batch(probs)
    b.set_keep_state(False)
    b.solve(capi.MODE_COLD, limit)
    tiny = b.results()
    b.close()
    for q, r, t in zip(probs, lane, tiny):
        qp, rc, n = oracle_cold(oracle, q, nWSR=limit)
        assert r["status"] == qp.exitflag() == t["status"] and r["nWSR"] == n == t["nWSR"]
        assert (r["status"] == 20) == (limit >= 2)
        assert np.array_equal(qp.ws_bounds, r["ws_b"]) and np.array_equal(qp.ws_constraints, r["ws_c"])
        assert np.abs(qp.x - r["x"]).max() <= 1e-9 * max(1.0, np.abs(qp.x).max())
