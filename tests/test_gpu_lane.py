"""The lane-per-problem kernel (csrc/qp_lane.hip) against the CPU oracle and against the 8-lanes-per-problem kernel it relieves.

It serves cold starts of one-pattern batches of at most 8 x 2 that keep no hot-start state (bench.py's headline workload); by default
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
    """Default: batches of at most 16 384 members, batches that keep their state and hot starts stay on the 8-lane kernel -- and a hot
    start that follows a cold start of the lane kernel (which kept nothing) runs cold, as the handle promises."""
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
    b.set_keep_state(True)
    b.solve(capi.MODE_COLD, 1000)                     # keeps its state: the 8-lane kernel
    assert b.last_kernel() == 1
    p3 = [problems.perturb(rng, q, 0.05) for q in p2]
    orcs = [oracle_cold(oracle, q)[0] for q in p2]
    b.set_vectors_from(p3)
    b.solve(capi.MODE_HOT_VECTORS, 1000)
    assert b.last_kernel() == 1
    for q, qp, r in zip(p3, orcs, b.results()):
        rc, n = qp.hotstart(q.g, q.lb, q.ub, q.lbA, q.ubA, 1000)
        assert_same_solution(qp, r, n)
    b.close()
    monkeypatch.delenv("RSQP_LANE")
    b = capi.Batch(probs)
    b.set_keep_state(False)
    b.solve(capi.MODE_COLD, 1000)
    assert b.last_kernel() == 1                       # 70 members: below the default threshold
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
