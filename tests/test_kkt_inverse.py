"""The register-resident KKT-tableau formulation (restartsqp_amd/csrc/qp_small_g.h; CPU prototype tools/proto_k/proto_g.cpp).

CPU part: the prototype against the oracle -- every QP either identical (status, working sets, nWSR, x / y to 1e-9) or a
clean bail-out. GPU part: cold-start-only batches (keep_state = 0) of mid-size problems run the kernel; members it bails
on are re-solved by the null-space kernel, so EVERY member must match the oracle exactly as on the default path."""
import os
import sys

import numpy as np
import pytest

from conftest import oracle_cold
from restartsqp_amd import parallel, problems

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_prototype_matches_oracle_or_bails(oracle):
    sys.path.insert(0, os.path.join(ROOT, "tools", "proto_k"))
    import check as K
    rng = np.random.default_rng(31)
    probs = problems.hs_batch(96) + [problems.random_qp(rng, int(rng.integers(9, 60)), int(rng.integers(1, 30)),
                                                        density=float(rng.choice([0.2, 0.5, 1.0]))) for _ in range(120)]
    tally = {}
    for q in probs:
        r = K.compare(q)
        assert not r.startswith("DIFF"), (q.name, q.nV, q.nC, r)
        tally[r] = tally.get(r, 0) + 1
    assert tally.get("same", 0) >= 150, tally                 # (the tableau carries the hs071-like members too; bails: rounding-band pivots)
    # degenerate inputs (exact ties): bails are fine, mismatches must stay as rare as for the GPU engines
    rng = np.random.default_rng(32)
    res = [K.compare(problems.degenerate_qp(rng, int(rng.integers(0, 5)))) for _ in range(400)]
    assert sum(r.startswith("DIFF") for r in res) <= 6, [r for r in res if r.startswith("DIFF")]


def _same(q, r, qp, n):
    assert r["status"] == qp.exitflag() and r["nWSR"] == n, (q.name, q.nV, q.nC, r["status"], qp.exitflag(), r["nWSR"], n)
    assert np.array_equal(r["ws_b"], qp.ws_bounds) and np.array_equal(r["ws_c"], qp.ws_constraints), q.name
    assert np.abs(r["x"] - qp.x).max() <= 1e-9 * max(1.0, np.abs(qp.x).max())
    assert np.abs(r["y"] - qp.y).max() <= 1e-9 * max(1.0, np.abs(qp.y).max())


@pytest.mark.gpu
def test_cold_only_batches_match_oracle(capi, oracle):
    """BASELINE configs[4] (512 mixed hs0xx QPs, largest first) and 300 random convex QPs of 9..69 variables as cold-start-only
    batches: the KKT-tableau kernel + the null-space kernel on what it bails on; every member vs the oracle, and
    the batch bit-identical when solved again."""
    allp = problems.hs_batch(512)
    rng = np.random.default_rng(7)
    rnd = [problems.random_qp(rng, int(rng.integers(9, 70)), int(rng.integers(1, 29)), density=float(rng.choice([0.2, 0.5, 1.0])))
           for _ in range(300)]
    rnd2 = [problems.random_qp(rng, int(rng.integers(33, 64)), int(rng.integers(20, 64)), density=float(rng.choice([0.2, 0.5, 1.0])))
            for _ in range(60)]                  # the 64 x 64 instantiation of the kernel
    for probs in ([allp[k] for k in parallel.balanced_order(allp)], rnd, rnd2):
        b = capi.Batch(probs)
        b.set_keep_state(False)
        b.solve(capi.MODE_COLD, 1000)
        res = b.results()
        ok, kkt = b.test_optimality()
        for q, r, o in zip(probs, res, ok):
            qp, rc, n = oracle_cold(oracle, q)
            _same(q, r, qp, n)
            assert o == 1
        b.solve(capi.MODE_COLD, 1000)
        for r, r2 in zip(res, b.results()):
            assert np.array_equal(r["x"], r2["x"]) and np.array_equal(r["y"], r2["y"]) and r["nWSR"] == r2["nWSR"]
        b.close()


@pytest.mark.gpu
def test_bailed_members_take_the_null_space_path(capi, oracle):
    """Non-convex and degenerate members (the reference's dumps, singular Hessians, LP-like data) in a cold-start-only batch
    next to convex ones: what the KKT-tableau kernel cannot carry must come back exactly as the default path solves it."""
    from conftest import dump_paths
    from restartsqp_amd.qpdump import read_qore_dump
    rng = np.random.default_rng(11)
    probs = [read_qore_dump(p) for p in dump_paths()]
    probs += [problems.degenerate_qp(rng, k % 5) for k in range(40)]
    probs += [problems.random_qp(rng, 40, 20) for _ in range(8)]
    b0 = capi.Batch(probs)                     # default: hot-start state kept -> null-space kernels only
    b0.solve(capi.MODE_COLD, 1000)
    ref = b0.results()
    b0.close()
    b = capi.Batch(probs)
    b.set_keep_state(False)
    b.solve(capi.MODE_COLD, 1000)
    for q, r, r0 in zip(probs, b.results(), ref):
        assert r["status"] == r0["status"] and r["nWSR"] == r0["nWSR"], (q.name, r["status"], r0["status"], r["nWSR"], r0["nWSR"])
        assert np.array_equal(r["ws_b"], r0["ws_b"]) and np.array_equal(r["ws_c"], r0["ws_c"]), q.name
        if q.name.startswith("hs-shape") or "random" in q.name or q.name == "":
            continue
        if r["status"] == 20:
            assert np.abs(r["x"] - r0["x"]).max() <= 1e-9 * max(1.0, np.abs(r0["x"]).max()), q.name
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("bail_after", [None, "0", "1", "3"])
def test_hot_starts_and_hand_over(bail_after):
    """Default batches (hot-start state kept): members of 20..60 variables run the KKT-tableau kernel cold AND hot;
    non-convex members sit in the same batch with null-space states. With the test hook RSQP_K_DEBUG_BAIL=n every hot start
    of the kernel bails out before its n-th change and the null-space kernel continues from the stored state (factors rebuilt
    for the stored working set, homotopy data kept): same nWSR, working sets and point as the oracle's hot start either way
    (tests/checks/k_hot_check.py; a child process: the hook is read once per process)."""
    import subprocess
    env = dict(os.environ)
    if bail_after is not None:
        env["RSQP_K_DEBUG_BAIL"] = bail_after
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "checks", "k_hot_check.py")], capture_output=True, text=True,
                       timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0 and "K HOT OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
