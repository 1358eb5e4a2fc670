"""Robustness beyond the test suite: the lane-per-problem kernel against the 8-lanes-per-problem kernel on DEGENERATE one-pattern batches
(problems.degenerate_qp: duplicate constraint, zero row, constraint parallel to a bound, integer data, singular H; shapes <= 8 x 2; 64
members each, gradients perturbed): members that end in another working set are counted, and whether x / the objective differ.
    python tools/lane_degenerate_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from restartsqp_amd import capi, problems
rng = np.random.default_rng(31)
tot = diff = objdiff = 0
stat = {}
for trial in range(4000):
    kind = int(rng.integers(0, 5))
    q = problems.degenerate_qp(rng, kind)
    if q.nV > 8 or q.nC > 2:
        continue
    probs = []
    for _ in range(64):
        p = problems.perturb(rng, q, 0.0)
        p.g = q.g + (0.0 if kind == 3 else 0.01) * rng.normal(size=q.g.shape)
        probs.append(p)
    out = []
    for lane in ("1", "0"):
        os.environ["RSQP_LANE"] = lane
        b = capi.Batch(probs); b.set_keep_state(False)
        b.solve(capi.MODE_COLD, 1000)
        out.append((b.last_kernel(), b.results())); b.close()
    assert out[0][0] == 2 and out[1][0] == 1
    for r, t in zip(out[0][1], out[1][1]):
        tot += 1
        stat[r["status"]] = stat.get(r["status"], 0) + 1
        same = r["status"] == t["status"] and r["nWSR"] == t["nWSR"] and np.array_equal(r["ws_b"], t["ws_b"]) and np.array_equal(r["ws_c"], t["ws_c"])
        if not same:
            diff += 1
            if abs(r['obj'] - t['obj']) > 1e-9 * max(1.0, abs(t['obj'])) or np.abs(r['x'] - t['x']).max() > 1e-7: objdiff += 1
            if diff <= 5: print("differ: kind", kind, q.nV, q.nC, "lane", r["status"], r["nWSR"], "8-lane", t["status"], t["nWSR"])
print("degenerate members", tot, "lane != 8-lane:", diff, "of them with another objective or x:", objdiff, "statuses", stat)
