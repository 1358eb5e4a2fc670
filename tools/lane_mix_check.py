"""Tuning / robustness: the two hs071-scale batch kernels on a HETEROGENEOUS one-shape batch -- the QPs of the whole hs071 SQP
trajectory (tests/golden/sqp_traces.json) with seeded perturbations, cold start: members differ in path and length (the lane-per-
problem kernel executes the union of the paths of the 64 members of a wave).     python tools/lane_mix_check.py [nq]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from restartsqp_amd import capi, problems
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
tr = json.load(open(os.path.join(ROOT, "tests", "golden", "sqp_traces.json")))["hs071"]["qps"]
base = [problems.handler_qp(problems.hs071_nlp(np.array(g["x"]), np.array(g["lam"])), delta=g["delta"], rho=g["rho"]) for g in tr]
print("distinct QPs", len(base), "entry counts (A, H):", sorted(set((len(q.A_val), len(q.H_val)) for q in base)))
# (two sparsity patterns among the six: H has 11 or 16 entries -- the lane-per-problem kernel then lets every lane walk the CSC arrays
#  of its own problem; ONE_PATTERN=1 keeps the QPs of the most common pattern only: everything through the wave's block)
if os.environ.get("ONE_PATTERN") == "1":
    key = lambda q: (tuple(q.A_jc), tuple(q.A_ir), tuple(q.H_jc), tuple(q.H_ir))
    from collections import Counter
    best = Counter(key(q) for q in base).most_common(1)[0][0]
    base = [q for q in base if key(q) == best]
    print("QPs of the most common pattern:", len(base), "H entries", len(base[0].H_val))
rng = np.random.default_rng(20260104)
for label, pick in (("trajectory mix, member k = QP k mod %d" % len(base), lambda k: base[k % len(base)]),
                    ("trajectory mix, blocks of 64 equal QPs", lambda k: base[(k // 64) % len(base)])):
    probs = [problems.perturb(rng, pick(k)) for k in range(nq)]
    ref = None
    for lane in ("0", "1"):
        os.environ["RSQP_LANE"] = lane
        b = capi.Batch(probs)
        b.set_keep_state(False)
        for _ in range(3):
            b.solve(capi.MODE_COLD, 1000, sync=False)
        capi.check(capi.lib().rsqp_batch_sync(b._h))
        b.timer_start()
        for _ in range(20):
            b.solve(capi.MODE_COLD, 1000, sync=False)
        ms = b.timer_stop_ms() / 20
        res = b.results()
        nw = np.array([r["nWSR"] for r in res]); st = np.array([r["status"] for r in res])
        x = np.concatenate([r["x"] for r in res])
        if ref is None:
            ref = (nw, st, x)
        same = bool((nw == ref[0]).all() and (st == ref[1]).all() and np.abs(x - ref[2]).max() < 1e-9 * max(1.0, np.abs(ref[2]).max()))
        print("%s | %s: kernel %d, %.4f ms = %.0f M solves/s, mean nWSR %.2f max %d, solved %d / %d, same answers as the other kernel: %s"
              % (label, "one lane per QP" if lane == "1" else "8 lanes per QP", b.last_kernel(), ms, nq / ms / 1e3, nw.mean(), nw.max(),
                 int((st == 20).sum()), nq, same))
        b.close()
