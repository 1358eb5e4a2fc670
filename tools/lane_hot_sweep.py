"""Tuning: hs071-scale batches that KEEP their state -- cold start (state written back: lane-per-problem kernel above 16 384 members,
RSQP_LANE=0: the 8-lane kernel), hot start on new vectors, hot start with new matrices (always the 8-lane kernel, from the state
either kernel wrote): median launch time of 10 solves each (HIP events), alternating between two sets of vectors.
    python tools/lane_hot_sweep.py [nq]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from restartsqp_amd import capi, problems
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(5)
probs = problems.hs071_scale_batch(nq)
alt = [[problems.perturb(rng, q, 0.05) for q in probs] for _ in range(2)]
for lane in ("0", "1"):
    os.environ["RSQP_LANE"] = lane
    b = capi.Batch(probs)
    b.set_keep_state(True)
    out = {}
    for name, mode in (("cold start, state kept", capi.MODE_COLD), ("hot start, new vectors", capi.MODE_HOT_VECTORS), ("hot start, new matrices", capi.MODE_HOT_MATRICES)):
        ms, nw = [], []
        for k in range(10):
            b.set_vectors_from(alt[k % 2] if mode != capi.MODE_COLD else probs)
            b.solve(mode, 1000, sync=True)
            ms.append(b.last_solve_ms())
        res = b.results()
        out[name] = (float(np.median(ms)), float(np.mean([r["nWSR"] for r in res])), int(sum(r["status"] == 20 for r in res)))
    print("%s:" % ("cold starts on the lane-per-problem kernel" if lane == "1" else "everything on the 8-lane kernel"),
          "; ".join("%s %.4f ms (mean nWSR %.2f, %d solved)" % ((k,) + v) for k, v in out.items()))
    b.close()
