"""Runs the KKT-tableau kernel on the 69 x 28 members of the hs0xx batch (profiling target of tools/pmc_k.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from restartsqp_amd import capi, problems
probs = [p for p in problems.hs_batch(512) if p.nV == 69]
b = capi.Batch(probs)
b.set_keep_state(False)
ms = []
for _ in range(12):
    b.solve(capi.MODE_COLD, 1000)
    ms.append(b.last_solve_ms())
res = b.results()
print("members", len(probs), "ms", float(np.median(ms)), "nWSR", [r["nWSR"] for r in res])
