"""Time per shape class of the 512-QP hs0xx batch (BASELINE configs[4]): each class solved alone, largest first.
Shows which members bound the mixed batch once the largest class gets faster."""
import os, sys
from collections import defaultdict
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from restartsqp_amd import capi, problems
allp = problems.hs_batch(512)
cl = defaultdict(list)
for p in allp:
    cl[(p.nV, p.nC)].append(p)
rows = []
for (nV, nC), probs in cl.items():
    b = capi.Batch(probs)
    b.solve(capi.MODE_COLD, 1000)
    ms = sorted(b.solve(capi.MODE_COLD, 1000) or b.last_solve_ms() for _ in range(5))[2]
    res = b.results()
    rows.append((ms, nV, nC, len(probs), max(r["nWSR"] for r in res)))
    b.close() if hasattr(b, "close") else None
for ms, nV, nC, n, mw in sorted(rows, reverse=True)[:14]:
    print("%3d x %3d  members %3d  max nWSR %4d  %.3f ms  (%.1f us per change of the longest)" % (nV, nC, n, mw, ms, 1e3 * ms / max(mw, 1)), flush=True)
