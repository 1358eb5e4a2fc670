"""Which members of the 512-QP hs0xx batch the tableau kernel (qp_small_g.h) hands to the null-space kernel, and why: with
RSQP_SMALL_KKT_ONLY=1 the second pass is not launched, a bailed member keeps status 25 and nWSR = 1000 + bail reason
(qp_small_g.h: 1 / 2 pivot inside the rounding band, 3 independence undecidable, 5 singular block pivot, 8 exchange without partner, 10 not eligible, 11 free variable in the cold working set, ...).   (GPU box)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RSQP_SMALL_KKT_ONLY"] = "1"
from restartsqp_amd import capi, problems, parallel
allp = problems.hs_batch(512)
b = capi.Batch(allp); b.set_keep_state(False); b.solve(capi.MODE_COLD, 1000)
res = b.results()
h = collections.Counter()
for q, r in zip(allp, res):
    if r.get("ret", None) == 9 or r["status"] not in (20,) :
        h[(q.nV, q.nC, r.get("nflips"), r["status"], r["nWSR"])] += 1
for k, v in sorted(h.items(), key=lambda x: -x[1])[:20]: print(k, v)
print(list(res[0].keys()))
