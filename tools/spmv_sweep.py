"""Sweep the SpMV kernel variants on the BASELINE sparse shape (development aid)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from restartsqp_amd import capi, problems
n, m, nnz, nb = 10000, 20000, 200000, 256
jc, ir, rng = problems.sparse_pattern(n, m, nnz)
vals = rng.normal(size=(nb, nnz)); x = rng.normal(size=(nb, n)); y = rng.normal(size=(nb, m))
bt = 12 * nnz + 4 * (n + 1) + 8 * n + 8 * m
bn = 12 * nnz + 4 * (m + 1) + 8 * m + 8 * n
ref = {}
for var, nsl, i16 in [(0, 1, 1), (12, 1, 1), (14, 1, 1), (31, 1, 0), (32, 1, 0), (33, 1, 0), (34, 1, 0), (35, 1, 0), (36, 1, 0), (37, 1, 0), (38, 1, 0), (31, 1, 1), (32, 1, 1), (33, 1, 1), (34, 1, 1), (35, 1, 1), (36, 1, 1), (37, 1, 1), (38, 1, 1)]:
    os.environ["RSQP_SPMV_IDX16"] = str(i16)
    os.environ["RSQP_SPMV_SLICES"] = str(nsl)
    os.environ["RSQP_SPMV_VARIANT"] = str(var)
    p = capi.SpmvPlan(m, n, jc, ir, nb)
    p.upload(vals, x, False); p.upload(None, y, True)
    out = []
    for tr, b in ((True, bt), (False, bn)):
        p.run(tr, 3)
        ms = min(p.run(tr, 10) for _ in range(3))
        res = p.download(tr)
        if var == 0:
            ref[tr] = res
        err = np.abs(res - ref[tr]).max() / np.abs(ref[tr]).max()
        out.append("%s %.3f ms %.0f GB/s relerr %.1e" % ("A'y" if tr else "Ax ", ms, b * nb / ms / 1e6, err))
    print("variant", var, "slices", nsl, "idx16", i16, " | ".join(out), flush=True)
    p.close()
