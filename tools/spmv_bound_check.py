"""The LDS-vector SpMV kernels on the BASELINE sparse shape (n = 10k, m = 20k, 200k entries, 256 members):
median launch time per variant and the largest deviation from a float64 numpy product on 4 members.
Variants 50 / 51 (streaming upper bound, wrong results by construction) exist only in the tuning build
-DRSQP_SPMV_EXPERIMENT selected with RSQP_LIB."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from restartsqp_amd import capi, problems
import scipy.sparse as sp
n, m, nnz, nb = 10000, 20000, 200000, 256
jc, ir, rng = problems.sparse_pattern(n, m, nnz)
vals = rng.normal(size=(nb, nnz)); x = rng.normal(size=(nb, n)); y = rng.normal(size=(nb, m))
bt = 12 * nnz + 4 * (n + 1) + 8 * n + 8 * m
bn = 12 * nnz + 4 * (m + 1) + 8 * m + 8 * n
variants = [int(a) for a in sys.argv[1:]] or [35, 38, 40]
for var in variants:
    os.environ["RSQP_SPMV_VARIANT"] = str(var)
    p = capi.SpmvPlan(m, n, jc, ir, nb)
    p.upload(vals, x, False); p.upload(None, y, True)
    out = []
    for tr, b in ((True, bt), (False, bn)):
        p.run(tr, 3)
        ms = sorted(p.run(tr, 1) for _ in range(30))[15]
        res = p.download(tr)
        err = 0.0
        for k in (0, 1, 100, 255):
            A = sp.csc_matrix((vals[k], ir, jc), shape=(m, n))
            ref = A.T @ y[k] if tr else A @ x[k]
            err = max(err, np.abs(res[k] - ref).max() / np.abs(ref).max())
        out.append("%s %.4f ms %.0f GB/s algorithmic (%.3f of 8 TB/s) relerr %.1e variant %s" % ("A'y" if tr else "Ax ", ms, b * nb / ms / 1e6, b * nb / ms / 1e6 / 8000, err, p.variant(tr)))
    print("requested", var, " | ".join(out), flush=True)
    p.close()
