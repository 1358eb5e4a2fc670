"""Randomized check of the batched LDS-vector SpMV kernels (entry-parallel csx_ldsvec_segscan where the plan takes it)
against scipy on random patterns: random shapes, column-length distributions with empty / one-entry / chunk-sized
majors. Not part of the test suite. Usage: python tools/spmv_random_sweep.py [patterns] [seed]"""
import os, sys
import numpy as np
import scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from restartsqp_amd import capi
npat = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
bad = 0
for t in range(npat):
    nrow = int(rng.integers(200, 20000)); ncol = int(rng.integers(50, 12000)); nb = 64
    kind = t % 5
    if kind == 0: lens = rng.integers(0, 40, size=ncol)
    elif kind == 1: lens = rng.geometric(0.08, size=ncol)
    elif kind == 2: lens = np.where(rng.random(ncol) < 0.02, rng.integers(400, 513, size=ncol), rng.integers(0, 6, size=ncol))
    elif kind == 3: lens = np.full(ncol, int(rng.integers(1, 30)))
    else: lens = rng.integers(0, 3, size=ncol)
    lens = np.minimum(lens, min(nrow, 512)).astype(np.int64)
    jc = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ir = np.concatenate([np.sort(rng.choice(nrow, size=int(l), replace=False)) for l in lens] + [np.zeros(0, np.int64)]).astype(np.int32)
    nnz = int(jc[-1])
    if nnz == 0: continue
    vals = rng.normal(size=(nb, nnz)); x = rng.normal(size=(nb, ncol)); y = rng.normal(size=(nb, nrow))
    p = capi.SpmvPlan(nrow, ncol, jc, ir, nb)
    vt, vn = p.variant(True)[0], p.variant(False)[0]
    p.upload(vals, x, transposed=False); p.upload(None, y, transposed=True)
    p.run(False); p.run(True)
    Ax, ATy = p.download(False), p.download(True)
    err = 0.0
    for k in (0, 17, 63):
        A = sp.csc_matrix((vals[k], ir, jc), shape=(nrow, ncol))
        a, b = A @ x[k], A.T @ y[k]
        err = max(err, np.abs(Ax[k] - a).max() / max(1.0, np.abs(a).max()), np.abs(ATy[k] - b).max() / max(1.0, np.abs(b).max()))
    ok = err < 1e-12
    bad += not ok
    print("pattern %2d kind %d: %5d x %5d nnz %7d variants (A'y %d, Ax %d) relerr %.1e %s" % (t, kind, nrow, ncol, nnz, vt, vn, err, "ok" if ok else "BAD"), flush=True)
    p.close()
print("DONE: %d bad" % bad)
