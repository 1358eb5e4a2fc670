"""Synthetic inputs for the configurations named in BASELINE.json (generators only; no solver).

* ``hs071_first_qp``  -- first SQP subproblem of hs071, derived analytically from the model in
  reference ``test/CUTE_examples/hs071.nl`` (x0 = (1,5,5,1), 1 <= x <= 5, c1 = x1x2x3x4 >= 25,
  c2 = sum x^2 = 40) with the reference defaults delta = 1, rho = 1 (``src/Options.cpp:33,44``)
  and zero initial multipliers. "Derived", not captured from a reference run.
* ``dense_qp``        -- SURVEY.md 8(d): n = 2048, m = 4096, seed 20260101.
* ``sparse_qp`` / ``sparse_sequence`` -- n = 10 000, m = 20 000, 200 000 Jacobian non-zeros,
  seed 20260102; H = diag(1 + |N(0,1)|).
* ``hs_batch``        -- hs0xx-scale batch: hs071 fixture + definite random QPs of the shapes
  of the 18 reference dumps, with seeded 1 % perturbations.
"""
import numpy as np

from .qpdump import QPData, dense_to_csc
from .sqptypes import INF, IdentityInfo, NLPInfo, SpTripletMat


# ------------------------------------------------------------------------------------
# hs071
# ------------------------------------------------------------------------------------
def hs071_nlp(x=None, lam=None):
    """Closed-form hs071 at x (default: the .nl starting point): f, grad f, c, Jacobian (1-based
    COO, 8 entries), Hessian of the Lagrangian f - lam'c (lower triangle, 10 entries)."""
    x = np.array([1.0, 5.0, 5.0, 1.0]) if x is None else np.asarray(x, float)
    lam = np.zeros(2) if lam is None else np.asarray(lam, float)
    x1, x2, x3, x4 = x
    f = x1 * x4 * (x1 + x2 + x3) + x3
    grad = np.array([x4 * (2 * x1 + x2 + x3), x1 * x4, x1 * x4 + 1.0, x1 * (x1 + x2 + x3)])
    c = np.array([x1 * x2 * x3 * x4, x1 * x1 + x2 * x2 + x3 * x3 + x4 * x4])
    J = SpTripletMat(2, 4, [1, 1, 1, 1, 2, 2, 2, 2], [1, 2, 3, 4, 1, 2, 3, 4],
                     [x2 * x3 * x4, x1 * x3 * x4, x1 * x2 * x4, x1 * x2 * x3, 2 * x1, 2 * x2, 2 * x3, 2 * x4], False)
    # Hessian of f minus sum lam_i Hessian of c_i (SQPTNLP::Eval_Hessian negates lambda, src/SQPTNLP.cpp:124-126)
    l1, l2 = lam
    Hf = {(1, 1): 2 * x4, (2, 1): x4, (3, 1): x4, (4, 1): 2 * x1 + x2 + x3, (4, 2): x1, (4, 3): x1}
    Hc1 = {(2, 1): x3 * x4, (3, 1): x2 * x4, (3, 2): x1 * x4, (4, 1): x2 * x3, (4, 2): x1 * x3, (4, 3): x1 * x2}
    rows, cols, vals = [], [], []
    for r in range(1, 5):
        for cidx in range(1, r + 1):
            v = Hf.get((r, cidx), 0.0) - l1 * Hc1.get((r, cidx), 0.0) - (l2 * 2.0 if r == cidx else 0.0)
            rows.append(r); cols.append(cidx); vals.append(v)
    H = SpTripletMat(4, 4, rows, cols, vals, True)
    return dict(x=x, f=f, grad=grad, c=c, J=J, H=H,
                x_l=np.ones(4), x_u=5.0 * np.ones(4), c_l=np.array([25.0, 40.0]), c_u=np.array([np.inf, 40.0]),
                info=NLPInfo(nCon=2, nVar=4, nnz_jac_g=8, nnz_h_lag=10))


def hs035_nlp(x=None, lam=None):
    """Closed-form hs035 as the reference's AMPL file states it (``test/CUTE_examples/hs035.nl``): 3 variables with lower
    bounds 0, x0 = (0.5, 0.5, 0.5); objective 9 - 8x1 - 6x2 - 4x3 + 2x1^2 + 2x2^2 + x3^2 + 2x1x2 + 2x1x3; one linear
    constraint x1 + x2 + 2x3 <= 3. A convex QP itself: the SQP run is the trust region growing until the QP step fits."""
    x = np.array([0.5, 0.5, 0.5]) if x is None else np.asarray(x, float)
    x1, x2, x3 = x
    f = 9.0 - 8 * x1 - 6 * x2 - 4 * x3 + 2 * x1 * x1 + 2 * x2 * x2 + x3 * x3 + 2 * x1 * x2 + 2 * x1 * x3
    grad = np.array([-8 + 4 * x1 + 2 * x2 + 2 * x3, -6 + 4 * x2 + 2 * x1, -4 + 2 * x3 + 2 * x1])
    c = np.array([x1 + x2 + 2 * x3])
    J = SpTripletMat(1, 3, [1, 1, 1], [1, 2, 3], [1.0, 1.0, 2.0], False)
    H = SpTripletMat(3, 3, [1, 2, 2, 3, 3], [1, 1, 2, 1, 3], [4.0, 2.0, 4.0, 2.0, 2.0], True)
    return dict(x=x, f=f, grad=grad, c=c, J=J, H=H, x_l=np.zeros(3), x_u=np.full(3, np.inf),
                c_l=np.array([-np.inf]), c_u=np.array([3.0]), info=NLPInfo(nCon=1, nVar=3, nnz_jac_g=3, nnz_h_lag=5))


def hs065_nlp(x=None, lam=None):
    """Closed-form hs065 as the reference's AMPL file states it (``test/CUTE_examples/hs065.nl``): 3 free variables,
    x0 = (-5, 5, 0); objective (x1-x2)^2 + (x1+x2-10)^2/9 + (x3-5)^2; FOUR constraints -- c0 = x1^2+x2^2+x3^2 <= 48 and
    the three boxes -4.5 <= x1, x2 <= 4.5, -5 <= x3 <= 5 modelled as range constraints c1..c3 = x1..x3 (the `r` block),
    6 Jacobian entries, Hessian of the Lagrangian f - lam'c (lower triangle, 4 entries: the 3 diagonal ones + (2,1))."""
    x = np.array([-5.0, 5.0, 0.0]) if x is None else np.asarray(x, float)
    lam = np.zeros(4) if lam is None else np.asarray(lam, float)
    x1, x2, x3 = x
    f = (x1 - x2) ** 2 + (x1 + x2 - 10.0) ** 2 / 9.0 + (x3 - 5.0) ** 2
    grad = np.array([2 * (x1 - x2) + 2 * (x1 + x2 - 10.0) / 9.0, -2 * (x1 - x2) + 2 * (x1 + x2 - 10.0) / 9.0, 2 * (x3 - 5.0)])
    c = np.array([x1 * x1 + x2 * x2 + x3 * x3, x1, x2, x3])
    J = SpTripletMat(4, 3, [1, 1, 1, 2, 3, 4], [1, 2, 3, 1, 2, 3], [2 * x1, 2 * x2, 2 * x3, 1.0, 1.0, 1.0], False)
    l0 = lam[0]
    H = SpTripletMat(3, 3, [1, 2, 2, 3], [1, 1, 2, 3],
                     [2.0 + 2.0 / 9.0 - 2.0 * l0, -2.0 + 2.0 / 9.0, 2.0 + 2.0 / 9.0 - 2.0 * l0, 2.0 - 2.0 * l0], True)
    return dict(x=x, f=f, grad=grad, c=c, J=J, H=H, x_l=np.full(3, -np.inf), x_u=np.full(3, np.inf),
                c_l=np.array([-np.inf, -4.5, -4.5, -5.0]), c_u=np.array([48.0, 4.5, 4.5, 5.0]),
                info=NLPInfo(nCon=4, nVar=3, nnz_jac_g=6, nnz_h_lag=4))


def handler_qp(nlp, delta=1.0, rho=1.0, name="hs071_first_qp"):
    """The QP that QPhandler builds from an NLP iterate (src/QPhandler.cpp:39-51,185-201,272-297):
    variables (p, u, v), A = [J I -I], H = blkdiag(H_k, 0), g = (grad f, rho e)."""
    n, m = nlp["info"].nVar, nlp["info"].nCon
    nV = n + 2 * m
    J, Ht = nlp["J"], nlp["H"]
    A = np.zeros((m, nV))
    for r, c, v in zip(J.RowIndex, J.ColIndex, J.MatVal):
        A[r - 1, c - 1] = v
    A[:, n:n + m] = np.eye(m)
    A[:, n + m:] = -np.eye(m)
    H = np.zeros((nV, nV))
    for r, c, v in zip(Ht.RowIndex, Ht.ColIndex, Ht.MatVal):
        H[r - 1, c - 1] = v
        H[c - 1, r - 1] = v
    lb = np.zeros(nV); ub = np.full(nV, INF)
    lb[:n] = np.maximum(nlp["x_l"] - nlp["x"], -delta)
    ub[:n] = np.minimum(nlp["x_u"] - nlp["x"], delta)
    g = np.concatenate([nlp["grad"], rho * np.ones(2 * m)])
    lbA = nlp["c_l"] - nlp["c"]; ubA = nlp["c_u"] - nlp["c"]
    return QPData(nV, m, *dense_to_csc(H), *dense_to_csc(A), g, lb, ub, lbA, ubA, name=name)


def hs071_first_qp():
    return handler_qp(hs071_nlp())


# ------------------------------------------------------------------------------------
# random convex QPs
# ------------------------------------------------------------------------------------
def random_qp(rng, nV, nC, density=0.5, name=""):
    M = rng.normal(size=(nV, nV))
    H = M @ M.T / nV + np.eye(nV)
    A = rng.normal(size=(nC, nV)) * (rng.random((nC, nV)) < density)
    g = 3.0 * rng.normal(size=nV)
    xh = rng.normal(size=nV)
    lb = xh - np.abs(rng.normal(size=nV)); ub = xh + np.abs(rng.normal(size=nV))
    lbA = A @ xh - np.abs(rng.normal(size=nC)); ubA = A @ xh + np.abs(rng.normal(size=nC))
    return QPData(nV, nC, *dense_to_csc(H), *dense_to_csc(A), g, lb, ub, lbA, ubA, name=name)


def banded_qp(rng, nV, nC, density=0.3, hb=2, free=False, name="banded"):
    """random convex QP with a (2 hb + 1)-band strictly diagonally dominant Hessian (the general range-space path's banded operator)"""
    H = np.diag(1.0 + np.abs(rng.normal(size=nV)))
    for off in range(1, hb + 1):
        o = 0.3 * rng.normal(size=nV - off)
        H += np.diag(o, off) + np.diag(o, -off)
    H += np.diag(np.abs(H - np.diag(np.diag(H))).sum(axis=1))
    A = rng.normal(size=(nC, nV)) * (rng.random((nC, nV)) < density)
    g = 3.0 * rng.normal(size=nV)
    xh = rng.normal(size=nV)
    lb = xh - np.abs(rng.normal(size=nV)); ub = xh + np.abs(rng.normal(size=nV))
    if free:
        lb[::3] = -np.inf; ub[1::3] = np.inf
    lbA = A @ xh - np.abs(rng.normal(size=nC)); ubA = A @ xh + np.abs(rng.normal(size=nC))
    return QPData(nV, nC, *dense_to_csc(H), *dense_to_csc(A), g, lb, ub, lbA, ubA, name=name)


def perturb(rng, q, rel=0.01):
    """Seeded perturbation of g and of the bounds (keeps lb <= ub)."""
    def pb(lo, hi):
        fin_lo, fin_hi = np.abs(lo) < 1e17, np.abs(hi) < 1e17
        lo2 = np.where(fin_lo, lo + rel * rng.normal(size=lo.shape), lo)
        hi2 = np.where(fin_hi, hi + rel * rng.normal(size=hi.shape), hi)
        return lo2, np.maximum(hi2, lo2)
    lb, ub = pb(q.lb, q.ub)
    lbA, ubA = pb(q.lbA, q.ubA)
    g = q.g * (1.0 + rel * rng.normal(size=q.g.shape))
    return QPData(q.nV, q.nC, q.H_jc, q.H_ir, q.H_val, q.A_jc, q.A_ir, q.A_val, g, lb, ub, lbA, ubA, name=q.name)


def degenerate_qp(rng, kind):
    """Small QPs with the degeneracies an active-set method has to survive: 0 duplicate
    constraint, 1 zero row, 2 constraint parallel to a bound, 3 integer data (ties), 4 singular H."""
    nV, nC = int(rng.integers(2, 10)), int(rng.integers(2, 12))
    q = random_qp(rng, nV, nC, density=0.7)
    A, H = q.dense_A(), q.dense_H()
    if kind == 0:
        A[1] = A[0]; q.lbA[1] = q.lbA[0]; q.ubA[1] = q.ubA[0]
    elif kind == 1:
        A[0] = 0.0; q.lbA[0] = -1.0; q.ubA[0] = 1.0
    elif kind == 2:
        A[0] = 0.0; A[0, 0] = 1.0; q.lbA[0] = q.lb[0]; q.ubA[0] = q.ub[0]
    elif kind == 3:
        A = np.round(2.0 * A); q.g = np.round(q.g); q.lb = np.floor(q.lb); q.ub = np.ceil(q.ub) + 1.0
        q.lbA = -2.0 * np.ones(nC); q.ubA = 2.0 * np.ones(nC)
    elif kind == 4:
        H[:, -1] = 0.0; H[-1, :] = 0.0
    return QPData(nV, nC, *dense_to_csc(H), *dense_to_csc(A), q.g, q.lb, q.ub, q.lbA, q.ubA, name="degenerate-%d" % kind)


# shapes (nV, nC) of the 18 dumps under reference test/unsolved_QP_data/
HS_DUMP_SHAPES = [(8, 3), (10, 4), (8, 3), (5, 1), (13, 5), (7, 2), (4, 0), (15, 4), (5, 1), (16, 6), (12, 4),
                  (12, 4), (23, 6), (5, 1), (7, 1), (20, 6), (37, 14), (69, 28)]


def hs_batch(nq, seed=20260103, shapes=None, max_nV=None):
    """hs0xx-scale batch: problem 0 is the hs071 first QP, the others are its seeded 1 %
    perturbations interleaved with definite random QPs of the reference dumps' shapes."""
    rng = np.random.default_rng(seed)
    shapes = shapes or HS_DUMP_SHAPES
    if max_nV:
        shapes = [s for s in shapes if s[0] <= max_nV]
    base = hs071_first_qp()
    out = [base]
    k = 0
    while len(out) < nq:
        if len(out) % 2 == 1:
            out.append(perturb(rng, base))
        else:
            nV, nC = shapes[k % len(shapes)]
            k += 1
            out.append(random_qp(rng, nV, nC, density=0.4, name="hs-shape-%dx%d" % (nV, nC)))
    return out[:nq]


def hs071_scale_batch(nq, seed=20260103):
    """nq QPs of exactly the hs071 shape (8 x 2): the fixture and seeded perturbations of it."""
    rng = np.random.default_rng(seed)
    base = hs071_first_qp()
    return [base] + [perturb(rng, base) for _ in range(nq - 1)]


# ------------------------------------------------------------------------------------
# SURVEY.md 8(d) synthetic configurations
# ------------------------------------------------------------------------------------
def dense_qp(n=2048, m=4096, seed=20260101):
    rng = np.random.default_rng(seed)
    M = rng.normal(size=(n, n))
    H = M @ M.T / n + np.eye(n)
    A = rng.normal(size=(m, n))
    g = rng.normal(size=n)
    xh = rng.normal(size=n)
    lbA = A @ xh - np.abs(rng.normal(size=m)); ubA = A @ xh + np.abs(rng.normal(size=m))
    return QPData(n, m, *dense_to_csc(H), *dense_to_csc(A), g, -10.0 * np.ones(n), 10.0 * np.ones(n), lbA, ubA,
                  name="dense_%dx%d" % (n, m))


def sparse_pattern(n=10000, m=20000, nnz=200000, seed=20260102):
    """Exactly nnz distinct positions of an m x n matrix, sampled without replacement; CSC."""
    rng = np.random.default_rng(seed)
    lin = rng.choice(m * n, size=nnz, replace=False)
    rows, cols = (lin % m).astype(np.int64), (lin // m).astype(np.int64)
    order = np.lexsort((rows, cols))
    rows, cols = rows[order], cols[order]
    jc = np.zeros(n + 1, np.int64)
    np.add.at(jc, cols + 1, 1)
    return np.cumsum(jc).astype(np.int32), rows.astype(np.int32), rng


def sparse_qp(n=10000, m=20000, nnz=200000, seed=20260102, box=1.0, band=0):
    """BASELINE configs[3]: H = diag(1 + |N(0,1)|); band=5: "+ optional 5-band SPD" -- two off-diagonals on
    each side, 0.25 N(0,1), the diagonal raised by the absolute row sums so that H stays strictly diagonally dominant (SPD).
    Jacobian, gradient and limits are those of the diagonal configuration (the band's numbers are drawn after everything else)."""
    jc, ir, rng = sparse_pattern(n, m, nnz, seed)
    val = rng.normal(size=nnz)
    h = 1.0 + np.abs(rng.normal(size=n))
    H_jc = np.arange(n + 1, dtype=np.int32); H_ir = np.arange(n, dtype=np.int32)
    g = rng.normal(size=n)
    xh = rng.normal(size=n) * 0.3
    Ax = np.zeros(m)
    cols = np.repeat(np.arange(n), np.diff(jc))
    np.add.at(Ax, ir, val * xh[cols])
    lbA = Ax - np.abs(rng.normal(size=m)); ubA = Ax + np.abs(rng.normal(size=m))
    name = "sparse_%dx%d" % (n, m)
    if band:
        assert band == 5, "band: 0 (diagonal) or 5"
        o1 = 0.25 * rng.normal(size=n - 1); o2 = 0.25 * rng.normal(size=n - 2)
        rs = np.zeros(n)
        rs[:-1] += np.abs(o1); rs[1:] += np.abs(o1); rs[:-2] += np.abs(o2); rs[2:] += np.abs(o2)
        d = h + rs
        rows, colsH, vals = [], [], []
        for off, o in ((0, d), (1, o1), (2, o2)):
            i = np.arange(n - off)
            rows.append(i + off); colsH.append(i); vals.append(o)
            if off:
                rows.append(i); colsH.append(i + off); vals.append(o)
        rows, colsH, vals = np.concatenate(rows), np.concatenate(colsH), np.concatenate(vals)
        order = np.lexsort((rows, colsH))
        H_ir = rows[order].astype(np.int32); h = vals[order]
        H_jc = np.concatenate(([0], np.cumsum(np.bincount(colsH, minlength=n)))).astype(np.int32)
        name += "_band5"
    return QPData(n, m, H_jc, H_ir, h, jc, ir, val, g, -box * np.ones(n), box * np.ones(n), lbA, ubA, name=name)


def sparse_sequence(q, nsteps=50, seed=20260102):
    """Warm-started sequence of SURVEY.md 8(d): QP_k = the base QP with g and the bounds perturbed by 1 %
    (odd k: FIXED matrices -> hot start on vectors); even k additionally rescale the base Jacobian values by
    (1 + 0.01 N(0,1)) (VARIED). Every member is a perturbation of the BASE problem: accumulating the
    perturbations is a random walk of 20 000 constraint intervals, which makes the QP infeasible after ~35
    steps (observed: status 22, confirmed by a cold start). Yields (QPData, matrices_changed)."""
    rng = np.random.default_rng(seed + 1)
    for k in range(1, nsteps + 1):
        nxt = perturb(rng, q)
        changed = k % 2 == 0
        if changed:
            nxt.A_val = q.A_val * (1.0 + 0.01 * rng.normal(size=q.A_val.shape))
        yield nxt, changed


# ------------------------------------------------------------------------------------
# the reference's second set of real QP inputs (test/unsolved_QPs/*.hpp)
# ------------------------------------------------------------------------------------
def unsolved_qps(path=None):
    """The nine QP data headers of reference test/unsolved_QPs (qpOASES CSC layout) from the JSON fixture
    tests/golden/unsolved_qps.json (made by tests/golden/make_unsolved_qp_fixtures.py). Returns
    [(QPData, expected_status or None)]. All are non-convex and most carry a NON-SYMMETRIC H (the dumper of
    that reference revision permuted the mirrored values); the arrays are passed on exactly as recorded."""
    import json
    import os
    if path is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "unsolved_qps.json")
    f = lambda a: np.array([float(v) for v in a], dtype=np.float64)
    i = lambda a: np.array(a, dtype=np.int32)
    out = []
    for name, q in json.load(open(path))["qps"].items():
        out.append((QPData(q["nV"], q["nC"], i(q["H_jc"]), i(q["H_ir"]), f(q["H_val"]), i(q["A_jc"]), i(q["A_ir"]),
                           f(q["A_val"]), f(q["g"]), f(q["lb"]), f(q["ub"]), f(q["lbA"]), f(q["ubA"]), name=name),
                    q.get("expected_status")))
    return out
