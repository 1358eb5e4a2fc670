"""Oracle vs the reference's own fixtures / unit-test properties for the sparse containers
(SURVEY.md 8(c)): the CSC recorded from a run of the reference's SpHbMat, and the properties
checked by reference test/unitTest/test_SpHbMat.cpp:404-479 and test_SpTripletMat.cpp:186-326
(dense<->sparse round trip, SpMV / SpMV' against a dense double loop, COO -> HB -> COO) on
matrices <= 10x10 with integer entries 1..10 -- seeded here, exact == on doubles as there."""
import numpy as np
import pytest

from restartsqp_amd import problems
from restartsqp_amd.qpdump import csc_to_dense


def test_recorded_reference_fixture(oracle):
    # 2x3 J with the [J I -I] IdentityInfo, output of the reference's SpHbMat
    irow, jcol, val = [1, 1, 2, 2], [1, 2, 2, 3], [1.0, 2.0, 3.0, 4.0]
    ident = [(1, 4, 2, 1.0), (1, 6, 2, -1.0)]
    jc, ir, v, order = oracle.sphb_set_structure(2, 7, irow, jcol, val, ident)
    assert jc.tolist() == [0, 1, 3, 4, 5, 6, 7, 8]
    assert ir.tolist() == [0, 0, 1, 1, 0, 1, 0, 1]
    assert v.tolist() == [1, 2, 3, 4, 1, 1, -1, -1]
    assert order.tolist() == list(range(8))
    x = np.array([1.0, 1.0, 3.0, 0.0, 1.0, 0.0, 0.0])
    assert oracle.sphb_times(2, 7, jc, ir, v, x).tolist() == [3.0, 16.0]


def _rand_int_matrix(rng):
    nr, nc = int(rng.integers(1, 11)), int(rng.integers(1, 11))
    nnz = int(rng.integers(1, nr * nc + 1))
    M = np.zeros(nr * nc)
    M[rng.permutation(nr * nc)[:nnz]] = rng.integers(1, 11, size=nnz)
    return M.reshape(nr, nc)


@pytest.mark.parametrize("seed", range(25))
@pytest.mark.parametrize("compressed_row", [False, True])
def test_unit_test_properties(oracle, seed, compressed_row):
    rng = np.random.default_rng(seed)
    M = _rand_int_matrix(rng)
    nr, nc = M.shape
    ptr, idx, val = oracle.sphb_from_dense(M, compressed_row)
    # TEST_DENSE_SPARSE_MATRIX_CONVERSION
    assert np.array_equal(oracle.sphb_to_dense(nr, nc, ptr, idx, val, compressed_row), M)
    # TEST_SPARSE_MATRIX_VECTOR_MULTIPLICATION / TRANSPOSED
    p = rng.integers(1, 11, size=nc).astype(float)
    pt = rng.integers(1, 11, size=nr).astype(float)
    assert np.array_equal(oracle.sphb_times(nr, nc, ptr, idx, val, p, compressed_row), M @ p)
    assert np.array_equal(oracle.sphb_transposed_times(nr, nc, ptr, idx, val, pt, compressed_row), M.T @ pt)
    # TEST_TRIPLET_HB_MATRIX_CONVERSION: COO (shuffled) -> HB equals the dense-built HB
    r, c = np.nonzero(M)
    perm = rng.permutation(len(r))
    irow, jcol, tv = (r[perm] + 1), (c[perm] + 1), M[r[perm], c[perm]]
    ptr2, idx2, val2, order = oracle.sphb_set_structure(nr, nc, irow, jcol, tv, None, compressed_row)
    assert np.array_equal(ptr2, ptr) and np.array_equal(idx2, idx) and np.array_equal(val2, val)
    # order_ maps triplet position -> compressed position; setMatVal goes through it
    newv = rng.integers(1, 11, size=len(tv)).astype(float)
    mv = oracle.sphb_set_matval(order, newv, val2.copy(), 0)
    assert np.array_equal(mv[order], newv)
    # triplet products (test_SpTripletMat)
    assert np.array_equal(oracle.triplet_times(nr, nc, irow, jcol, tv, p), M @ p)
    assert np.array_equal(oracle.triplet_times(nr, nc, irow, jcol, tv, pt, transposed=True), M.T @ pt)


@pytest.mark.parametrize("seed", range(10))
def test_symmetric_structure(oracle, seed):
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(1, 9))
    L = np.tril(rng.integers(0, 6, size=(n, n))).astype(float)
    r, c = np.nonzero(L)
    if len(r) == 0:
        L[0, 0] = 1.0
        r, c = np.nonzero(L)
    S = L + np.tril(L, -1).T
    jc, ir, val, order = oracle.sphb_set_structure_sym(n, n, r + 1, c + 1, L[r, c], True)
    assert np.array_equal(csc_to_dense(n, n, jc, ir, val), S)
    # setMatVal(rhs): two writes per off-diagonal (SpHbMat.cpp:383-393)
    newv = rng.integers(1, 9, size=len(r)).astype(float)
    L2 = np.zeros_like(L); L2[r, c] = newv
    mv = oracle.sphb_set_matval_sym(r + 1, c + 1, True, order, newv, val.copy())
    assert np.array_equal(csc_to_dense(n, n, jc, ir, mv), L2 + np.tril(L2, -1).T)
    x = rng.integers(1, 5, size=n).astype(float)
    assert np.array_equal(oracle.triplet_times(n, n, r + 1, c + 1, L[r, c], x, is_symmetric=True), S @ x)


def test_hs071_handler_formulas(oracle):
    """QPhandler::set_bounds / set_g on the analytically derived hs071 iterate."""
    nlp = problems.hs071_nlp()
    assert nlp["grad"].tolist() == [12.0, 1.0, 2.0, 11.0]
    assert nlp["c"].tolist() == [25.0, 52.0]
    lb, ub, lbA, ubA = oracle.handler_set_bounds(1.0, nlp["x_l"], nlp["x_u"], nlp["x"], nlp["c_l"], nlp["c_u"], nlp["c"])
    assert lb.tolist() == [0, -1, -1, 0, 0, 0, 0, 0]
    assert ub.tolist() == [1, 0, 0, 1, 1e18, 1e18, 1e18, 1e18]
    assert lbA.tolist() == [0.0, -12.0] and ubA[1] == -12.0 and np.isinf(ubA[0])
    g = oracle.handler_set_g(nlp["grad"], 1.0, 2)
    assert g.tolist() == [12, 1, 2, 11, 1, 1, 1, 1]
    q = problems.hs071_first_qp()
    assert np.array_equal(q.lb, lb) and np.array_equal(q.ub, ub) and np.array_equal(q.g, g)
    # [J I -I] through setStructure equals the dense assembly
    J = nlp["J"]
    jc, ir, val, _ = oracle.sphb_set_structure(2, 8, J.RowIndex, J.ColIndex, J.MatVal, [(1, 5, 2, 1.0), (1, 7, 2, -1.0)])
    assert np.array_equal(jc, q.A_jc) and np.array_equal(ir, q.A_ir) and np.array_equal(val, q.A_val)
    assert len(val) == 12
    # update_bounds does not refresh ubA (QPhandler.cpp:358-360)
    lb2, ub2, lbA2 = lb.copy(), ub.copy(), lbA.copy()
    oracle.handler_update_bounds(0.5, nlp["x_l"], nlp["x_u"], nlp["x"], nlp["c_l"], nlp["c"] + 1.0, lb2, ub2, lbA2)
    assert lb2[:4].tolist() == [0, -0.5, -0.5, 0] and ub2[:4].tolist() == [0.5, 0, 0, 0.5] and lbA2.tolist() == [-1.0, -13.0]
