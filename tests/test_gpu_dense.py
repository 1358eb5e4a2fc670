"""Dense f64 building blocks of the HBM-resident engine (restartsqp_amd/csrc/dense_la.hip) against
numpy, through the C ABI. Tolerances: GEMM 1e-13 relative to sum |a||b|; factorisations 1e-12
relative residuals (f64, n <= 700)."""
import ctypes as C
import numpy as np
import pytest

from restartsqp_amd import capi

pytestmark = pytest.mark.gpu


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def gemm(ta, tb, A, B, Cm, alpha, beta):
    m, n = Cm.shape
    k = A.shape[0] if ta else A.shape[1]
    Af, Bf, Cf = (np.array(x, dtype=np.float64, order="F", copy=True) for x in (A, B, Cm))
    rc = capi.lib().rsqp_dense_gemm(int(ta), int(tb), m, n, k, alpha, _dp(Af), Af.shape[0], _dp(Bf), Bf.shape[0], beta,
                                    _dp(Cf), Cf.shape[0], 0, None)
    assert rc == 0
    return Cf


@pytest.mark.parametrize("ta,tb", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("m,n,k", [(1, 1, 1), (64, 64, 64), (130, 70, 33), (300, 257, 129), (64, 1000, 2500), (513, 40, 64)])
def test_gemm_matches_numpy(ta, tb, m, n, k):
    rng = np.random.default_rng(m * 1000 + n * 10 + k + ta * 2 + tb)
    A = rng.normal(size=(k, m) if ta else (m, k))
    B = rng.normal(size=(n, k) if tb else (k, n))
    C0 = rng.normal(size=(m, n))
    alpha, beta = 0.7, -1.3
    got = gemm(ta, tb, A, B, C0, alpha, beta)
    opA, opB = (A.T if ta else A), (B.T if tb else B)
    want = alpha * opA @ opB + beta * C0
    scale = np.abs(opA) @ np.abs(opB) + np.abs(C0) + 1.0
    assert np.max(np.abs(got - want) / scale) < 1e-13


def test_gemm_exact_on_integers():
    """layout check with exact data: any wrong lane <-> element map changes an integer result"""
    rng = np.random.default_rng(3)
    A = rng.integers(-8, 9, size=(150, 77)).astype(float)
    B = rng.integers(-8, 9, size=(77, 201)).astype(float)
    got = gemm(0, 0, A, B, np.zeros((150, 201)), 1.0, 0.0)
    assert np.array_equal(got, A @ B)


@pytest.mark.parametrize("m,n", [(5, 3), (64, 64), (200, 130), (700, 450), (300, 300), (129, 1)])
def test_blocked_qr(m, n):
    rng = np.random.default_rng(m + n)
    B0 = rng.normal(size=(m, n))
    B = np.asfortranarray(B0.copy()); Q = np.zeros((m, m), order="F"); Ri = np.zeros((n, n), order="F")
    ndep = C.c_int(-1)
    assert capi.lib().rsqp_dense_qr(m, n, _dp(B), _dp(Q), _dp(Ri), 1e-9, C.byref(ndep), None) == 0
    assert ndep.value == 0
    R = np.triu(B[:n, :])
    assert np.max(np.abs(Q.T @ Q - np.eye(m))) < 1e-12
    assert np.max(np.abs(Q[:, :n] @ R - B0)) < 1e-12 * max(1.0, np.abs(B0).max()) * m
    if m > n:
        assert np.max(np.abs(Q[:, n:].T @ B0)) < 1e-12 * m      # null-space basis of B0'
    assert np.max(np.abs(Ri @ R - np.eye(n))) < 1e-10
    assert np.max(np.abs(np.tril(Ri, -1))) == 0.0


@pytest.mark.parametrize("m,n,scale", [(700, 450, 1.0), (1500, 1000, 1.0), (2100, 640, 1e3), (1000, 1000, 1.0)])
def test_blocked_qr_panels_by_cholesky_qr(m, n, scale):
    """Panels of 64 columns with at least 128 rows below their diagonal are factorised by Cholesky-QR twice + Householder
    reconstruction (dense_la.hip k_cholqr_pass1/2) instead of one launch per column: the SAME reflectors, so R agrees with
    LAPACK's column-by-column Householder R (same sign rule) entry by entry, not only up to row signs. `scale` spreads the
    column norms over three decades."""
    import scipy.linalg as sl
    rng = np.random.default_rng(m * n)
    B0 = rng.normal(size=(m, n)) * np.logspace(0, np.log10(scale), n)[None, :]
    B = np.asfortranarray(B0.copy()); Q = np.zeros((m, m), order="F"); Ri = np.zeros((n, n), order="F")
    ndep = C.c_int(-1)
    assert capi.lib().rsqp_dense_qr(m, n, _dp(B), _dp(Q), _dp(Ri), 1e-9, C.byref(ndep), None) == 0
    assert ndep.value == 0
    R = np.triu(B[:n, :])
    Rl = sl.qr(B0, mode="r")[0][:n]
    assert np.max(np.abs(R - Rl) / np.abs(np.diag(Rl))[:, None]) < 1e-10
    assert np.max(np.abs(Q.T @ Q - np.eye(m))) < 1e-12
    assert np.max(np.abs(Q[:, :n] @ R - B0) / np.abs(B0).max(axis=0)[None, :]) < 1e-12 * m
    assert np.max(np.abs(Ri @ R - np.eye(n))) < 1e-9


def test_blocked_qr_ill_conditioned_panel_takes_the_column_kernel():
    """A panel whose columns agree to seven digits: the Gram matrix cannot resolve it (pivot ratio 1e-14 < 1e-10), the
    factorisation is repeated with the column kernel and stays accurate; the columns are independent at eps_li = 1e-9."""
    rng = np.random.default_rng(77)
    m, n = 900, 200
    B0 = rng.normal(size=(m, n))
    B0[:, 70] = B0[:, 66] + 1e-7 * rng.normal(size=m)
    B = np.asfortranarray(B0.copy()); Q = np.zeros((m, m), order="F"); Ri = np.zeros((n, n), order="F")
    ndep = C.c_int(-1)
    assert capi.lib().rsqp_dense_qr(m, n, _dp(B), _dp(Q), _dp(Ri), 1e-9, C.byref(ndep), None) == 0
    assert ndep.value == 0
    R = np.triu(B[:n, :])
    assert np.max(np.abs(Q.T @ Q - np.eye(m))) < 1e-12
    assert np.max(np.abs(Q[:, :n] @ R - B0)) < 1e-12 * m
    assert 1e-8 < abs(R[70, 70]) < 1e-4


def test_blocked_qr_flags_dependent_columns():
    rng = np.random.default_rng(9)
    B0 = rng.normal(size=(120, 70))
    B0[:, 40] = B0[:, 3] * 2.0 - B0[:, 17]
    B = np.asfortranarray(B0.copy())
    ndep = C.c_int(-1)
    assert capi.lib().rsqp_dense_qr(120, 70, _dp(B), None, None, 1e-9, C.byref(ndep), None) == 0
    assert ndep.value >= 1


@pytest.mark.parametrize("n", [1, 7, 64, 65, 200, 513])
def test_cholesky_inverse(n):
    rng = np.random.default_rng(n)
    M = rng.normal(size=(n + 5, n))
    G0 = M.T @ M + 0.1 * np.eye(n)
    G = np.asfortranarray(G0.copy()); Gi = np.zeros((n, n), order="F")
    bad = C.c_int(-1)
    assert capi.lib().rsqp_dense_chol_inverse(n, _dp(G), _dp(Gi), 1e-10, 1e-25, C.byref(bad), None) == 0
    assert bad.value == 0
    U = np.triu(G)
    assert np.max(np.abs(U.T @ U - G0)) < 1e-11 * np.abs(G0).max()
    assert np.max(np.abs(Gi @ G0 - np.eye(n))) < 1e-8


def test_cholesky_reports_indefinite():
    rng = np.random.default_rng(1)
    M = rng.normal(size=(100, 100))
    G0 = M + M.T          # indefinite
    G = np.asfortranarray(G0.copy())
    bad = C.c_int(0)
    assert capi.lib().rsqp_dense_chol_inverse(100, _dp(G), None, 1e-10, 1e-25, C.byref(bad), None) == 0
    assert bad.value != 0
