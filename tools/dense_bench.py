"""TFLOP/s of the f64 MFMA GEMM and time of the blocked QR on the shapes the engine uses."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from restartsqp_amd import capi
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
rng = np.random.default_rng(0)
L = capi.lib()
for (ta, tb, m, n, k) in [(0, 0, 4096, 4096, 4096), (1, 0, 4096, 4096, 4096), (1, 0, 64, 8192, 8192), (0, 0, 8192, 8192, 64), (1, 0, 64, 64, 8192), (0, 1, 2048, 2048, 2048)]:
    A = np.asfortranarray(rng.normal(size=(k, m) if ta else (m, k))); B = np.asfortranarray(rng.normal(size=(n, k) if tb else (k, n)))
    Cm = np.zeros((m, n), order="F"); ms = C.c_float(0)
    assert L.rsqp_dense_gemm(ta, tb, m, n, k, 1.0, dp(A), A.shape[0], dp(B), B.shape[0], 0.0, dp(Cm), m, 5, C.byref(ms)) == 0
    print("gemm ta=%d tb=%d %5d x %5d x %5d: %8.3f ms  %6.2f TFLOP/s" % (ta, tb, m, n, k, ms.value, 2.0 * m * n * k / ms.value / 1e9), flush=True)
for (m, n) in [(2048, 1500), (6000, 4500)]:
    B = np.asfortranarray(rng.normal(size=(m, n))); Q = np.zeros((m, m), order="F"); Ri = np.zeros((n, n), order="F")
    nd = C.c_int(0); ms = C.c_float(0)
    assert L.rsqp_dense_qr(m, n, dp(B), dp(Q), dp(Ri), 1e-9, C.byref(nd), C.byref(ms)) == 0
    print("qr + R^-1 + explicit Q  %5d x %5d: %8.2f ms, orth err %.2e" % (m, n, ms.value, np.abs(Q[:, :64].T @ Q - np.eye(m)[:64]).max()), flush=True)
