"""The matrix-core set-up of the range-space paths as a stand-alone workload (rocprofv3 / --pmc target): the Gram matrix G = B'B
(m x n -> n x n; the engine computes the upper tiles only, this entry the full product) by k_dgemm, then blocked Cholesky +
triangular inverse + U^-1 U^-T (rsqp_dense_chol_inverse) -- what hotstart(H, g, A, ..) / init(.., x0, y0, bounds) re-factorise
(reference call sites src/qpOASESInterface.cpp:184,197,204-206).   python tools/setup_profile_run.py [m n]
Defaults: 5000 x 3840, half of the sparse 10k configuration's 9 980 x 7 656: a counter pass serialises every launch."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from restartsqp_amd import capi
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
m = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
rng = np.random.default_rng(0)
B = np.asfortranarray(rng.normal(size=(m, n)))
G = np.zeros((n, n), order="F"); Gi = np.zeros((n, n), order="F")
ms = C.c_float(0)
assert capi.lib().rsqp_dense_gemm(1, 0, n, n, m, 1.0, dp(B), m, dp(B), m, 0.0, dp(G), n, 1, C.byref(ms)) == 0
print("gram %d x %d x %d: %.2f ms = %.1f TFLOP/s" % (n, n, m, ms.value, 2.0 * n * n * m / (ms.value * 1e-3) / 1e12))
bad = C.c_int(0); ms2 = C.c_float(0)
assert capi.lib().rsqp_dense_chol_inverse(n, dp(G), dp(Gi), 1e-10, 1e-25, C.byref(bad), C.byref(ms2)) == 0
print("chol + inverse %d: %.2f ms = %.1f TFLOP/s, not_pd %d" % (n, ms2.value, float(n) ** 3 / (ms2.value * 1e-3) / 1e12, bad.value))
