"""One blocked Cholesky + triangular inverse + U^-1 U^-T of the size the range-space set-up of the sparse 10k x 20k configuration
factorises (rocprofv3 target: where its 30 ms go).   python tools/chol_profile_run.py [n]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from restartsqp_amd import capi
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 7656
rng = np.random.default_rng(0)
B = rng.normal(size=(n + 2300, n))
G = np.asfortranarray(B.T @ B); Gi = np.zeros((n, n), order="F")
bad = C.c_int(0); ms = C.c_float(0)
assert capi.lib().rsqp_dense_chol_inverse(n, dp(G), dp(Gi), 1e-10, 1e-25, C.byref(bad), C.byref(ms)) == 0
print("chol + inverse %d: %.1f ms, not_pd %d" % (n, ms.value, bad.value))
