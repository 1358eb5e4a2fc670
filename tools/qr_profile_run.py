"""One blocked Householder QR + explicit Q + R^-1 of the size the sparse 10k x 20k configuration
re-factorises on a hot start with new matrices (rocprofv3 target of tools/round_profiles.sh)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from restartsqp_amd import capi
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
rng = np.random.default_rng(0)
m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10000, 7670)
B = np.asfortranarray(rng.normal(size=(m, n))); Q = np.zeros((m, m), order="F"); Ri = np.zeros((n, n), order="F")
nd = C.c_int(0); ms = C.c_float(0)
assert capi.lib().rsqp_dense_qr(m, n, dp(B), dp(Q), dp(Ri), 1e-9, C.byref(nd), C.byref(ms)) == 0
print("qr %d x %d: %.1f ms" % (m, n, ms.value))
