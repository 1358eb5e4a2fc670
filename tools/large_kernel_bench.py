"""Device time of the HBM-resident engine's streaming kernels (y = M w, y = M'x, rank-1 update) on the block shapes
of the sparse 10 000 x 20 000 configuration (n = 10 000) and of the dense 2048 x 4096 one (n = 2048).
    python3 tools/large_kernel_bench.py [n ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from restartsqp_amd import capi
L = capi.lib()
names = ["gemv_n", "gemv_t", "ger"]
for n in [int(a) for a in sys.argv[1:]] or [10000, 2048]:
    for nrows, ncols in ((n, n), (n, n // 2), (n, n // 4), (n // 2, n // 2)):
        row = []
        for kind in range(3):
            ms = C.c_float(0)
            capi.check(L.rsqp_time_large_kernel(0, n, kind, nrows, ncols, 30, C.byref(ms)))
            byts = (16.0 if kind == 2 else 8.0) * nrows * ncols
            row.append("%s %8.2f us %5.0f GB/s" % (names[kind], 1e3 * ms.value, byts / ms.value / 1e6))
        print("n %5d block %5d x %5d: %s" % (n, nrows, ncols, " | ".join(row)), flush=True)
