"""Diagnostic: where does the LDS-resident QP kernel spend its cycles? Builds a -DRSQP_STAMPS
variant of the library (never used by the product) and prints cycles per phase for block 0."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from restartsqp_amd import build
lib = os.path.join(ROOT, "restartsqp_amd", "lib", "librsqp_hip_stamps.so")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-DRSQP_STAMPS", "-Wno-unused-result",
       "-o", lib] + [os.path.join(build.CSRC, f) for f in build.SOURCES]
subprocess.check_call(cmd)
from restartsqp_amd import capi, problems
capi.LIB_PATH = lib
L = capi.lib()
L.rsqp_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
if len(sys.argv) > 2:   # stamp_small_kernel.py <nV> <nC>: the problems of that shape in the 512-QP hs0xx batch
    probs = [p for p in problems.hs_batch(512) if (p.nV, p.nC) == (int(sys.argv[1]), int(sys.argv[2]))]
else:
    probs = problems.hs071_scale_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 16384)
b = capi.Batch(probs)
buf = (C.c_ulonglong * 16)()
b.solve(capi.MODE_COLD, 1000)
L.rsqp_debug_stamps(buf, 1)
reps = 5
for _ in range(reps):
    b.solve(capi.MODE_COLD, 1000)
L.rsqp_debug_stamps(buf, 0)
names = {0: "prologue (zero image, stage matrices)", 2: "targets + setup_aux", 3: "step_direction", 4: "ratio_tests",
         5: "step + A x", 6: "change_active_set", 7: "drift_correction", 8: "objective", 9: "results + image write-back"}
tot = sum(buf[k] for k in names)
for k, n in names.items():
    print("%-40s %9.0f cycles  %5.1f %%" % (n, buf[k] / reps, 100.0 * buf[k] / tot))
print("total %.0f cycles per QP (block 0); kernel %.3f ms; nWSR of QP 0: %d" % (tot / reps, b.last_solve_ms(), b.results()[0]["nWSR"]))
