"""Diagnostic: cycles per phase of the explicit-KKT-inverse kernel (qp_small_k.h) for block 0 of a batch of 69 x 28 members
of the hs0xx batch (-DRSQP_STAMPS build via tools/small_experiment.sh; never used by the product).
    python tools/stamp_k_kernel.py [nV nC]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
env = dict(os.environ, EXPDEF="-DRSQP_SMALL_EXPERIMENT=2")
subprocess.check_call([os.path.join(ROOT, "tools", "small_experiment.sh"), "-DRSQP_STAMPS"] + os.environ.get("EXTRA_DEFS", "").split(), env=env)
os.environ["RSQP_LIB"] = os.path.join(ROOT, "restartsqp_amd", "lib", "librsqp_exp.so")
from restartsqp_amd import capi, problems
L = capi.lib()
L.rsqp_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
shape = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (69, 28)
QUIET = os.environ.get("STAMP_QUIET") == "1"
probs = [p for p in problems.hs_batch(512) if (p.nV, p.nC) == shape]
b = capi.Batch(probs)
b.set_keep_state(False)
buf = (C.c_ulonglong * 48)()
b.solve(capi.MODE_COLD, 1000)
L.rsqp_debug_stamps(buf, 1)
reps = 5
for _ in range(reps):
    b.solve(capi.MODE_COLD, 1000)
L.rsqp_debug_stamps(buf, 0)
names = {30: "exact drift + rhs products (every 8th change)", 31: "M r -> dx_FR, dy_AC", 32: "A dx | H dx - A'dy -> dy_FX", 33: "ratio: decode", 43: "ratio: candidates (loads, division)", 44: "ratio: block argmin (wave mins, barrier, combine)",
         34: "homotopy step + k of the change", 37: "u = M k", 38: "independence test (A'xi, dots)", 39: "exchange (ratio, y shift, partner k)",
         41: "pivot (dots, division)", 42: "rank-1 update of M", 40: "working set + element-wise drift / rhs + barrier"}
tot = sum(buf[k] for k in names)
nw = b.results()[0]["nWSR"]
for k, n in names.items():
    print("%-40s %9.0f ticks  %5.1f %%  (%6.0f per working-set change)" % (n, buf[k] / reps, 100.0 * buf[k] / tot, buf[k] / reps / max(nw, 1)))
print("total %.0f ticks per QP (block 0); kernel %.3f ms; nWSR of QP 0: %d; %d members" % (tot / reps, b.last_solve_ms(), nw, len(probs)))
