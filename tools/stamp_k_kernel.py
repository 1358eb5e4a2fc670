"""Diagnostic: cycles per phase of the tableau kernel (qp_small_g.h) for block 0 of a batch of 69 x 28 members
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
names = {30: "exact A x, A'y - H x from the data (every 8th change, after exchanges / flips)",
         38: "x on bounds, drift correction, input of the product + barrier (incl. working-set bookkeeping of the change before)",
         31: "A: out = G in (LDS reads, FMAs, row sums)", 32: "A: row owners: dx / dy / A dx, ratio-test candidates (2 divisions)",
         33: "A: block argmin (wave mins, barrier, combine)", 34: "B: homotopy step by the row owners, publish row q + barrier",
         35: "C: pivot tests (curvature: |u_FR|^2)", 36: "C: independence test (|P a|, |a_FR|; residual test in the band)",
         37: "C: exchange (ratio, y shift, partner row, 2 x 2 block pivot) / rank-1 update of G"}
tot = sum(buf[k] for k in names)
nw = b.results()[0]["nWSR"]
for k, n in names.items():
    print("%-96s %9.0f ticks  %5.1f %%  (%6.0f per working-set change)" % (n, buf[k] / reps, 100.0 * buf[k] / tot, buf[k] / reps / max(nw, 1)))
print("total %.0f ticks per QP (block 0); kernel %.3f ms; nWSR of QP 0: %d; %d members" % (tot / reps, b.last_solve_ms(), nw, len(probs)))
