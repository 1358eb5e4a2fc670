"""Diagnostic: where does the LDS-resident QP kernel spend its cycles? Builds a -DRSQP_STAMPS variant of the kernel
(tools/small_experiment.sh, never used by the product) and prints cycles per phase for block 0.
    python tools/stamp_small_kernel.py            hs071-scale batch, 8-lane Givens / TQ kernel
    python tools/stamp_small_kernel.py 69 28      the 69 x 28 members of the 512-QP hs0xx batch, four-wave kernel"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wide = len(sys.argv) > 2
env = dict(os.environ, EXPDEF="-DRSQP_SMALL_EXPERIMENT=%d" % (2 if wide else 1))
subprocess.check_call([os.path.join(ROOT, "tools", "small_experiment.sh"), "-DRSQP_STAMPS"], env=env)
os.environ["RSQP_LIB"] = os.path.join(ROOT, "restartsqp_amd", "lib", "librsqp_exp.so")
from restartsqp_amd import capi, problems
L = capi.lib()
L.rsqp_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
if wide:
    probs = [p for p in problems.hs_batch(512) if (p.nV, p.nC) == (int(sys.argv[1]), int(sys.argv[2]))]
else:
    probs = problems.hs071_scale_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 16384)
b = capi.Batch(probs)
buf = (C.c_ulonglong * 48)()
b.solve(capi.MODE_COLD, 1000)
L.rsqp_debug_stamps(buf, 1)
reps = 5
for _ in range(reps):
    b.solve(capi.MODE_COLD, 1000)
L.rsqp_debug_stamps(buf, 0)
names = {0: "prologue (zero image, stage matrices)", 2: "targets + setup_aux", 3: "step_direction", 4: "ratio_tests",
         5: "step + A x", 10: "  sd: dx_FX loop + barrier", 11: "  sd: A dx | H dx (one fused stage)", 12: "  sd: rhs loops + barrier",
         13: "  sd: Minv bA, Y wY (2 stages)", 14: "  sd: H xY, +, Z', Wz, Z (5 stages)", 15: "  sd: merge loop + barrier", 16: "  cas: row of A", 17: "  cas: Z'a | Y'a (one fused stage)", 18: "  cas: two dots", 19: "  cas: house (dot, vector)", 20: "  cas: Z v | Wz v (one fused stage)",
         21: "  cas: Z -= beta t v'", 22: "  cas: theta, Wz last column", 23: "  cas: Wz update", 24: "  cas: copy column + Minv append", 25: "  cas: add_bound tail (Y, Minv updates)",
         26: "  cas: wz_grow (remove paths incl.)", 27: "  cas: remove_constraint", 28: "  cas: remove_bound", 29: "  cas: ensure_LI (exchange)",
         6: "change_active_set (rest)", 7: "drift_correction", 8: "objective", 9: "results + image write-back"}
tot = sum(buf[k] for k in names)
nw = b.results()[0]["nWSR"]
for k, n in names.items():
    print("%-40s %9.0f cycles  %5.1f %%  (%6.0f per working-set change)" % (n, buf[k] / reps, 100.0 * buf[k] / tot, buf[k] / reps / max(nw, 1)))
print("total %.0f cycles per QP (block 0, s_memtime ticks of 100 MHz x ...); kernel %.3f ms; nWSR of QP 0: %d" % (tot / reps, b.last_solve_ms(), nw))
