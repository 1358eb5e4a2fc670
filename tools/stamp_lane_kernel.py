"""Diagnostic: cycles per part of the lane-per-problem kernel (qp_lane.hip) for wave 0 of the headline batch
(-DRSQP_STAMPS build via tools/lane_experiment.sh; never used by the product).   python tools/stamp_lane_kernel.py [nq]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if not os.environ.get("NO_BUILD"):
    subprocess.check_call([os.path.join(ROOT, "tools", "lane_experiment.sh"), "-DRSQP_STAMPS"] + os.environ.get("EXTRA_DEFS", "").split())
os.environ["RSQP_LIB"] = os.path.join(ROOT, "restartsqp_amd", "lib", "librsqp_exp.so")
os.environ["RSQP_LANE"] = "1"
from restartsqp_amd import capi, problems
L = capi.lib()
L.rsqp_debug_lane_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
# STAMP_MODE=cold (default; the headline launch) | keep (cold start, state written back)
which = os.environ.get("STAMP_MODE", "cold")
probs = problems.hs071_scale_batch(nq)
b = capi.Batch(probs)
b.set_keep_state(which != "cold")
buf = (C.c_ulonglong * 16)()
b.solve(capi.MODE_COLD, 1000)
assert b.last_kernel() == 2
L.rsqp_debug_lane_stamps(buf, 1)
reps = 5
for k in range(reps):
    b.solve(capi.MODE_COLD, 1000)
L.rsqp_debug_lane_stamps(buf, 0)
names = {10: "staging: loads issued, pattern pointers", 11: "staging: loads arrived, dropped into LDS", 12: "staging: own vectors, A scattered",
         13: "staging: H scattered and read", 0: "staging: rest", 1: "set-up (auxiliary QP)", 2: "homotopy: tail of the last pass",
         5: "homotopy: x on bounds, refresh, drift, input", 6: "homotopy: out = G in", 7: "homotopy: dx / dy, candidates (divisions)",
         8: "homotopy: decode, step", 9: "homotopy: change (row fetch, tests, pivot, working set)",
         3: "refinement step + exact products + objective", 15: "state block written back", 4: "results to HBM"}
tot = sum(buf[k] for k in names)
for k, n in names.items():
    print("%-60s %9.0f cycles  %5.1f %%" % (n, buf[k] / reps, 100.0 * buf[k] / max(tot, 1)))
print("total %.0f cycles per wave of 64 QPs (wave 0); kernel %.4f ms for %d QPs" % (tot / reps, b.last_solve_ms(), nq))
