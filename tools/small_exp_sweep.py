"""Tuning sweep of the 8-lane LDS kernel built by tools/small_experiment.sh (RSQP_LIB must point to it):
waves per SIMD x compile-time shape x keep_state, each setting in its own process."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    from restartsqp_amd import capi, problems
    n = int(sys.argv[2])
    probs = problems.hs071_scale_batch(n)
    b = capi.Batch(probs)
    b.set_keep_state(int(os.environ.get("KEEP", "1")))
    b.solve(capi.MODE_COLD, 1000)
    ms = []
    for _ in range(20):
        b.solve(capi.MODE_COLD, 1000)
        ms.append(b.last_solve_ms())
    res = b.results()
    ms.sort()
    print(json.dumps({"W": os.environ.get("RSQP_SMALL_WAVES"), 
                      "keep": os.environ.get("KEEP"), "n": n, "ms_median": ms[len(ms) // 2], "ms_min": ms[0],
                      "Msolves_per_s": n / ms[len(ms) // 2] / 1e3, "solved": sum(r["status"] == 20 for r in res),
                      "objsum": sum(r["obj"] for r in res), "nwsr": sum(r["nWSR"] for r in res)}))
    sys.exit(0)
n = sys.argv[1] if len(sys.argv) > 1 else "65536"
for noshape in ("1", "0"):
    for nospread in ("1", "0"):
        for W in ("2", "3"):
            for keep in ("1", "0"):
                env = dict(os.environ, RSQP_SMALL_WAVES=W, RSQP_SMALL_NOSHAPE=noshape, RSQP_SMALL_NOSPREAD=nospread, KEEP=keep)
                r = subprocess.run([sys.executable, __file__, "--one", n], env=env, capture_output=True, text=True, timeout=300)
                print("noshape", noshape, "nospread", nospread, r.stdout.strip() or r.stderr[-800:], flush=True)
