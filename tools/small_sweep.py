"""Throughput of the LDS-resident batch kernel for lanes-per-problem / waves-per-SIMD settings
(each setting in its own process: the launcher reads the environment once)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import numpy as np
    from restartsqp_amd import capi, problems
    n = int(sys.argv[2])
    probs = problems.hs071_scale_batch(n)
    b = capi.Batch(probs)
    b.solve(capi.MODE_COLD, 1000)
    ms = []
    for _ in range(5):
        b.solve(capi.MODE_COLD, 1000)
        ms.append(b.last_solve_ms())
    res = b.results()
    print(json.dumps({"L": os.environ.get("RSQP_SMALL_LANES"), "W": os.environ.get("RSQP_SMALL_WAVES"), "n": n,
                      "ms": min(ms), "Msolves_per_s": n / min(ms) / 1e3, "solved": sum(r["status"] == 5 for r in res), "objsum": sum(r["obj"] for r in res), "nwsr": sum(r["nWSR"] for r in res)}))
    sys.exit(0)
n = sys.argv[1] if len(sys.argv) > 1 else "16384"
for L in (64, 32, 16):
    for W in (2, 3, 4, 6):
        if W == 6 and L != 64:
            continue
        env = dict(os.environ, RSQP_SMALL_LANES=str(L), RSQP_SMALL_WAVES=str(W))
        r = subprocess.run([sys.executable, __file__, "--one", n], env=env, capture_output=True, text=True, timeout=300)
        print(r.stdout.strip() or r.stderr[-500:], flush=True)
