"""Check that every lanes-per-problem / waves setting of the batch kernel returns identical
results (bitwise: the arithmetic order does not depend on the packing)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import numpy as np
    from restartsqp_amd import capi, problems
    kind, out = sys.argv[2], sys.argv[3]
    if kind == "hs071":
        probs = problems.hs071_scale_batch(4099)
    elif kind == "hs":
        probs = problems.hs_batch(1500)
    else:
        rng = np.random.default_rng(5)
        probs = [problems.random_qp(rng, int(rng.integers(2, 15)), int(rng.integers(1, 12))) for k in range(1000)]
        probs += [problems.degenerate_qp(rng, k % 5) for k in range(200)]
    b = capi.Batch(probs)
    b.solve(capi.MODE_COLD, 1000)
    res = b.results()
    # hot start on perturbed vectors
    prng = np.random.default_rng(11)
    pert = [problems.perturb(prng, p, 0.05) for p in probs]
    b.set_vectors_from(pert)
    b.solve(capi.MODE_HOT_VECTORS, 1000)
    res2 = b.results()
    cat = lambda rs, k: np.concatenate([np.atleast_1d(np.asarray(r[k], dtype=float)) for r in rs])
    np.savez(out, **{k + s: cat(rs, k) for s, rs in (("_c", res), ("_h", res2)) for k in ("x", "y", "ws_b", "ws_c", "status", "nWSR", "obj")})
    sys.exit(0)
import numpy as np
quick = "--quick" in sys.argv
bad = 0
for kind in (("hs071", "random") if quick else ("hs071", "hs", "random")):
    ref = None
    variants = ((64, 4), (32, 2), (16, 2)) if quick else ((64, 4), (64, 6), (32, 2), (32, 4), (16, 2), (16, 3), (16, 4))
    if kind == "hs071":
        variants = variants + ((8, 2),)    # 8 lanes per problem: only batches with nV, nC <= 8
    for L, W in variants:
        env = dict(os.environ, RSQP_SMALL_LANES=str(L), RSQP_SMALL_WAVES=str(W))
        out = "/tmp/pack_%s_%d_%d.npz" % (kind, L, W)
        r = subprocess.run([sys.executable, __file__, "--one", kind, out], env=env, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            print(kind, L, W, "FAILED", r.stderr[-400:]); bad += 1; continue
        d = dict(np.load(out))
        if ref is None:
            ref = d
            print(kind, "reference L=%d W=%d: nWSR cold %d hot %d, solved %d/%d" % (L, W, d["nWSR_c"].sum(), d["nWSR_h"].sum(), (d["status_c"] == 5).sum(), len(d["status_c"])), flush=True)
            continue
        diffs = [k for k in ref if not np.array_equal(ref[k], d[k], equal_nan=True)]
        nan = [k for k in d if np.isnan(d[k]).any()]
        print(kind, "L=%d W=%d" % (L, W), "identical" if not diffs else "DIFFERS in %s" % diffs, ("NaN in %s" % nan) if nan else "", flush=True)
        if diffs:
            bad += 1
            for k in diffs[:3]:
                idx = np.flatnonzero(~((ref[k] == d[k]) | (np.isnan(ref[k]) & np.isnan(d[k]))))
                print("   ", k, "first idx", idx[:5], ref[k][idx[:3]], d[k][idx[:3]])
print("BAD" if bad else "ALL IDENTICAL")
sys.exit(1 if bad else 0)
