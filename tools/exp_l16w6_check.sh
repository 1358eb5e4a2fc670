#!/bin/bash
# The packed L=16 build at 6 waves per SIMD (80 VGPRs, ~200 spilled values) that round 1 reported as returning wrong
# results: compare it with the shipped L=16 W=2 kernel on all three sets of tools/small_pack_check.py (cold + hot start).
cd "$(dirname "$0")/.."
export RSQP_SMALL_LANES=16 RSQP_SMALL_ENGINE=0
./tools/small_experiment.sh -DRSQP_EXP_L16W6 "$@" > /dev/null 2>&1 || { echo "build failed"; exit 1; }
for kind in hs071 hs random; do
    python tools/small_pack_check.py --one $kind /tmp/ref.npz || exit 1
    RSQP_LIB=$PWD/restartsqp_amd/lib/librsqp_exp.so python tools/small_pack_check.py --one $kind /tmp/exp.npz || { echo "run failed: $kind"; continue; }
    python - "$kind" <<'PY'
import sys, numpy as np
a, b = dict(np.load("/tmp/ref.npz")), dict(np.load("/tmp/exp.npz"))
bad = {k: int((~((a[k] == b[k]) | (np.isnan(a[k]) & np.isnan(b[k])))).sum()) for k in a}
print("set %-7s L=16 W=6 (80 VGPRs) vs L=16 W=2:" % sys.argv[1], "IDENTICAL" if not any(bad.values()) else "DIFFERS %s" % {k: v for k, v in bad.items() if v},
      "(%d problems, nWSR cold %d hot %d)" % (len(a["status_c"]), a["nWSR_c"].sum(), a["nWSR_h"].sum()))
PY
done
