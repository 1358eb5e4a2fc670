#!/bin/bash
# Tuning: the hs071-scale tableau kernel (qp_tiny.hip) with 64 / 128 / 256 / 512 threads per workgroup; headline batch time each.
cd "$(dirname "$0")/.."
for tb in 64 128 256 512; do
  tools/tiny_experiment.sh -DTINY_BLOCK=$tb > /dev/null || exit 1
  RSQP_LIB=restartsqp_amd/lib/librsqp_exp.so python3 - <<PY
import sys; sys.path.insert(0, ".")
import numpy as np
from restartsqp_amd import capi, problems
b = capi.Batch(problems.hs071_scale_batch(65536)); b.set_keep_state(False)
b.solve(capi.MODE_COLD, 1000)
ms = []
for _ in range(30):
    b.solve(capi.MODE_COLD, 1000, sync=True); ms.append(b.last_solve_ms())
print("TINY_BLOCK=$tb  median %.4f ms  min %.4f  -> %.0f M solves/s" % (np.median(ms), min(ms), 65536 / np.median(ms) / 1e3))
PY
done
