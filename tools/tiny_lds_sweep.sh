#!/bin/bash
cd "$(dirname "$0")/.." 2>/dev/null || cd /root/repo
for tb in 64 128 256; do
  tools/tiny_experiment.sh -DTINY_BLOCK=$tb > /dev/null || exit 1
  for L in 0 1; do
  RSQP_TINY_LDS=$L RSQP_LIB=restartsqp_amd/lib/librsqp_exp.so python3 - <<PY
import sys; sys.path.insert(0, ".")
import numpy as np
from restartsqp_amd import capi, problems
b = capi.Batch(problems.hs071_scale_batch(65536)); b.set_keep_state(False)
b.solve(capi.MODE_COLD, 1000)
ms = []
for _ in range(30):
    b.solve(capi.MODE_COLD, 1000, sync=True); ms.append(b.last_solve_ms())
print("TINY_BLOCK=$tb lds=$L median %.4f ms  min %.4f  -> %.0f M solves/s" % (np.median(ms), min(ms), 65536 / np.median(ms) / 1e3))
PY
  done
done
