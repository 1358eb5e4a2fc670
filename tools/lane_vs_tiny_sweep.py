"""Tuning: the two hs071-scale batch kernels by batch size -- 8 lanes per problem (qp_tiny.hip) against one lane per problem
(qp_lane.hip): average launch time over 50 back-to-back cold solves each (HIP events on the launch stream).
    python tools/lane_vs_tiny_sweep.py [sizes ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from restartsqp_amd import capi, problems
sizes = [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384, 24576, 32768, 49152, 65536, 98304, 131072]
probs = problems.hs071_scale_batch(max(sizes))
for n in sizes:
    row = []
    for lane in ("0", "1"):
        os.environ["RSQP_LANE"] = lane
        b = capi.Batch(probs[:n])
        b.set_keep_state(False)
        for _ in range(5):
            b.solve(capi.MODE_COLD, 1000, sync=False)
        capi.check(capi.lib().rsqp_batch_sync(b._h))
        b.timer_start()
        for _ in range(50):
            b.solve(capi.MODE_COLD, 1000, sync=False)
        ms = b.timer_stop_ms() / 50
        assert b.last_kernel() == (2 if lane == "1" else 1)
        row.append(ms)
        b.close()
    print("nq %7d   8 lanes per QP %.4f ms (%.0f M/s)   one lane per QP %.4f ms (%.0f M/s)" % (n, row[0], n / row[0] / 1e3, row[1], n / row[1] / 1e3))
