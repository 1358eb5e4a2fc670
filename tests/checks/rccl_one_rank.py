"""The C ABI's own RCCL call sites on a 1-rank communicator (all a one-GPU box can run: RCCL refuses two ranks on one
device): rsqp_rccl_unique_id / rsqp_rccl_comm_create, rsqp_batch_allgather_records (device-side packing into this rank's slot +
in-place ncclAllGather on the batch's stream), rsqp_rccl_broadcast_dev. A process WITHOUT torch, like the C++ host the entry
points are for: torch ships its own copies of the ROCm runtime libraries, and a process that loads both sets (librsqp_hip.so
binds /opt/rocm's, `import torch` brings torch/lib's) ends up with an RCCL whose HSA runtime was never initialised.
Usage (GPU box): python tests/checks/rccl_one_rank.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from restartsqp_amd import capi, problems  # noqa: E402

assert "torch" not in sys.modules
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
H2D, D2H = 1, 2


def dev_alloc(nbytes):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), nbytes) == 0
    return p


probs = problems.hs_batch(23)
b = capi.Batch(probs, device=0)
b.solve(capi.MODE_COLD, 1000)
b.test_optimality()
want = b.pack_records()
comm = capi.RcclComm(capi.rccl_unique_id(), 0, 1, 0)
per_rank, stride = 32, b.record_stride                          # 9 padding records
n = per_rank * stride
dev = dev_alloc(8 * n)
fill = np.full(n, -7.0)
assert hip.hipMemcpy(dev, fill.ctypes.data_as(C.c_void_p), 8 * n, H2D) == 0
b.allgather_records(comm, per_rank, dev.value)
got = np.zeros(n)
assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), dev, 8 * n, D2H) == 0
got = got.reshape(per_rank, stride)
assert np.array_equal(got[:23], want), "gathered records differ from the host-packed ones"
assert np.all(got[23:] == 0.0), "padding records must be zero"
assert int((got[:23, 0] == 20).sum()) == sum(1 for r in b.results() if r["status"] == 20)
try:                                                            # fewer slots than members: refused, nothing sent
    b.allgather_records(comm, 22, dev.value)
    raise SystemExit("count_per_rank < members was accepted")
except capi.RsqpError:
    pass
# broadcast of shared problem data (root = the only rank: the buffer must come back unchanged)
buf = np.arange(1000, dtype=np.float64)
d2 = dev_alloc(8000)
assert hip.hipMemcpy(d2, buf.ctypes.data_as(C.c_void_p), 8000, H2D) == 0
comm.broadcast_dev(d2.value, 8000, root=0)
back = np.zeros(1000)
assert hip.hipMemcpy(back.ctypes.data_as(C.c_void_p), d2, 8000, D2H) == 0
assert np.array_equal(back, buf)
comm.close(); b.close()
print("RCCL ONE RANK OK: %d records gathered, stride %d" % (23, stride))
