"""Multi-process path (world_size 2, gloo on CPU): sharding + gather of result records.
The HIP engine cannot run here, so the per-rank solve is played by the oracle -- this test
checks the partition / gather logic, not the kernels."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT
from restartsqp_amd import parallel, problems


def test_shard_range_covers_everything():
    for nq in (1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(nq, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == nq
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert parallel.shard_range(512, 3, 8) == (192, 256)     # 64 QPs per GPU at 8 GPUs


def test_balanced_order_is_permutation():
    ps = problems.hs_batch(37)
    perm = parallel.balanced_order(ps)
    assert sorted(perm.tolist()) == list(range(37))


def test_balanced_shards_spread_the_expensive_problems():
    """BASELINE configs[4]: 512 mixed hs0xx QPs on 8 GPUs. Contiguous blocks put whatever the input order
    holds on a rank; the balanced deal gives every rank the same number of the largest (69 x 28) members."""
    ps = problems.hs_batch(512)
    shards = parallel.balanced_shards(ps, 8)
    assert sorted(np.concatenate(shards).tolist()) == list(range(512)) and all(len(s) == 64 for s in shards)
    big = [sum(1 for k in s if ps[k].nV == 69) for s in shards]
    assert max(big) - min(big) <= 1 and sum(big) == sum(1 for p in ps if p.nV == 69)
    cost = [sum(ps[k].nV * max(ps[k].nC, 1) for k in s) for s in shards]
    assert max(cost) - min(cost) <= 69 * 28      # within one largest member (14 of them do not divide by 8)
    for s in shards:    # largest first inside a shard too
        c = [ps[k].nV * max(ps[k].nC, 1) for k in s]
        assert c == sorted(c, reverse=True)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    probs = problems.hs_batch(21)

    def solve_fn(block):
        res, kkt = [], []
        for p in block:
            qp = O.OracleQP(p.nV, p.nC)
            qp.set_A_csc(p.A_jc, p.A_ir, p.A_val); qp.set_H_csc(p.H_jc, p.H_ir, p.H_val)
            rc, n = qp.init(p.g, p.lb, p.ub, p.lbA, p.ubA, 1000)
            res.append(dict(x=qp.x, y=qp.y, ws_b=qp.ws_bounds, ws_c=qp.ws_constraints, status=qp.exitflag(), nWSR=n,
                            obj=qp.objective))
            kkt.append(0.0)
        return res, kkt

    ref, _ = solve_fn(probs)
    same = lambda out, ref: all(
        np.array_equal(a["x"], b["x"]) and np.array_equal(a["y"], b["y"]) and a["nWSR"] == b["nWSR"] and
        np.array_equal(a["ws_b"], b["ws_b"]) and np.array_equal(a["ws_c"], b["ws_c"]) for a, b in zip(out, ref))
    out = parallel.solve_sharded(probs, solve_fn, dist)
    ok = same(out, ref)
    # heterogeneous batch dealt largest-first round-robin: same answers, problem by problem
    ok = ok and same(parallel.solve_sharded(probs, solve_fn, dist, balance=True), ref)
    # parameter scan: only rank 0 knows the base problem; the broadcast hands it to rank 1
    base = problems.random_qp(np.random.default_rng(3), 9, 5) if rank == 0 else None
    if rank == 0:
        base.lbA[0] = -np.inf       # infinities survive the broadcast
    member = lambda b, k: problems.perturb(np.random.default_rng(1000 + k), b)
    scan = parallel.parameter_scan(base, 11, member, solve_fn, dist)
    full = problems.random_qp(np.random.default_rng(3), 9, 5)
    full.lbA[0] = -np.inf
    ref_scan, _ = solve_fn([member(full, k) for k in range(11)])
    ok = ok and same(scan, ref_scan) and len(scan) == 11
    q.put((rank, ok, len(out)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gather():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    assert got == [(0, True, 21), (1, True, 21)]


def _hip_worker(rank, world, port, q):
    """two ranks, BOTH on device 0 (a one-GPU box), collectives over gloo on host tensors: the solver is the HIP engine
    (capi.Batch through the C ABI), the sharding / packing / gather logic is the product's parallel.solve_sharded"""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from restartsqp_amd import capi
    probs = problems.hs_batch(96)

    def solve_fn(block):
        b = capi.Batch(block, device=0)
        b.set_keep_state(False)
        b.solve(capi.MODE_COLD, 1000)
        res = b.results()
        ok, kkt = b.test_optimality()
        b.close()
        return res, list(kkt)

    out = parallel.solve_sharded(probs, solve_fn, dist, balance=True)
    q.put((rank, [(r["status"], r["nWSR"], r["ws_b"].tolist(), r["ws_c"].tolist(), r["x"].tolist()) for r in out]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_hip_engine_over_gloo(capi, oracle):
    """The N > 1 path with the HIP engine as the per-rank solver (a one-GPU box: both ranks share device 0, gloo carries the
    records): every rank ends with every member's answer, identical on both ranks and to the oracle. The ranks are child
    processes started BEFORE anything here touches the GPU in them (spawn), never an exec from a GPU process."""
    import torch.multiprocessing as mp
    from conftest import oracle_cold
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_hip_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(60)
    assert sorted(got) == [0, 1] and got[0] == got[1] and len(got[0]) == 96
    for p, (status, nwsr, wb, wc, x) in zip(problems.hs_batch(96), got[0]):
        qp, rc, n = oracle_cold(oracle, p)
        assert status == qp.exitflag() and nwsr == n and wb == qp.ws_bounds.tolist() and wc == qp.ws_constraints.tolist()
        assert np.abs(np.array(x) - qp.x).max() <= 1e-9 * max(1.0, np.abs(qp.x).max())


@pytest.mark.gpu
def test_native_rccl_allgather_of_records_one_rank():
    """The C ABI's own RCCL call sites (rsqp_rccl_comm_create + rsqp_batch_allgather_records: device-side packing into this rank's
    slot, in-place ncclAllGather on the batch's stream; rsqp_rccl_broadcast_dev) on a 1-rank communicator -- all a one-GPU box can
    run. In a child process WITHOUT torch (tests/checks/rccl_one_rank.py says why): the records must equal the host-packed ones,
    padding records must be zero, too few slots are refused."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "checks", "rccl_one_rank.py")], capture_output=True, text=True,
                       timeout=300, cwd=ROOT)
    assert r.returncode == 0 and "RCCL ONE RANK OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
