"""Multi-GPU batch path: independent QPs sharded over the GPUs of one node.

One process per GPU (``torch.distributed``, backend "nccl" = RCCL on ROCm; "gloo" for CPU
rehearsals). The path has NO exchange step during a solve -- every QP is independent
 -- so the only collectives are the optional broadcast of shared problem
data from rank 0 and the gather of fixed-stride result records; both are KB-MB sized and
latency bound, xGMI bandwidth is irrelevant here.
"""
import numpy as np


def shard_range(nq, rank, world):
    """Contiguous block of problems for `rank`; sizes differ by at most one."""
    base, rem = divmod(nq, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def balanced_order(problems):
    """Heterogeneous batches: sort by nV*max(nC,1) and deal round-robin so that every rank
    (and every wave of workgroups) sees a similar mix. Returns the permutation."""
    cost = np.array([p.nV * max(p.nC, 1) for p in problems])
    return np.argsort(-cost, kind="stable")


RECORD_HEAD = 4  # status, nWSR, objective, kkt


def pack_records(results, kkt, nVmax, nCmax):
    """Fixed-stride result records {status, nWSR, obj, KKT, x[nVmax], y[nVmax+nCmax], ws_b, ws_c}."""
    stride = RECORD_HEAD + 2 * nVmax + nCmax + nVmax + nCmax
    rec = np.zeros((len(results), stride))
    for k, r in enumerate(results):
        nV, nC = len(r["x"]), len(r["ws_c"])
        rec[k, 0], rec[k, 1], rec[k, 2], rec[k, 3] = r["status"], r["nWSR"], r["obj"], kkt[k]
        o = RECORD_HEAD
        rec[k, o:o + nV] = r["x"]; o += nVmax
        rec[k, o:o + nV] = r["y"][:nV]; rec[k, o + nVmax:o + nVmax + nC] = r["y"][nV:]; o += nVmax + nCmax
        rec[k, o:o + nV] = r["ws_b"]; o += nVmax
        rec[k, o:o + nC] = r["ws_c"]
    return rec


def unpack_record(rec, nV, nC, nVmax, nCmax):
    o = RECORD_HEAD
    x = rec[o:o + nV].copy(); o += nVmax
    y = np.concatenate([rec[o:o + nV], rec[o + nVmax:o + nVmax + nC]]); o += nVmax + nCmax
    ws_b = rec[o:o + nV].astype(np.int32); o += nVmax
    ws_c = rec[o:o + nC].astype(np.int32)
    return dict(status=int(rec[0]), nWSR=int(rec[1]), obj=float(rec[2]), kkt=float(rec[3]), x=x, y=y, ws_b=ws_b,
                ws_c=ws_c)


def gather_records(rec, dist, device="cpu"):
    """All-gather the per-rank record blocks (ranks may hold different counts)."""
    import torch
    world = dist.get_world_size()
    n_local = torch.tensor([rec.shape[0]], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    nmax = int(max(int(c.item()) for c in counts))
    pad = torch.zeros((nmax, rec.shape[1]), dtype=torch.float64, device=device)
    pad[:rec.shape[0]] = torch.from_numpy(rec).to(device)
    parts = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return np.concatenate([p[:int(c.item())].cpu().numpy() for p, c in zip(parts, counts)], axis=0)


def solve_sharded(problems, solve_fn, dist=None, device="cpu"):
    """Shard `problems` over the ranks, solve the local block with `solve_fn(list) ->
    (results, kkt)`, gather everything everywhere. Without `dist` runs single process."""
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    lo, hi = shard_range(len(problems), rank, world)
    results, kkt = solve_fn(problems[lo:hi]) if hi > lo else ([], [])
    nVmax = max(p.nV for p in problems); nCmax = max(p.nC for p in problems)
    rec = pack_records(results, kkt, nVmax, nCmax)
    if dist is not None and world > 1:
        rec = gather_records(rec, dist, device)
    return [unpack_record(rec[k], problems[k].nV, problems[k].nC, nVmax, nCmax) for k in range(len(problems))]
