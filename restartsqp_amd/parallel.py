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


def balanced_shards(problems, world):
    """Heterogeneous batches (BASELINE configs[4]: 4x0 ... 69x28 in one batch): the launch of a rank lasts as
    long as its slowest members, so the expensive problems are dealt round-robin, largest first. Returns
    one index array per rank (each sorted largest-first as well, so that the waves that start first hold
    the long-running problems). Deterministic: every rank computes the same partition."""
    order = balanced_order(problems)
    shards = [[] for _ in range(world)]
    for j, k in enumerate(order):           # snake deal: 0..W-1, W-1..0, ... evens out the sums as well
        rnd, pos = divmod(j, world)
        shards[pos if rnd % 2 == 0 else world - 1 - pos].append(k)
    return [np.array(s, dtype=np.int64) for s in shards]


def solve_sharded(problems, solve_fn, dist=None, device="cpu", balance=False):
    """Shard `problems` over the ranks, solve the local block with `solve_fn(list) ->
    (results, kkt)`, gather everything everywhere. Without `dist` runs single process.
    balance=False: contiguous blocks (shard_range); True: balanced_shards (heterogeneous sizes)."""
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    if balance:
        shards = balanced_shards(problems, world)
    else:
        shards = [np.arange(*shard_range(len(problems), r, world)) for r in range(world)]
    mine = shards[rank]
    results, kkt = solve_fn([problems[k] for k in mine]) if len(mine) else ([], [])
    nVmax = max(p.nV for p in problems); nCmax = max(p.nC for p in problems)
    rec = pack_records(results, kkt, nVmax, nCmax)
    if dist is not None and world > 1:
        rec = gather_records(rec, dist, device)
    where = np.concatenate(shards)           # record row j belongs to problem where[j]
    row_of = np.empty(len(problems), np.int64)
    row_of[where] = np.arange(len(problems))
    return [unpack_record(rec[row_of[k]], problems[k].nV, problems[k].nC, nVmax, nCmax) for k in range(len(problems))]


# ------------------------------------------------------------------------------------
# parameter scans: shared structure from rank 0 (north_star: "RCCL broadcast/gather")
# ------------------------------------------------------------------------------------
def broadcast_problem(q, dist, device="cpu", src=0):
    """Broadcast one QP -- sizes, CSC patterns and values of A and H, the five vectors -- from rank `src`
    to every rank: the set-up step of a parameter scan, where all members share the matrices and each
    rank then perturbs / loads only its own vectors (SURVEY 8(e) "Collectives"). Three collectives:
    a 4-int header, one int32 payload (jc / ir arrays), one float64 payload. `q` is a QPData on `src`
    and ignored elsewhere. Returns a QPData on every rank."""
    import torch
    from .qpdump import QPData
    rank = dist.get_rank()
    head = torch.zeros(4, dtype=torch.int64, device=device)
    if rank == src:
        head = torch.tensor([q.nV, q.nC, len(q.A_val), len(q.H_val)], dtype=torch.int64, device=device)
    dist.broadcast(head, src)
    nV, nC, annz, hnnz = (int(v) for v in head.tolist())
    ni, nd = 2 * (nV + 1) + annz + hnnz, annz + hnnz + 3 * nV + 2 * nC
    if rank == src:
        ints = np.concatenate([q.A_jc, q.A_ir, q.H_jc, q.H_ir]).astype(np.int32)
        dbls = np.concatenate([q.A_val, q.H_val, q.g, q.lb, q.ub, q.lbA, q.ubA]).astype(np.float64)
        # +-inf bounds travel as they are (IEEE); nothing is clamped here
        ti, td = torch.from_numpy(ints).to(device), torch.from_numpy(dbls).to(device)
    else:
        ti, td = torch.zeros(ni, dtype=torch.int32, device=device), torch.zeros(nd, dtype=torch.float64, device=device)
    dist.broadcast(ti, src)
    dist.broadcast(td, src)
    ints, dbls = ti.cpu().numpy(), td.cpu().numpy()
    o = 0
    A_jc = ints[o:o + nV + 1]; o += nV + 1
    A_ir = ints[o:o + annz]; o += annz
    H_jc = ints[o:o + nV + 1]; o += nV + 1
    H_ir = ints[o:o + hnnz]
    parts, o = [], 0
    for n in (annz, hnnz, nV, nV, nV, nC, nC):
        parts.append(dbls[o:o + n].copy()); o += n
    A_val, H_val, g, lb, ub, lbA, ubA = parts
    return QPData(nV, nC, H_jc.copy(), H_ir.copy(), H_val, A_jc.copy(), A_ir.copy(), A_val, g, lb, ub, lbA, ubA,
                  name=getattr(q, "name", "") if rank == src else "broadcast")


def parameter_scan(base, nq, perturb_fn, solve_fn, dist=None, device="cpu"):
    """Parameter scan over `nq` members that share the matrices of `base` (known on rank 0 only):
    broadcast the structure, let every rank build the members of its contiguous shard with
    `perturb_fn(base, k)` (k = global member index: seeded, so any rank can build any member),
    solve the shard, gather the records. Returns the per-member records on every rank."""
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    if dist is not None and world > 1:
        base = broadcast_problem(base, dist, device)
    lo, hi = shard_range(nq, rank, world)
    members = [perturb_fn(base, k) for k in range(lo, hi)]
    results, kkt = solve_fn(members) if members else ([], [])
    rec = pack_records(results, kkt, base.nV, base.nC)
    if dist is not None and world > 1:
        rec = gather_records(rec, dist, device)
    return [unpack_record(rec[k], base.nV, base.nC, base.nV, base.nC) for k in range(nq)]
