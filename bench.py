#!/usr/bin/env python3
"""bench.py -- QP-subproblem solves/sec of the MI355X-native engine (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: every rank solves its shard of
independent hs071-scale QPs (the derived hs071 first QP and seeded 1 % perturbations of it,
8 variables x 2 constraints through the QPhandler formulation) from a COLD start with ONE
launch of the LDS-resident active-set kernel, inputs already resident in HBM. Shards are
independent (no data-path collective) -> weak scaling; `value` = QPs of all ranks / max time.

The JSON line also carries
  roofline       -- the dominant kernel of the timed region (the batched QP kernel): algorithmic
                    HBM bytes per launch / its HIP-event duration. The kernel is LDS/latency
                    bound, so the fraction is small by nature; see DESIGN.md.
  roofline_spmv  -- the n=10k x m=20k, 200k-nnz Jacobian product A'y (SpHbMat::transposed_times)
                    batched over distinct matrices (> 2x the 256 MiB Infinity Cache), the
                    kernel BASELINE.json's roofline target names; measured outside the timed region.
  cpu_baseline   -- the CPU oracle (a port: qpOASES cannot be built here) on one host core,
                    bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def qp_algorithmic_bytes(q):
    """Bytes one cold solve has to move through HBM: CSC of A and H (8 B value + 4 B index per
    entry, 4 B per column pointer), the five data vectors, and the result record
    (x, y, working set, status / nWSR / objective)."""
    nV, nC = q.nV, q.nC
    inp = 12 * len(q.A_val) + 4 * (nV + 1) + 12 * len(q.H_val) + 4 * (nV + 1) + 8 * (3 * nV + 2 * nC)
    out = 8 * nV + 8 * (nV + nC) + 4 * (nV + nC) + 4 + 4 + 8
    return inp + out


def pmc_traffic(substr, fetch_factor):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE
    are collected in separate runs and reported in KiB). MI355X_MICROARCH.md: on gfx950
    FETCH_SIZE counts half the bytes of wide coalesced streaming reads -> fetch_factor 2 for
    the streaming SpMV; WRITE_SIZE is exact. Returns None when no profile is committed."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))
    if not paths:
        return None
    d = json.load(open(paths[-1]))   # the newest committed PMC pass
    for k, v in d.items():
        if substr in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            return fetch_factor * v["FETCH_SIZE"]["mean_per_dispatch"] * 1024 + v["WRITE_SIZE"]["mean_per_dispatch"] * 1024
    return None


def spmv_roofline(capi, problems, nbatch, repeats):
    n, m, nnz = 10000, 20000, 200000
    jc, ir, rng = problems.sparse_pattern(n, m, nnz)
    plan = capi.SpmvPlan(m, n, jc, ir, nbatch)
    vals = rng.normal(size=(nbatch, nnz))
    plan.upload(vals, rng.normal(size=(nbatch, n)), transposed=False)
    plan.upload(None, rng.normal(size=(nbatch, m)), transposed=True)
    out = {}
    for name, tr, bytes_one in (("ATy_csc", True, 12 * nnz + 4 * (n + 1) + 8 * n + 8 * m),
                                ("Ax_csr", False, 12 * nnz + 4 * (m + 1) + 8 * m + 8 * n)):
        plan.run(tr, 2)
        ms = plan.run(tr, repeats)
        gbs = bytes_one * nbatch / (ms * 1e-3) / 1e9
        out[name] = {"ms_per_launch": ms, "bytes_per_launch": bytes_one * nbatch, "achieved": gbs}
    best = out["ATy_csc"]
    # bytes the kernel actually streams: the plan keeps 16-bit copies of the index arrays when
    # both dimensions are < 65536 (10 B instead of 12 B per entry)
    streamed = (10 * nnz + 4 * (n + 1) + 8 * n + 8 * m) * nbatch
    res = {"kernel": "csx_ldsvec_spmv_pipe2 (A'y on CSC = SpHbMat::transposed_times; input vector resident in LDS)",
           "bound": "hbm", "achieved": best["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": best["achieved"] / HBM_PEAK_GBS,
           "traffic": pmc_traffic("csx_ldsvec_spmv_pipe2<4, 3", 2.0) if nbatch == 256 else None,
           "matrices_per_launch": nbatch,
           "bytes_per_matrix": 12 * nnz + 4 * (n + 1) + 8 * n + 8 * m, "ms_per_launch": best["ms_per_launch"],
           "streamed_bytes_per_launch_est": streamed, "raw_stream_GBs_est": streamed / (best["ms_per_launch"] * 1e-3) / 1e9,
           "Ax_csr_GBs": out["Ax_csr"]["achieved"], "Ax_csr_frac": out["Ax_csr"]["achieved"] / HBM_PEAK_GBS}
    plan.close()
    return res


def large_configs(capi, problems, hot_steps=6):
    """BASELINE configs 3 and 4 on the HBM-resident engine (outside the timed region): cold
    solve of the dense 2048 x 4096 QP, cold solve + warm-started sequence of the sparse
    10 000 x 20 000 QP ("wall-clock per SQP iteration, n=10k sparse"), and a mid-size dense QP
    solved by both the GPU engine and the CPU oracle."""
    import oracle as O
    out = {}

    def load(q):
        s = capi.Solver(q.nV, q.nC)
        s.set_A_csc(q.A_jc, q.A_ir, q.A_val); s.set_H_csc(q.H_jc, q.H_ir, q.H_val)
        for w, v in zip(range(5), (q.g, q.lb, q.ub, q.lbA, q.ubA)):
            s.set_vector(w, v)
        return s

    q = problems.dense_qp()
    s = load(q)
    t = time.perf_counter(); n = s.solve(capi.MODE_COLD, 200000); t = time.perf_counter() - t
    ok, st, _, _ = s.test_optimality()
    out["dense_2048x4096_cold"] = {"seconds": t, "nWSR": n, "ms_per_working_set_change": 1e3 * t / max(n, 1),
                                   "KKT_error": st.KKT_error, "certified": bool(ok)}
    s.close()
    q = problems.sparse_qp()
    s = load(q)
    t = time.perf_counter(); n = s.solve(capi.MODE_COLD, 400000); t = time.perf_counter() - t
    ok, st, _, _ = s.test_optimality()
    out["sparse_10000x20000_cold"] = {"seconds": t, "nWSR": n, "ms_per_working_set_change": 1e3 * t / max(n, 1),
                                      "KKT_error": st.KKT_error, "certified": bool(ok)}
    times, its, kinds, good = [], [], [], True
    for qk, changed in problems.sparse_sequence(q, nsteps=2 * hot_steps):
        t = time.perf_counter()
        for w, v in zip(range(5), (qk.g, qk.lb, qk.ub, qk.lbA, qk.ubA)):
            s.set_vector(w, v)
        if changed:      # QPhandler VARIED: new Jacobian values, hotstart(H, g, A, ...) re-factorises (blocked QR / Cholesky)
            s.set_A_csc(qk.A_jc, qk.A_ir, qk.A_val)
        nk = s.solve(capi.MODE_HOT_MATRICES if changed else capi.MODE_HOT_VECTORS, 400000)
        okk, stk, _, _ = s.test_optimality()          # QPhandler::solveQP = optimizeQP + certificate
        times.append(time.perf_counter() - t); its.append(nk); kinds.append(changed); good = good and bool(okk)
    times, its, kinds = np.array(times), np.array(its), np.array(kinds)
    out["sparse_10000x20000_warm_sequence"] = {
        "qps": len(times), "wall_ms_per_sqp_iteration_mean": 1e3 * float(np.mean(times)),
        "wall_ms_per_sqp_iteration_median": 1e3 * float(np.median(times)), "nWSR_mean": float(np.mean(its)),
        "fixed_matrix_steps": {"qps": int((~kinds).sum()), "wall_ms_mean": 1e3 * float(times[~kinds].mean()), "nWSR_mean": float(its[~kinds].mean())},
        "varied_matrix_steps": {"qps": int(kinds.sum()), "wall_ms_mean": 1e3 * float(times[kinds].mean()), "nWSR_mean": float(its[kinds].mean())},
        "all_certified": good, "note": "alternating FIXED (hotstart on vectors) and VARIED (new Jacobian values: upload, blocked "
                                       "re-factorisation, hotstart) steps, each incl. host transfers and the KKT certificate"}
    s.close()
    q = problems.dense_qp(600, 1200, seed=20260101)
    s = load(q)
    t = time.perf_counter(); n = s.solve(capi.MODE_COLD, 200000); t = time.perf_counter() - t
    qp = O.OracleQP(q.nV, q.nC); qp.set_A_csc(q.A_jc, q.A_ir, q.A_val); qp.set_H_csc(q.H_jc, q.H_ir, q.H_val)
    t0 = time.perf_counter(); rc, n2 = qp.init(q.g, q.lb, q.ub, q.lbA, q.ubA, 200000); t0 = time.perf_counter() - t0
    wb, wc = s.working_set_raw()
    out["dense_600x1200_gpu_vs_cpu_oracle"] = {
        "gpu_seconds": t, "cpu_oracle_seconds": t0, "nWSR_gpu": n, "nWSR_cpu": n2,
        "same_working_set": bool(np.array_equal(wb, qp.ws_bounds) and np.array_equal(wc, qp.ws_constraints)),
        "max_abs_dx": float(np.abs(s.x - qp.x).max())}
    s.close()
    return out


def hs_batch_config(capi, problems, reps=50):
    """BASELINE configs[4]: the batch of 512 independent hs0xx-scale QPs (mixed shapes, so the
    problems sharing a wave diverge), whole on one GPU and as the 64-QP shard one of 8 GPUs gets.
    Cold solve + fused KKT certificate per launch pair; device time by HIP events."""
    out = {}
    for nq in (512, 64):
        probs = problems.hs_batch(nq)
        b = capi.Batch(probs)
        b.solve(capi.MODE_COLD, 1000)
        b.timer_start()
        for _ in range(reps):
            b.solve(capi.MODE_COLD, 1000, sync=False)
        ms = b.timer_stop_ms() / reps
        ok, _ = b.test_optimality()
        res = b.results()
        out["%d_qps" % nq] = {"ms_per_batch": ms, "qp_solves_per_s": nq / (ms * 1e-3), "all_certified": bool(all(o == 1 for o in ok)),
                              "mean_nWSR": float(np.mean([r["nWSR"] for r in res]))}
        b.close()
    return out


def hs071_single_qp_latency(problems, iters=3000):
    """Wall-clock per SQP iteration of hs071 at the boundary, ONE QP at a time: the C++ host
    adapter (restartsqp_amd/csrc/host) replays QPhandler::update_delta + solveQP (hot start +
    mandatory KKT certificate). A single 8-variable QP is launch/sync-latency bound on any GPU;
    the CPU oracle does the same work in a few microseconds -- reported for honesty."""
    import subprocess
    import oracle as O
    host = os.path.join(ROOT, "restartsqp_amd", "csrc", "host")
    subprocess.check_call(["make", "-s", "-C", host, "host_replay"])
    out = subprocess.run([os.path.join(host, "host_replay"), "--bench", str(iters)], capture_output=True, text=True,
                         timeout=300).stdout
    res = {}
    for line in out.splitlines():
        if line.startswith("bench "):
            tok = line.split()
            res["gpu_us_" + ("solveQP" if "certificate" in tok[1] else "optimizeQP_only")] = float(tok[3])
    q1, q2 = problems.handler_qp(problems.hs071_nlp(), delta=1.0), problems.handler_qp(problems.hs071_nlp(), delta=0.5)
    qp = O.OracleQP(q1.nV, q1.nC)
    qp.set_A_csc(q1.A_jc, q1.A_ir, q1.A_val); qp.set_H_csc(q1.H_jc, q1.H_ir, q1.H_val)
    qp.init(q1.g, q1.lb, q1.ub, q1.lbA, q1.ubA, 1000)
    A, H = (q1.A_jc, q1.A_ir, q1.A_val), (q1.H_jc, q1.H_ir, q1.H_val)
    cl = lambda v: np.clip(v, -1e20, 1e20)
    t0 = time.perf_counter()
    for it in range(iters):
        q = q2 if it & 1 else q1
        qp.hotstart(q.g, q.lb, q.ub, q.lbA, q.ubA, 1000)
        x, y = qp.x, qp.y
        Wb, Wc = O.kkt_get_working_set(q.nV, q.nC, A, x, cl(q.lb), cl(q.ub), cl(q.lbA), cl(q.ubA), qp.ws_bounds,
                                       qp.ws_constraints)
        O.kkt_test_optimality(q.nV, q.nC, A, H, q.g, cl(q.lb), cl(q.ub), cl(q.lbA), cl(q.ubA), x, y, Wb, Wc)
    res["cpu_oracle_us_solveQP_incl_python_ctypes"] = 1e6 * (time.perf_counter() - t0) / iters
    return res


def _cpu_worker(args):
    """one process of the all-cores leg: the C oracle on its own copies of the sample QPs"""
    probs, seconds = args
    import oracle as O
    handles = []
    for q in probs:
        qp = O.OracleQP(q.nV, q.nC)
        qp.set_A_csc(q.A_jc, q.A_ir, q.A_val); qp.set_H_csc(q.H_jc, q.H_ir, q.H_val)
        handles.append(qp)
    reps, n, t0 = 50, 0, time.perf_counter()
    while True:
        for qp, q in zip(handles, probs):
            qp.init_repeat(q.g, q.lb, q.ub, q.lbA, q.ubA, 1000, reps)     # C loop: no interpreter inside
        n += reps * len(probs)
        t = time.perf_counter() - t0
        if t >= seconds:
            return n, t


def cpu_baseline(probs, seconds):
    """Oracle (oracle/qp_oracle.c) timed on this host: one thread (the reference's CPU path is
    single-threaded), and -- for fairness -- one QP stream per core on all cores. Test
    infrastructure used as the reported baseline only -- never on the measured GPU path."""
    import oracle as O
    O.build()
    n, t = _cpu_worker((probs, seconds))
    model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    out = {"value": n / t, "unit": "QP solves/s", "cores": 1, "kind": "port", "host_cpu": model, "host_nproc": os.cpu_count(),
           "sample": "%d cold solves of the first %d QPs of the rank-0 batch in %.1f s (in-repo C oracle, gcc -O2, "
                     "1 thread, solve loop in C; qpOASES 3.2.1 is not available)" % (n, len(probs), t)}
    try:
        import multiprocessing as mp
        cores = len(os.sched_getaffinity(0))
        with mp.get_context("fork").Pool(cores) as pool:
            rs = pool.map(_cpu_worker, [(probs, max(1.0, seconds / 3))] * cores)
        out["all_cores"] = {"value": sum(r[0] / r[1] for r in rs), "cores": cores,
                            "note": "one independent QP stream per core (the reference itself has no threading)"}
    except Exception as e:   # the all-cores figure is informational
        out["all_cores"] = {"error": repr(e)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-per-gpu", type=int, default=65536)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--spmv-batch", type=int, default=256)
    ap.add_argument("--no-extras", action="store_true", help="skip cpu_baseline and roofline_spmv")
    ap.add_argument("--no-large", action="store_true", help="skip the dense 2048x4096 / sparse 10k configurations")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    from restartsqp_amd import build, capi, problems
    build.build_lib()
    if capi.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the HIP engine has no CPU fallback")

    B = args.batch_per_gpu
    probs = problems.hs071_scale_batch(B, seed=20260103 + rank)
    batch = capi.Batch(probs, device=local_rank)

    def sync():
        batch and capi.check(capi.lib().rsqp_batch_sync(batch._h))
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        batch.solve(capi.MODE_COLD, 1000, sync=False)
    sync()
    t0 = time.perf_counter()
    batch.timer_start()          # HIP events on the launch stream, around the K launches
    for _ in range(args.steps):
        batch.solve(capi.MODE_COLD, 1000, sync=False)
    kernel_ms_total = batch.timer_stop_ms()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # correctness guard: every QP solved, certificate green (outside the timed region)
    res = batch.results()
    ok, kkt = batch.test_optimality()
    n_bad = sum(1 for r, o in zip(res, ok) if r["status"] != 20 or o != 1)

    if rank == 0:
        total = world * B * args.steps
        k_ms = kernel_ms_total / args.steps   # average launch duration over the timed region
        bytes_launch = float(sum(qp_algorithmic_bytes(q) for q in probs))
        achieved = bytes_launch / (k_ms * 1e-3) / 1e9
        line = {
            "metric": "QP-subproblem solves/sec", "value": total / elapsed, "unit": "QP solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "hs071-scale QP batch (derived hs071 first QP + seeded 1 %% perturbations, "
                                   "nV=8 x nC=2 via QPhandler [J I -I]), cold start, %d QPs/GPU per step" % B,
                       "qps_per_gpu": B, "engine": "small_qp_kernel<Engine<8>> (LDS-resident, 8 lanes per QP = 8 QPs per wave)",
                       "mean_nWSR": float(np.mean([r["nWSR"] for r in res])), "unsolved_or_kkt_fail": n_bad},
            "roofline": {"kernel": "small_qp_kernel<Engine<8>>", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic("Engine<8", 1.0) if B == 65536 else None,
                         "traffic_note": "FETCH_SIZE uncorrected (narrow loads, uncalibrated) + WRITE_SIZE; the "
                                         "writes are the 1.7 KB/QP of engine state a hot start needs",
                         "kernel_ms": k_ms, "algorithmic_bytes_per_launch": bytes_launch,
                         "note": "latency/LDS-bound kernel: HBM fraction is not its limiter"},
        }
        if not args.no_extras and world == 1:   # extras (CPU baseline, secondary rooflines, large configs): N = 1 only
            line["cpu_baseline"] = cpu_baseline(probs[:256], args.cpu_seconds)
            line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
            line["roofline_spmv"] = spmv_roofline(capi, problems, args.spmv_batch, 5)
            line["hs071_single_qp"] = hs071_single_qp_latency(problems)
            line["hs0xx_batch_512"] = hs_batch_config(capi, problems)
            if not args.no_large:
                line["large_engine"] = large_configs(capi, problems)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
