#!/usr/bin/env python3
"""bench.py -- QP-subproblem solves/sec of the MI355X-native engine (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: every rank solves its shard of
independent hs071-scale QPs (the derived hs071 first QP and seeded 1 % perturbations of it,
8 variables x 2 constraints through the QPhandler formulation) from a COLD start with ONE
launch of the LDS-resident active-set kernel, inputs already resident in HBM. Shards are
independent (no data-path collective) -> weak scaling; `value` = QPs of all ranks / max time.

`--gpus N` with N > 1 and no torchrun environment: bench.py starts the N ranks itself (a child
`python -m torch.distributed.run`, before anything touches the GPU) and exits with its code.

The JSON line also carries
  roofline       -- the dominant kernel of the timed region (the batched QP kernel): algorithmic
                    HBM bytes per launch / its HIP-event duration. The kernel is LDS/latency
                    bound, so the fraction is small by nature; see DESIGN.md.
  kernel_ms_stats -- median / quartiles of >= 200 individually timed launches (HIP events).
  with_gather    -- (N > 1) the same step followed by the only collective of the path: the
                    all-gather of fixed-stride result records over RCCL; throughput and latency.
  roofline_spmv  -- the n=10k x m=20k, 200k-nnz Jacobian product A'y (SpHbMat::transposed_times)
                    batched over distinct matrices (> 2x the 256 MiB Infinity Cache), the
                    kernel BASELINE.json's roofline target names; measured outside the timed region.
  cpu_baseline   -- the CPU oracle (a port: qpOASES cannot be built here) built -O3 -march=native on
                    this host, one pinned core, bounded sample of the same workload.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
NO_BUILD = False


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
    the streaming SpMV; WRITE_SIZE is exact. Returns (bytes, file) or (None, None)."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))
    if not paths:
        return None, None
    d = json.load(open(paths[-1]))   # the newest committed PMC pass
    for k, v in d.items():
        if substr in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            b = fetch_factor * v["FETCH_SIZE"]["mean_per_dispatch"] * 1024 + v["WRITE_SIZE"]["mean_per_dispatch"] * 1024
            return b, os.path.basename(paths[-1])
    return None, None


def pmc_fetch_calibration():
    """How FETCH_SIZE has to be read for NARROW coalesced loads (MI355X_MICROARCH.md calibrates the x2 only for 16 B per lane and
    says to calibrate other widths on a known byte count): scatter_values in the same committed PMC file reads exactly
    12 B per entry (a 4-byte index and an 8-byte value per lane, both coalesced, 200 000 entries) -- known bytes / reported bytes
    is the factor. Returns (factor, text) or (None, None)."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))
    if not paths:
        return None, None
    d = json.load(open(paths[-1]))
    for k, v in d.items():
        if k.startswith("scatter_values(") and "FETCH_SIZE" in v:
            rep = v["FETCH_SIZE"]["mean_per_dispatch"] * 1024
            known = 12.0 * 200000
            if rep > 0:
                return known / rep, ("scatter_values (200 000 entries, 4 B + 8 B per lane, coalesced): %.0f B read, FETCH_SIZE reports %.0f B "
                                     "-> factor %.2f (%s)" % (known, rep, known / rep, os.path.basename(paths[-1])))
    return None, None


def pmc_issue_roofline(substr, n_cus=256, n_simd=1024, n_xcd=8):
    """What actually bounds the LDS-resident QP kernel ("achieved LDS / VALU utilisation"), from the
    committed rocprofv3 --pmc passes of `bench.py --no-extras` (tools/pmc_small.sh): LDS-array busy cycles
    (SQ_LDS_IDX_ACTIVE, of which SQ_LDS_BANK_CONFLICT are conflict cycles) per CU-cycle and VALU issue cycles
    (SQ_ACTIVE_INST_VALU, in units of 4 cycles) per SIMD-cycle; kernel cycles = GRBM_GUI_ACTIVE / XCDs."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_small_L8.json")))
    if not paths:
        return None
    try:
        d = json.load(open(paths[-1]))
    except (OSError, ValueError):
        return None
    for k, v in d.items():
        if substr in k and "SQ_LDS_IDX_ACTIVE" in v and "GRBM_GUI_ACTIVE" in v:
            g = lambda c: v[c].get("mean_per_dispatch") if c in v else None
            ratio = lambda a, b: (a / b) if (a is not None and b) else None     # a missing or zero counter gives None, never an exception
            cyc = ratio(g("GRBM_GUI_ACTIVE"), n_xcd)
            lds = ratio(g("SQ_LDS_IDX_ACTIVE"), cyc * n_cus if cyc else None)
            if lds is None:
                return None
            valu = g("SQ_ACTIVE_INST_VALU")
            return {"bound": "lds", "kernel": k[:80], "achieved": lds, "peak": 1.0, "unit": "LDS-array busy cycles per CU cycle",
                    "frac": lds, "lds_bank_conflict_share": ratio(g("SQ_LDS_BANK_CONFLICT"), g("SQ_LDS_IDX_ACTIVE")),
                    "valu_issue_frac": ratio(4.0 * valu if valu is not None else None, cyc * n_simd),
                    "kernel_cycles": cyc, "waves": g("SQ_WAVES"), "insts_valu": g("SQ_INSTS_VALU"), "insts_salu": g("SQ_INSTS_SALU"),
                    "insts_lds": g("SQ_INSTS_LDS"), "wait_inst_any_over_wave_cycles": ratio(g("SQ_WAIT_INST_ANY"), g("SQ_WAVE_CYCLES")),
                    "source": os.path.basename(paths[-1])}
    return None


MAX_LINE_BYTES = 4096      # the driver's record keeps only the tail of stdout: a longer line is cut and parses as nothing (round 4)


def _r(v, sig=6):
    """round a float to `sig` significant digits (None and non-floats pass through)"""
    if isinstance(v, float):
        return float("%.*g" % (sig, v)) if v == v and abs(v) != float("inf") else None
    return v


def _dig(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def compact_line(full):
    """The ONE line bench.py prints: the contract keys, `config` = a workload string + scalars, `roofline` and `cpu_baseline` without
    prose. Everything else `full` holds goes to the sidecar (gpurun_out/bench_extras.json) and stderr. Raises if the line would not
    survive the driver's record (a 20 KB line left BENCH_r04.parsed = null)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data")
    line = {k: _r(full[k], 7) for k in keep}
    c = full.get("config", {})
    cfg = {"workload": c.get("workload"), "qps_per_gpu": c.get("qps_per_gpu"), "mean_nWSR": _r(c.get("mean_nWSR")),
           "unsolved_or_kkt_fail": c.get("unsolved_or_kkt_fail")}
    scal = {
        # BASELINE metric, second half: wall-clock per SQP iteration (reference: src/Algorithm.cpp:57,138-139)
        "hs071_us_per_sqp_iteration_gpu": _dig(full, "hs071_trajectory_latency", "gpu"),
        "hs071_us_per_sqp_iteration_cpu": _dig(full, "hs071_trajectory_latency", "cpu_oracle"),
        "hs071_us_per_solveQP_gpu": _dig(full, "hs071_single_qp", "gpu_us_solveQP"),
        "sparse10k_s_per_sqp_iteration": _dig(c, "sparse10k_s_per_sqp_iteration_reference_rule", "mean"),
        "sparse10k_fixed_s": _dig(c, "sparse10k_s_per_sqp_iteration_reference_rule", "fixed"),
        "sparse10k_varied_s": _dig(c, "sparse10k_s_per_sqp_iteration_reference_rule", "varied"),
        "sparse10k_cold_s": _dig(full, "large_engine", "sparse_10000x20000_cold", "seconds"),
        "sparse10k_band5_cold_s": _dig(full, "large_engine", "sparse_10000x20000_band5", "cold_seconds"),
        "sparse10k_band5_varied_s": _dig(full, "large_engine", "sparse_10000x20000_band5", "varied_seconds"),
        "dense_2048x4096_cold_s": c.get("dense_2048x4096_cold_s"),
        "hs0xx_batch_512_ms": c.get("hs0xx_batch_512_ms"),
        "hs0xx_batch_64_shard_ms": c.get("hs0xx_batch_64_shard_ms"),
        "roofline_mfma_frac": c.get("roofline_mfma_frac"),
        "roofline_spmv_frac": c.get("roofline_spmv_frac"),
        "roofline_spmv_in_solver_frac": _dig(full, "large_engine", "in_solver_spmv", "frac"),
    }
    cfg.update({k: _r(v, 5) for k, v in scal.items() if v is not None})
    line["config"] = cfg
    rf = full.get("roofline", {})
    line["roofline"] = {k: _r(rf.get(k)) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                                                    "kernel_ms", "algorithmic_bytes_per_launch")}
    cb = full.get("cpu_baseline")
    if cb:
        out = {k: _r(cb.get(k)) for k in ("value", "unit", "cores", "kind", "host_cpu", "host_nproc")}
        out["sample"] = (cb.get("sample") or "")[:200]
        ac = _dig(cb, "all_cores", "value")
        if ac is not None:
            out["all_cores_value"] = _r(ac)
        line["cpu_baseline"] = out
        line["speedup_vs_cpu_baseline"] = _r(full.get("speedup_vs_cpu_baseline"), 5)
    wg = full.get("with_gather")
    if wg:
        line["with_gather"] = {k: _r(wg.get(k)) for k in ("value", "unit", "ms_per_step", "all_gather_ms", "ranks_seen")}
    hs = full.get("hs0xx_batch_scaling")
    if hs:
        line["hs0xx_batch_scaling"] = {k: _r(_dig(hs, k, "value")) for k in ("weak_512_per_gpu", "strong_512_total") if k in hs}
    line["extras"] = "gpurun_out/bench_extras.json"
    text = json.dumps(line)
    assert len(text) < MAX_LINE_BYTES and json.loads(text)["metric"], "bench line %d bytes: must stay below %d" % (len(text), MAX_LINE_BYTES)
    return text


def write_extras(full):
    """Sidecar with everything the line no longer carries (per-leg dictionaries, notes, CPU samples)."""
    path = os.path.join(ROOT, "gpurun_out", "bench_extras.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
    except OSError as e:
        print("bench.py: could not write %s: %s" % (path, e), file=sys.stderr)
    return path


def quartiles(ms):
    a = np.sort(np.asarray(ms, dtype=np.float64))
    return {"n": int(len(a)), "median": float(np.median(a)), "q1": float(np.percentile(a, 25)),
            "q3": float(np.percentile(a, 75)), "min": float(a[0]), "max": float(a[-1])}


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
        ms = [plan.run(tr, 1) for _ in range(max(repeats, 20))]
        st = quartiles(ms)
        gbs = bytes_one * nbatch / (st["median"] * 1e-3) / 1e9
        out[name] = {"ms_per_launch": st["median"], "ms_stats": st, "bytes_per_launch": bytes_one * nbatch, "achieved": gbs,
                     "kernel_variant": plan.variant(tr)[0], "idx16": plan.variant(tr)[1]}
    best = out["ATy_csc"]
    # bytes the kernel actually streams: the plan keeps 16-bit copies of the index arrays when both dimensions are
    # < 65536 (10 B instead of 12 B per entry); the entry-parallel kernel reads one start bit per entry instead of the pointers
    streamed = (10 * nnz + (nnz // 8 if best["kernel_variant"] == 40 else 4 * (n + 1)) + 8 * n + 8 * m) * nbatch
    names = {40: ("csx_ldsvec_segscan", "csx_ldsvec_segscan"), 35: ("csx_ldsvec_spmv_pipe2<4,3,ushort>", "csx_ldsvec_spmv_pipe2<4, 3"),
             38: ("csx_ldsvec_spmv_pipe2<2,4,ushort>", "csx_ldsvec_spmv_pipe2<2, 4")}
    kname, ksub = names.get(best["kernel_variant"], ("variant %d" % best["kernel_variant"], "csx_"))
    traffic, tfile = (None, None)
    if nbatch == 256:
        # the entry-parallel kernel has one instance per (no empty major, results staged in LDS): A'y of this pattern
        # is <true, false> (every column has entries; the 160 KB vector leaves no room for the staging window)
        traffic, tfile = pmc_traffic(ksub + "<true, false>", 2.0) if best["kernel_variant"] == 40 else (None, None)
        if traffic is None:
            traffic, tfile = pmc_traffic(ksub, 2.0)
    res = {"consumer": "NONE on a solver path: this batched kernel is reached only through rsqp_spmv_plan_* (this bench leg and the parity "
                       "tests) -- the batched micro-benchmark SURVEY 7 asks for (one 2.7 MB matrix is launch-bound and cache-resident). "
                       "The engines call csx_stream_spmv (single matrix, entry order, bit-exact) and small_certificate_kernel",
           "kernel": kname + " (A'y on CSC = SpHbMat::transposed_times; input vector resident in LDS; "
                     "parity: tests/test_gpu_parity.py::test_roofline_spmv_kernels_match_the_oracle, "
                     "test_entry_parallel_spmv_edge_patterns)",
           "bound": "hbm", "achieved": best["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": best["achieved"] / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tfile,
           "matrices_per_launch": nbatch, "kernel_variant": best["kernel_variant"],
           "bytes_per_matrix": 12 * nnz + 4 * (n + 1) + 8 * n + 8 * m, "ms_per_launch": best["ms_per_launch"],
           "ms_stats": best["ms_stats"],
           "streamed_bytes_per_launch_est": streamed, "raw_stream_GBs_est": streamed / (best["ms_per_launch"] * 1e-3) / 1e9,
           "Ax_csr_GBs": out["Ax_csr"]["achieved"], "Ax_csr_frac": out["Ax_csr"]["achieved"] / HBM_PEAK_GBS,
           "Ax_csr_kernel_variant": out["Ax_csr"]["kernel_variant"]}
    plan.close()
    return res


def value_refresh_roofline(capi, problems):
    """SpHbMat::setMatVal on the device at the QPhandler shape of the sparse configuration
    ([J I -I]: 200 000 Jacobian entries + 40 000 identity entries; SURVEY 8(d): 20 B per entry)."""
    from restartsqp_amd.sqptypes import IdentityInfo
    n, m, nnz = 10000, 20000, 200000
    jc, ir, rng = problems.sparse_pattern(n, m, nnz)
    cols = np.repeat(np.arange(n), np.diff(jc))
    s = capi.Solver(n + 2 * m, m)
    s.set_engine(2)
    ident = IdentityInfo([1, 1], [n + 1, n + m + 1], [m, m], [1.0, -1.0]).blocks()
    s.set_A_triplet(ir + 1, cols + 1, rng.normal(size=nnz), ident)
    s.set_A_triplet(ir + 1, cols + 1, rng.normal(size=nnz), ident)      # value refresh: stages the triplet values
    ms_s, ms_g = s.time_value_refresh(200)
    ms_f = s.time_value_refresh_fused(200)
    t_struct = s.structure_seconds(0)
    s.close()
    bs, bg = 20.0 * nnz, 20.0 * (nnz + 2 * m)
    bf = 32.0 * nnz          # 8 B value + 2 x 4 B index read, 2 x 8 B written
    return {"fused_csc_csr": {"kernel": "scatter_values_csc_csr (what rsqp_set_A_triplet launches on a known pattern: ONE launch)",
                              "entries": nnz, "algorithmic_bytes": bf, "ms_per_launch": ms_f, "achieved": bf / (ms_f * 1e-3) / 1e9,
                              "frac": bf / (ms_f * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "survey_bytes_20_per_entry_frac": bs / (ms_f * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "scatter_values": {"entries": nnz, "algorithmic_bytes": bs, "ms_per_launch": ms_s, "achieved": bs / (ms_s * 1e-3) / 1e9,
                               "frac": bs / (ms_s * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "gather_values": {"entries": nnz + 2 * m, "algorithmic_bytes": bg, "ms_per_launch": ms_g,
                              "achieved": bg / (ms_g * 1e-3) / 1e9, "frac": bg / (ms_g * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "unit": "GB/s", "peak": HBM_PEAK_GBS, "bound": "hbm",
            "set_structure_seconds": t_struct,
            "set_structure_note": "one-off SpHbMat::setStructure equivalent of the first set_A ([J I -I], 240 000 entries, 50 000 "
                                  "columns: host sort by (col, row), CSC + CSR copy + SpMV plan, upload), timed apart from the "
                                  "per-iteration value refresh (SURVEY 8(d))",
            "note": "4.8 MB per launch = 0.6 us at peak: a single refresh is launch-latency bound, not HBM bound; scatter_values / "
                    "gather_values are the two separate kernels of rounds 1-3 (the gather still refreshes the CSR copy on the CSC setter)"}


# the headline batch's kernel by rsqp_batch_get_last_kernel: (name, substring of it in the committed PMC files, description, note)
HEADLINE_KERNELS = {
    1: ("tiny_qp_kernel<2,2>", "tiny_qp_kernel<2",
        "tiny_qp_kernel<2,2> (qp_tiny.hip: register-resident tableau G = -SWEEP_S(K) of the 10 x 10 KKT matrix, 8 lanes per QP = 8 QPs per wave, dense K staged in LDS)",
        "instruction-issue-bound kernel (2 waves per SIMD, ~46k cycles per wave of 8 QPs): the HBM fraction is not its limiter; see roofline_lds / DESIGN.md 6"),
    2: ("lane_qp_kernel<2, false, true>", "lane_qp_kernel<2, false, true>",
        "lane_qp_kernel<2, false, true> (qp_lane.hip: ONE LANE PER QP, 64 QPs per wave, the upper triangle of the tableau G = -SWEEP_S(K) + dense A in LDS [entry][lane], solver state in registers, one wave per SIMD)",
        "instruction-issue-bound kernel (one wave per SIMD, ~105k cycles per wave of 64 QPs, of which ~30k wait for the inputs / drain the results): the HBM fraction is not its limiter; see roofline_lds / DESIGN.md 6"),
}
MFMA_F64_PEAK_TFLOPS = 78.6   # MI355X_MICROARCH.md: dense f64 matrix (v_mfma_f64_16x16x4_f64) peak


def _cpu_changes_leg(args):
    """child process of large_configs: the oracle's first working-set changes of the cold start of a large configuration, in
    chunks (init with a capped nWSR, then hot starts on the same data, which continue the homotopy), one pinned core"""
    which, chunk, seconds = args
    _pin(0)
    import oracle as O
    from restartsqp_amd import problems
    q = problems.sparse_qp() if which == "sparse" else problems.dense_qp()
    qp = O.OracleQP(q.nV, q.nC)
    qp.set_A_csc(q.A_jc, q.A_ir, q.A_val); qp.set_H_csc(q.H_jc, q.H_ir, q.H_val)
    v = (q.g, q.lb, q.ub, q.lbA, q.ubA)
    t = time.perf_counter(); qp.init(*v, 0); t_touch = time.perf_counter() - t     # nWSR = 0: set-up only -- faults the 3 nV^2 factor arrays in
    chunks, done, T = [], 0, 0.0
    while T < seconds:
        t = time.perf_counter()
        rc, n = qp.init(*v, chunk) if not chunks else qp.hotstart(*v, chunk)
        dt = time.perf_counter() - t
        if n == 0:
            break
        done += n; T += dt
        chunks.append({"changes_done": done, "ms_per_change": 1e3 * dt / n, "nFR": int((qp.ws_bounds == 0).sum()),
                       "nAC": int((qp.ws_constraints != 0).sum())})
    return {"value": 1e3 * T / max(done, 1), "unit": "ms per working-set change", "cores": 1, "kind": "port", "changes": done,
            "seconds": T, "first_touch_seconds_excluded": t_touch, "chunks": chunks,
            "sample": "the first %d working-set changes of the cold start (chunks of %d), in-repo C oracle on one pinned core; its cost "
                      "per change grows with the number of free variables (dense Q, T, R of nV^2), so this early sample is the "
                      "CHEAPEST regime of the CPU path" % (done, chunk)}


def large_configs(capi, problems, seq_steps=50, ref_rule_steps=10, cpu_seconds=10.0):
    """BASELINE configs 2 and 3 on the HBM-resident engine (outside the timed region): cold solve of the dense
    2048 x 4096 QP; cold solve + the 50-QP warm-started sequence of the sparse 10 000 x 20 000 QP through
    rsqp_optimize_qp ("wall-clock per SQP iteration, n=10k sparse") under BOTH re-initialisation rules -- the opt-in
    sign(y0) shortcut for all 50 steps and the reference's rule (library default, qpOASESInterface.cpp:199-207) for
    `ref_rule_steps` more steps --, the matrix-core (MFMA) roofline of one hot start with new matrices, a CPU figure per
    working-set change for both configurations, and a mid-size dense QP solved by both the GPU engine and the oracle."""
    import multiprocessing as mp
    import oracle as O
    out = {}

    def load(q):
        s = capi.Solver(q.nV, q.nC)
        s.set_options(qp_maxiter=400000)
        s.set_A_csc(q.A_jc, q.A_ir, q.A_val); s.set_H_csc(q.H_jc, q.H_ir, q.H_val)
        for w, v in zip(range(5), (q.g, q.lb, q.ub, q.lbA, q.ubA)):
            s.set_vector(w, v)
        return s

    def cpu_leg(which, chunk):
        try:
            with mp.get_context("fork").Pool(1) as pool:
                return pool.map(_cpu_changes_leg, [(which, chunk, cpu_seconds)])[0]
        except Exception as e:      # informational leg
            return {"error": repr(e)}

    def same_first_changes(s, k):
        """GPU time for the same first k changes of the cold start (nWSR capped at k; a 1-change solve first creates the engine)"""
        s.solve(capi.MODE_COLD, 1)
        t = time.perf_counter(); n = s.solve(capi.MODE_COLD, k); t = time.perf_counter() - t
        return {"changes": n, "ms_per_change": 1e3 * t / max(n, 1)}

    # ---- config 2: dense 2048 x 4096, cold
    q = problems.dense_qp()
    cpu = cpu_leg("dense", 20)
    s = load(q)
    first = same_first_changes(s, cpu.get("changes", 40))
    t = time.perf_counter(); n = s.solve(capi.MODE_COLD, 200000); t = time.perf_counter() - t
    ok, st, _, _ = s.test_optimality()
    out["dense_2048x4096_cold"] = {"seconds": t, "nWSR": n, "ms_per_working_set_change": 1e3 * t / max(n, 1),
                                   "path": capi.Solver.LARGE_PATHS.get(s.large_path(), "?"),
                                   "KKT_error": st.KKT_error, "certified": bool(ok), "cpu_baseline": cpu,
                                   "gpu_same_first_changes": first,
                                   "gpu_over_cpu_per_change": (cpu["value"] / first["ms_per_change"]) if "value" in cpu else None}
    gold = os.path.join(ROOT, "tests", "golden", "oracle_large_dense_2048x4096.json")
    if os.path.exists(gold):
        g = json.load(open(gold))
        wb, wc = s.working_set_raw()
        out["dense_2048x4096_cold"].update({
            "oracle_nWSR": g["nWSR"], "same_working_set_as_oracle": bool(np.array_equal(wb, g["ws_b"]) and np.array_equal(wc, g["ws_c"])),
            "max_abs_dx_vs_oracle": float(np.abs(s.x - np.array(g["x"])).max()),
            "oracle_seconds_build_container_1_core": g["oracle_seconds_build_container"]})
    s.close()

    # ---- config 3: sparse 10 000 x 20 000
    q = problems.sparse_qp()
    cpu = cpu_leg("sparse", 100)
    s = load(q)
    out["structure_analysis"] = {"set_A_csc_seconds": s.structure_seconds(0), "set_H_csc_seconds": s.structure_seconds(1),
                                 "note": "one-off (SURVEY 8(d)): CSR copy + SpMV plan + upload of an already compressed matrix; the "
                                         "triplet path (SpHbMat::setStructure incl. the sort) is timed in roofline_value_refresh"}
    first = same_first_changes(s, cpu.get("changes", 400))
    t = time.perf_counter(); n = s.optimize_qp(); t = time.perf_counter() - t
    ok, st, _, _ = s.test_optimality()
    out["sparse_10000x20000_cold"] = {"seconds": t, "nWSR": n, "ms_per_working_set_change": 1e3 * t / max(n, 1),
                                      "path": capi.Solver.LARGE_PATHS.get(s.large_path(), "?"),
                                      "KKT_error": st.KKT_error, "certified": bool(ok), "entry": "rsqp_optimize_qp",
                                      "cpu_baseline": cpu, "gpu_same_first_changes": first,
                                      "gpu_over_cpu_per_change": (cpu["value"] / first["ms_per_change"]) if "value" in cpu else None}

    def run_sequence(nsteps, seed, from_y0):
        s.set_reinit_guess(from_y0)
        times, its, kinds, good = [], [], [], True
        for qk, changed in problems.sparse_sequence(q, nsteps=nsteps, seed=seed):
            t = time.perf_counter()
            for w, v in zip(range(5), (qk.g, qk.lb, qk.ub, qk.lbA, qk.ubA)):
                s.set_vector(w, v)
            if changed:      # QPhandler VARIED: new Jacobian values
                s.set_A_csc(qk.A_jc, qk.A_ir, qk.A_val)
            nk = s.optimize_qp()                          # the FIXED / VARIED dispatch of qpOASESInterface.cpp:137-224 decides the mode
            okk, stk, _, _ = s.test_optimality()          # QPhandler::solveQP = optimizeQP + certificate
            times.append(time.perf_counter() - t); its.append(nk); kinds.append(changed); good = good and bool(okk)
        times, its, kinds = np.array(times), np.array(its), np.array(kinds)
        part = lambda m: {"qps": int(m.sum()), "wall_ms_mean": 1e3 * float(times[m].mean()), "nWSR_mean": float(its[m].mean()),
                          "ms_per_working_set_change": 1e3 * float(times[m].sum() / max(its[m].sum(), 1))} if m.any() else None
        return {"qps": len(times), "wall_ms_per_sqp_iteration_mean": 1e3 * float(np.mean(times)),
                "wall_ms_per_sqp_iteration_median": 1e3 * float(np.median(times)), "nWSR_mean": float(np.mean(its)),
                "fixed_matrix_steps": part(~kinds), "varied_matrix_steps": part(kinds), "all_certified": good,
                "entry": "rsqp_optimize_qp", "reinit_guess_from_y0": bool(from_y0)}

    note = ("BASELINE configs[3]: alternating FIXED (new vectors) and VARIED (new Jacobian values) steps through optimizeQP's "
            "dispatch; every VARIED step is a FIXED<->VARIED flip = init(.., x_qp, y_qp, &bounds) (qpOASESInterface.cpp:199-207), "
            "each step incl. host transfers and the KKT certificate")
    seq = run_sequence(seq_steps, 20260102, True)
    seq["note"] = note + "; re-init rule: OPT-IN shortcut rsqp_set_reinit_guess(1) (constraint sides from sign(y_qp)) -- not the reference's path"
    out["sparse_10000x20000_warm_sequence_y0_rule"] = seq
    ref = run_sequence(ref_rule_steps, 20260150, False)
    ref["note"] = (note + "; re-init rule: the REFERENCE's (library default): no guessed constraints, the working set of the constraints is "
                   "rebuilt one change at a time; %d steps continuing the sequence above (the full 50 would take ~1 min)" % ref_rule_steps)
    out["sparse_10000x20000_warm_sequence_reference_rule"] = ref

    # ---- the same configuration on the engine's GENERAL path: the synthetic Hessian of this configuration is diagonal, which the
    #      engine's range-space path exploits; a Hessian with off-diagonal entries -- or the zero curvature the
    #      QPhandler formulation gives its slack variables -- takes the null-space path. Its numbers, same inputs, path forced by
    #      RSQP_LARGE_NO_DUAL (read when the engine of a handle is created): cold start + one FIXED + one VARIED step
    os.environ["RSQP_LARGE_NO_DUAL"] = "1"
    os.environ["RSQP_LARGE_NO_RSH"] = "1"        # (round 5: a diagonal Hessian would otherwise take the GENERAL range-space path, as a band of width 0)
    try:
        s2 = load(q)
        t = time.perf_counter(); n2 = s2.optimize_qp(); t_cold2 = time.perf_counter() - t
        steps2 = []
        for qk, changed in problems.sparse_sequence(q, nsteps=2, seed=20260150):
            t = time.perf_counter()
            for w, v in zip(range(5), (qk.g, qk.lb, qk.ub, qk.lbA, qk.ubA)):
                s2.set_vector(w, v)
            if changed:
                s2.set_A_csc(qk.A_jc, qk.A_ir, qk.A_val)
            nk = s2.optimize_qp()
            okk, _, _, _ = s2.test_optimality()
            steps2.append({"matrices_changed": bool(changed), "seconds": time.perf_counter() - t, "nWSR": nk, "certified": bool(okk)})
        path2 = s2.large_path()
        s2.close()
        out["sparse_10000x20000_null_space_path"] = {
            "cold_seconds": t_cold2, "cold_nWSR": n2, "steps_reference_rule": steps2,
            "path": capi.Solver.LARGE_PATHS.get(path2, "?"),
            "note": "RSQP_LARGE_NO_DUAL=1 RSQP_LARGE_NO_RSH=1: what this configuration costs on the NULL-SPACE path -- since round 5 only an "
                    "indefinite, semidefinite or non-symmetric Hessian takes it (the same working-set sequences: the paths differ in how "
                    "the KKT systems are solved, not in the decisions)"}
    finally:
        del os.environ["RSQP_LARGE_NO_DUAL"]
        del os.environ["RSQP_LARGE_NO_RSH"]

    # ---- SURVEY 8(d)'s other Hessian for this configuration: "+ optional 5-band SPD" (problems.sparse_qp(band=5)): not diagonal,
    #      so DESIGN 4.4's path does not apply; it takes the GENERAL range-space path (bounds and constraints as rows of C,
    #      explicit inverse of C H^-1 C', H^-1 as a banded LDL' operator). Cold start + one FIXED + one VARIED step (reference rule)
    qb = problems.sparse_qp(band=5)
    s3 = load(qb)
    t = time.perf_counter(); n3 = s3.optimize_qp(); t_cold3 = time.perf_counter() - t
    ok3, st3, _, _ = s3.test_optimality()
    b5 = {"cold_seconds": t_cold3, "cold_nWSR": n3, "cold_ms_per_working_set_change": 1e3 * t_cold3 / max(n3, 1), "cold_certified": bool(ok3),
          "path": capi.Solver.LARGE_PATHS.get(s3.large_path(), "?"), "steps_reference_rule": []}
    for qk, changed in problems.sparse_sequence(qb, nsteps=2, seed=20260150):
        t = time.perf_counter()
        for w, v in zip(range(5), (qk.g, qk.lb, qk.ub, qk.lbA, qk.ubA)):
            s3.set_vector(w, v)
        if changed:
            s3.set_A_csc(qk.A_jc, qk.A_ir, qk.A_val)
        nk = s3.optimize_qp()
        okk, _, _, _ = s3.test_optimality()
        dt = time.perf_counter() - t
        b5["steps_reference_rule"].append({"matrices_changed": bool(changed), "seconds": dt, "nWSR": nk, "certified": bool(okk)})
        b5["varied_seconds" if changed else "fixed_seconds"] = dt
    s3.close()
    b5["note"] = ("H = diag + two off-diagonals on each side (strictly diagonally dominant); same Jacobian, gradient and limits as the diagonal "
                  "configuration; round 4 (null-space path, any non-diagonal H): cold 13.1 s, VARIED 5.08 s")
    out["sparse_10000x20000_band5"] = b5

    # ---- the matrix-core path: one more VARIED step right after a VARIED one = hotstart(H, g, A, ..) on the full working set:
    #      blocked Householder QR of A_AC,FR', explicit Q = [Y Z], R^-1, Z'HZ, its Cholesky factor and inverse (dense_la.hip)
    for qk, changed in problems.sparse_sequence(q, nsteps=2, seed=20260160):
        if not changed:
            continue
        for w, v in zip(range(5), (qk.g, qk.lb, qk.ub, qk.lbA, qk.ubA)):
            s.set_vector(w, v)
        s.set_A_csc(qk.A_jc, qk.A_ir, qk.A_val)
        t = time.perf_counter(); nk = s.optimize_qp(); t = time.perf_counter() - t
        okk, stk, _, _ = s.test_optimality()
        sp = s.setup_profile()
        if sp and sp["range_space"]:
            # diagonal Hessian: the engine's range-space path builds (A_AC,FR D^-1 A_AC,FR')^-1 -- Gram matrix by GEMM, blocked Cholesky,
            # triangular inverse, U^-1 U^-T -- instead of QR + Q + R^-1 + Z'HZ (that set-up is measured below with the path switched off)
            fl, ms = sp["flops_qr_q_rinv"] + sp["flops_zhz_chol_inv"], sp["ms_qr_q_rinv"] + sp["ms_zhz_chol_inv"]
            tf = fl / (ms * 1e-3) / 1e12
            out["roofline_mfma"] = {
                "kernel": "blocked set-up of hotstart(H, g, A, ..) on the range-space path (diagonal H): Gram matrix B'B by k_dgemm "
                          "(v_mfma_f64_16x16x4_f64), blocked Cholesky, triangular inverse, U^-1 U^-T, dense_la.hip",
                "bound": "mfma", "achieved": tf, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F64_PEAK_TFLOPS,
                "traffic": None, "algorithmic_flops": fl, "ms": ms, "sizes": {k: sp[k] for k in ("nFR", "nAC")},
                "parts": {"gram": {"ms": sp["ms_qr_q_rinv"], "flops": sp["flops_qr_q_rinv"],
                                   "tflops": sp["flops_qr_q_rinv"] / max(sp["ms_qr_q_rinv"] * 1e-3, 1e-9) / 1e12},
                          "chol_inv": {"ms": sp["ms_zhz_chol_inv"], "flops": sp["flops_zhz_chol_inv"],
                                       "tflops": sp["flops_zhz_chol_inv"] / max(sp["ms_zhz_chol_inv"] * 1e-3, 1e-9) / 1e12}},
                "step": {"mode": capi.Solver.MODE_NAMES.get(s.last_mode(), "?"), "mode_id": s.last_mode(), "wall_ms": 1e3 * t, "nWSR": nk, "certified": bool(okk)},
                "note": "algorithmic flops, symmetric results counted once: Gram matrix n^2 m (m = nFR, n = nAC; the GEMM as launched "
                        "computes both triangles: 2 n^2 m), Cholesky + triangular inverse + U^-1 U^-T n^3; time = HIP events on the engine's stream"}
        elif sp:
            fl, ms = sp["flops_qr_q_rinv"] + sp["flops_zhz_chol_inv"], sp["ms_qr_q_rinv"] + sp["ms_zhz_chol_inv"]
            tf = fl / (ms * 1e-3) / 1e12
            out["roofline_mfma"] = {
                "kernel": "blocked set-up of hotstart(H, g, A, ..): k_dgemm (v_mfma_f64_16x16x4_f64) + panel kernels, dense_la.hip",
                "bound": "mfma", "achieved": tf, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F64_PEAK_TFLOPS,
                "traffic": None, "algorithmic_flops": fl, "ms": ms, "sizes": {k: sp[k] for k in ("nFR", "nAC", "nZ")},
                "parts": {"qr_q_rinv": {"ms": sp["ms_qr_q_rinv"], "flops": sp["flops_qr_q_rinv"],
                                        "tflops": sp["flops_qr_q_rinv"] / (sp["ms_qr_q_rinv"] * 1e-3) / 1e12},
                          "zhz_chol_inv": {"ms": sp["ms_zhz_chol_inv"], "flops": sp["flops_zhz_chol_inv"],
                                           "tflops": sp["flops_zhz_chol_inv"] / max(sp["ms_zhz_chol_inv"] * 1e-3, 1e-9) / 1e12}},
                "step": {"mode": capi.Solver.MODE_NAMES.get(s.last_mode(), "?"), "mode_id": s.last_mode(), "wall_ms": 1e3 * t, "nWSR": nk, "certified": bool(okk)},
                "note": "algorithmic flops: QR 2n^2(m-n/3) + explicit Q 4(m^2 n - m n^2 + n^3/3) + R^-1 n^3/3 (m = nFR, n = nAC); "
                        "Z'HZ nV nZ^2 + Cholesky, inverse and U^-1 U^-T nZ^3; time = HIP events on the engine's stream; "
                        "MFMA-busy counters: profiles/r03_*_pmc_mfma_blocked_setup.json"}
    if "roofline_mfma" in out and "gram" in out["roofline_mfma"].get("parts", {}):
        # the null-space path's set-up (general H) at the same size, stand-alone through rsqp_dense_qr: blocked Householder QR with
        # Cholesky-QR panels + explicit Q + R^-1 of nFR x nAC (host buffers in and out; the time is HIP events around the device work)
        try:
            import ctypes as C
            m_, n_ = out["roofline_mfma"]["sizes"]["nFR"], out["roofline_mfma"]["sizes"]["nAC"]
            rng = np.random.default_rng(0)
            B = np.asfortranarray(rng.normal(size=(m_, n_))); Q = np.zeros((m_, m_), order="F"); Ri = np.zeros((n_, n_), order="F")
            nd, msq = C.c_int(0), C.c_float(0)
            dpp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
            if capi.lib().rsqp_dense_qr(m_, n_, dpp(B), dpp(Q), dpp(Ri), 1e-9, C.byref(nd), C.byref(msq)) == 0:
                m, n = float(m_), float(n_)
                flq = 2.0 * n * n * (m - n / 3.0) + 4.0 * (m * m * n - m * n * n + n * n * n / 3.0) + n * n * n / 3.0
                out["roofline_mfma_blocked_qr"] = {
                    "kernel": "blocked Householder QR (Cholesky-QR panels) + explicit Q + R^-1, the set-up of the null-space path (general H), dense_la.hip",
                    "bound": "mfma", "achieved": flq / (msq.value * 1e-3) / 1e12, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": flq / (msq.value * 1e-3) / 1e12 / MFMA_F64_PEAK_TFLOPS, "algorithmic_flops": flq, "ms": msq.value,
                    "sizes": {"m": m_, "n": n_}, "note": "stand-alone (rsqp_dense_qr on a random matrix of the size above); not on the default path of this configuration any more"}
            del B, Q, Ri
        except (AttributeError, MemoryError):
            pass
    # per-kernel rooflines of the HBM-resident engine: two more steps of the sequence (outside every timing above) with the
    # engine's own accounting on -- HIP events around every launch of a kernel class, algorithmic bytes per call
    s.set_reinit_guess(True)
    s.set_engine_profiling(True)
    for qk, changed in problems.sparse_sequence(q, nsteps=2, seed=20260199):
        for w, v in zip(range(5), (qk.g, qk.lb, qk.ub, qk.lbA, qk.ubA)):
            s.set_vector(w, v)
        if changed:
            s.set_A_csc(qk.A_jc, qk.A_ir, qk.A_val)
        s.optimize_qp()
    prof = s.engine_profile()
    if prof:
        if "spmv" in prof:      # the product the SOLVER runs (csx_stream_spmv, one matrix, entry order): its own figure, next to the
            out["in_solver_spmv"] = dict(prof["spmv"], kernel="csx_stream_spmv", note=(     # batched roofline kernel's
                "every A x / A'y of the HBM engine's homotopy on the sparse 10k x 20k configuration: 2.7 MB per call, launch-bound and "
                "cache-resident -- NOT the 0.69 of roofline_spmv, which is the batched kernel no solver path calls"))
        out["roofline_large_engine_kernels"] = {
            "workload": "sparse 10 000 x 20 000, one FIXED and one VARIED step of the warm-started sequence", "kernels": prof,
            "note": "achieved = algorithmic bytes (8 B x rows x cols for a product, 16 B for a rank-1 update) / HIP-event time per call"}
    s.close()
    q = problems.dense_qp(600, 1200, seed=20260101)
    s = load(q)
    t = time.perf_counter(); n = s.solve(capi.MODE_COLD, 200000); t = time.perf_counter() - t
    qp = O.OracleQP(q.nV, q.nC); qp.set_A_csc(q.A_jc, q.A_ir, q.A_val); qp.set_H_csc(q.H_jc, q.H_ir, q.H_val)
    t0 = time.perf_counter(); rc, n2 = qp.init(q.g, q.lb, q.ub, q.lbA, q.ubA, 200000); t0 = time.perf_counter() - t0
    wb, wc = s.working_set_raw()
    out["dense_600x1200_gpu_vs_cpu_oracle"] = {
        "gpu_seconds": t, "cpu_oracle_seconds": t0, "nWSR_gpu": n, "nWSR_cpu": n2,
        "gpu_ms_per_change": 1e3 * t / max(n, 1), "cpu_ms_per_change": 1e3 * t0 / max(n2, 1),
        "same_working_set": bool(np.array_equal(wb, qp.ws_bounds) and np.array_equal(wc, qp.ws_constraints)),
        "max_abs_dx": float(np.abs(s.x - qp.x).max())}
    s.close()
    return out


def hs_batch_config(capi, problems, parallel, reps=50, cpu_seconds=4.0):
    """BASELINE configs[4]: the batch of 512 independent hs0xx-scale QPs (mixed shapes, so the
    problems sharing a wave diverge), whole on one GPU and as the 64-QP shard one of 8 GPUs gets
    (parallel.balanced_shards: largest first, dealt round-robin). Cold solve per launch; device time by
    HIP events. CPU figure: the oracle on the same 512 QPs, one core and all cores."""
    out = {}
    all_probs = problems.hs_batch(512)
    order = parallel.balanced_order(all_probs)
    shard0 = parallel.balanced_shards(all_probs, 8)[0]
    for tag, idx in (("512_qps", order), ("64_qps_shard_of_8_gpus", shard0)):
        probs = [all_probs[k] for k in idx]
        b = capi.Batch(probs)
        b.set_keep_state(False)      # cold-start-only batch: the KKT-tableau kernel + the null-space kernel on what it bails on
        b.solve(capi.MODE_COLD, 1000)
        ms = []
        for _ in range(reps):
            b.solve(capi.MODE_COLD, 1000, sync=True)
            ms.append(b.last_solve_ms())
        st = quartiles(ms)
        ok, _ = b.test_optimality()
        res = b.results()
        out[tag] = {"ms_per_batch": st["median"], "ms_stats": st, "qp_solves_per_s": len(probs) / (st["median"] * 1e-3),
                    "all_certified": bool(all(o == 1 for o in ok)), "mean_nWSR": float(np.mean([r["nWSR"] for r in res])),
                    "order": "largest first (parallel.balanced_order)", "keep_state": False,
                    "engine": "small_qpg_kernel<3,1,9,4> (qp_small_g.h: tableau of the KKT matrix in registers, one pivot per working-set change) + small_qp_kernel<EngineX<256>> on the members it bails on"}
        b.close()
    cpu = cpu_batch_baseline(all_probs, cpu_seconds)
    out["cpu_baseline"] = cpu
    out["speedup_vs_cpu_1_core"] = out["512_qps"]["qp_solves_per_s"] / cpu["value"]
    if "value" in cpu.get("all_cores", {}):
        out["speedup_vs_cpu_all_cores"] = out["512_qps"]["qp_solves_per_s"] / cpu["all_cores"]["value"]
    return out


def hs_batch_scaling(capi, problems, parallel, torch, dist, rank, world, local_rank, cdev, reps=30):
    """BASELINE configs[4] as a scaling pair, on every rank and for every N (the driver's N = 1, 2, 4, 8 runs give both
    curves from one command): WEAK -- 512 mixed hs0xx QPs per GPU (each rank its own seeded batch) -- and STRONG -- the same
    512 QPs dealt over the ranks (parallel.balanced_shards: 64 per GPU at N = 8). `reps` cold launches each, bracketed by a
    barrier + device sync on both sides, max over ranks; no collective on the data path."""
    def run(probs):
        b = capi.Batch(probs, device=local_rank)
        b.set_keep_state(False)
        b.solve(capi.MODE_COLD, 1000)

        def sync():
            capi.check(capi.lib().rsqp_batch_sync(b._h))
            if dist is not None:
                dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            b.solve(capi.MODE_COLD, 1000, sync=False)
        sync()
        el = time.perf_counter() - t0
        ok, _ = b.test_optimality()
        bad = sum(1 for o in ok if o != 1)
        b.close()
        if dist is not None:
            t = torch.tensor([el, float(bad)], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el, bad = float(t[0].item()), int(t[1].item())
        return el, bad
    mine = problems.hs_batch(512, seed=20260103 + 1000 * rank)
    el_w, bad_w = run([mine[k] for k in parallel.balanced_order(mine)])
    shared = problems.hs_batch(512)
    shard = parallel.balanced_shards(shared, world)[rank]
    el_s, bad_s = run([shared[k] for k in shard])
    return {"weak_512_per_gpu": {"value": world * 512 * reps / el_w, "unit": "QP solves/s", "ms_per_step": 1e3 * el_w / reps,
                                 "qps_per_gpu": 512, "scaling": "weak", "kkt_failures_max_over_ranks": bad_w},
            "strong_512_total": {"value": 512 * reps / el_s, "unit": "QP solves/s", "ms_per_step": 1e3 * el_s / reps,
                                 "qps_per_gpu": len(shard), "scaling": "strong", "kkt_failures_max_over_ranks": bad_s},
            "n_gpus": world, "steps": reps}


def native_rccl_gather(capi, torch, dist, batch, B, rank, world, local_rank, ksteps):
    """the gather of the result records through the C ABI's own RCCL call site (rsqp_batch_allgather_records: pack on the
    device, in-place ncclAllGather on the batch's stream) -- what a C++ host without Python calls. The communicator is
    created through the C ABI as well; only the 128-byte unique id travels over the existing process group (world > 1)."""
    uid = capi.rccl_unique_id() if rank == 0 else bytes(128)
    if dist is not None:
        t = torch.tensor(list(uid), dtype=torch.uint8, device="cuda")
        dist.broadcast(t, 0)
        uid = bytes(t.cpu().tolist())
    comm = capi.RcclComm(uid, rank, world, local_rank)
    stride = batch.record_stride
    allrec = torch.zeros(world * B * stride, dtype=torch.float64, device="cuda")
    batch.solve(capi.MODE_COLD, 1000, sync=False)
    batch.allgather_records(comm, B, allrec.data_ptr())          # warm-up (RCCL builds its channels on the first call)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0, only = time.perf_counter(), 0.0
    for _ in range(ksteps):
        batch.solve(capi.MODE_COLD, 1000, sync=False)
        capi.check(capi.lib().rsqp_batch_sync(batch._h))
        tc = time.perf_counter()
        batch.allgather_records(comm, B, allrec.data_ptr())
        only += time.perf_counter() - tc
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    tg = time.perf_counter() - t0
    got = allrec.view(world * B, stride)
    seen = int((got[::B, 0] == 20).sum().item())
    comm.close()
    if dist is not None:
        tt = torch.tensor([tg, only], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        tg, only = (float(v) for v in tt.tolist())
    if seen != world:
        raise SystemExit("bench.py: native RCCL all-gather returned the records of %d ranks, %d expected" % (seen, world))
    return {"entry": "rsqp_batch_allgather_records (C ABI: pack + in-place ncclAllGather on the batch's stream; communicator from "
                     "rsqp_rccl_comm_create)", "steps": ksteps, "ms_per_step": 1e3 * tg / ksteps, "pack_plus_all_gather_ms": 1e3 * only / ksteps,
            "value": world * B * ksteps / tg, "unit": "QP solves/s", "ranks_seen": seen, "world": world,
            "bytes_gathered_per_rank": 8 * world * B * stride}


def trajectory_batch(capi, problems, nq=65536, reps=20):
    """The hs071-scale batch path on a REPRESENTATIVE mix, at the headline's batch size (rounds 3-4 ran 16 384 members of it on the
    8-lanes-per-problem kernel: 0.0416 ms = 394 M solves/s; 65 536 members take that kernel 0.138 ms): the QPs of a whole
    hs071 SQP run (tests/golden/sqp_traces.json: the
    iterates, multipliers, radii and penalties of the trajectory that tests/sqp_driver.py walks to the optimum), each with
    seeded 1 % perturbations, cold start -- not only the first QP of the run, whose two working-set changes make it the
    easiest one."""
    path = os.path.join(ROOT, "tests", "golden", "sqp_traces.json")
    if not os.path.exists(path):
        return None
    tr = json.load(open(path))["hs071"]["qps"]
    base = [problems.handler_qp(problems.hs071_nlp(np.array(g["x"]), np.array(g["lam"])), delta=g["delta"], rho=g["rho"]) for g in tr]
    rng = np.random.default_rng(20260104)
    probs = [problems.perturb(rng, base[k % len(base)]) for k in range(nq)]
    b = capi.Batch(probs)
    b.set_keep_state(False)
    b.solve(capi.MODE_COLD, 1000)
    ms = []
    for _ in range(reps):
        b.solve(capi.MODE_COLD, 1000, sync=True)
        ms.append(b.last_solve_ms())
    res = b.results()
    ok, _ = b.test_optimality()
    st = quartiles(ms)
    out = {"qps": nq, "distinct_trajectory_qps": len(base), "ms_per_batch": st["median"], "qp_solves_per_s": nq / (st["median"] * 1e-3),
           "mean_nWSR": float(np.mean([r["nWSR"] for r in res])), "max_nWSR": int(max(r["nWSR"] for r in res)),
           "solved_and_certified": int(sum(1 for r, o in zip(res, ok) if r["status"] == 20 and o == 1)),
           "kernel": {0: "LDS null-space kernels", 1: "tiny_qp_kernel (8 lanes per QP)", 2: "lane_qp_kernel (one lane per QP)"}.get(b.last_kernel()),
           "note": "cold starts of the %d QPs of the hs071 trajectory (x_k, lambda_k, delta_k of every SQP iteration) +- 1 %%" % len(base)}
    b.close()
    return out


def hs071_single_qp_latency(problems, iters=3000):
    """Wall-clock per SQP iteration of hs071 at the boundary, ONE QP at a time: the C++ host
    adapter (restartsqp_amd/csrc/host) replays QPhandler::update_delta + solveQP (hot start +
    mandatory KKT certificate). A single 8-variable QP is launch/sync-latency bound on any GPU;
    the CPU oracle does the same work in a few microseconds -- reported for honesty."""
    import oracle as O
    host = os.path.join(ROOT, "restartsqp_amd", "csrc", "host")
    if not NO_BUILD:
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
    cl = lambda v: np.clip(v, -1e20, 1e20)
    va, vb = [cl(v) for v in (q1.g, q1.lb, q1.ub, q1.lbA, q1.ubA)], [cl(v) for v in (q2.g, q2.lb, q2.ub, q2.lbA, q2.ubA)]
    qp.solveqp_repeat(va, vb, 1000, 1000)                       # warm-up
    n_cpu = 20 * iters
    t0 = time.perf_counter()
    good = qp.solveqp_repeat(va, vb, 1000, n_cpu)               # hot start + get_working_set + test_optimality, loop in C
    res["cpu_oracle_us_solveQP"] = 1e6 * (time.perf_counter() - t0) / n_cpu
    res["cpu_oracle_all_certified"] = bool(good == n_cpu)
    if "gpu_us_solveQP" in res:
        res["gpu_over_cpu_latency"] = res["gpu_us_solveQP"] / res["cpu_oracle_us_solveQP"]
    res["note"] = ("one 8-variable QP at a time is launch / host-sync latency bound on the GPU: the CPU oracle (same loop in C, one "
                   "core) is FASTER per solve; the GPU wins only on batches (the headline line)")
    return res


def hs071_trajectory_latency(reps=300):
    """"Wall-clock per SQP iteration (hs071)" as the reference clocks it (src/Algorithm.cpp:57,138-139), QP side: the WHOLE 6-QP
    trajectory of tests/golden/sqp_traces.json through the C++ boundary (host_replay --trajectory: setupQP with set_A / set_H
    refreshes and the per-element update_* storm, solveQP = optimizeQP + certificate, getters) and the same loop in C over the
    CPU oracle (oracle/traj_oracle.c) on one core."""
    import tempfile
    import oracle as O
    path = os.path.join(ROOT, "tests", "golden", "sqp_traces.json")
    if not os.path.exists(path):
        return None
    tr = json.load(open(path))["hs071"]["qps"]
    traj = [[q["delta"], q["rho"]] + list(q["x"]) + list(q["lam"]) for q in tr]
    host = os.path.join(ROOT, "restartsqp_amd", "csrc", "host")
    res = {"sqp_iterations": len(traj), "reps": reps}
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        for row in traj:
            f.write(" ".join(repr(float(v)) for v in row) + "\n")
        tpath = f.name
    try:
        out = subprocess.run([os.path.join(host, "host_replay"), "--trajectory", tpath, str(reps)], capture_output=True, text=True,
                             timeout=300).stdout
    finally:
        os.unlink(tpath)
    gx = None
    for line in out.splitlines():
        tok = line.split()
        if line.startswith("trajectory sqp_iterations"):
            kv = dict(zip(tok[1::2], tok[2::2]))
            res["gpu"] = float(kv["us_per_sqp_iteration"]); res["gpu_first_iteration"] = float(kv["us_first_iteration"])
            res["gpu_later_iterations"] = float(kv["us_later_iterations"]); res["gpu_qp_iter"] = int(kv["qp_iter"])
        elif line.startswith("trajectory_last_x"):
            gx = np.array([float(v) for v in tok[1:]])
    cpu = O.hs071_trajectory_replay(traj, 50 * reps)
    res["cpu_oracle"] = cpu["us_per_sqp_iteration"]; res["cpu_oracle_first_iteration"] = cpu["us_first_iteration"]
    res["cpu_oracle_later_iterations"] = cpu["us_later_iterations"]; res["cpu_oracle_qp_iter"] = cpu["qp_iter"]
    res["cpu_oracle_all_certified"] = cpu["failed_iteration"] == 0
    gold = np.array(tr[-1]["x_qp"])
    if gx is not None:
        res["gpu_last_qp_max_abs_dx_vs_trace"] = float(np.abs(gx - gold).max())
    res["cpu_last_qp_max_abs_dx_vs_trace"] = float(np.abs(cpu["x"] - gold).max())
    if "gpu" in res:
        res["gpu_over_cpu_latency"] = res["gpu"] / res["cpu_oracle"]
    res["unit"] = "microseconds per SQP iteration (QP side: setupQP + solveQP + getters; NLP evaluation closed form)"
    res["note"] = ("one 8-variable QP at a time is launch / host-sync latency bound on a GPU; ubA is refreshed in update_bounds "
                   "(the reference's stale ubA, QPhandler.cpp:358-360, makes its own hs071 run infeasible after the first step)")
    return res


def _pin(core):
    try:
        allowed = sorted(os.sched_getaffinity(0))
        os.sched_setaffinity(0, {allowed[core % len(allowed)]})
    except (AttributeError, OSError):
        pass


def _cpu_worker(args):
    """one process of a CPU leg: the C oracle on its own copies of the sample QPs, pinned to one core"""
    probs, seconds, core = args
    _pin(core)
    import oracle as O
    handles = []
    for q in probs:
        qp = O.OracleQP(q.nV, q.nC)
        qp.set_A_csc(q.A_jc, q.A_ir, q.A_val); qp.set_H_csc(q.H_jc, q.H_ir, q.H_val)
        handles.append(qp)
    small = max(q.nV for q in probs) <= 16
    reps, n, t0 = (50 if small else 1), 0, time.perf_counter()
    while True:
        for qp, q in zip(handles, probs):
            qp.init_repeat(q.g, q.lb, q.ub, q.lbA, q.ubA, 1000, reps)     # C loop: no interpreter inside
        n += reps * len(probs)
        t = time.perf_counter() - t0
        if t >= seconds:
            return n, t


def _host_info():
    model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model


def _all_cores(probs, seconds):
    try:
        import multiprocessing as mp
        cores = len(os.sched_getaffinity(0))
        with mp.get_context("fork").Pool(cores) as pool:
            rs = pool.map(_cpu_worker, [(probs, seconds, c) for c in range(cores)])
        return {"value": sum(r[0] / r[1] for r in rs), "cores": cores,
                "note": "one independent QP stream per core, each pinned (the reference itself has no threading)"}
    except Exception as e:   # the all-cores figure is informational
        return {"error": repr(e)}


def cpu_baseline(probs, seconds):
    """Oracle (oracle/qp_oracle.c) timed on this host: one pinned thread (the reference's CPU path is
    single-threaded), and -- for fairness -- one QP stream per core on all cores. Test
    infrastructure used as the reported baseline only -- never on the measured GPU path."""
    runs = [_run_worker_in_child(probs, seconds / 3.0) for _ in range(3)]
    rates = [n / t for n, t in runs]
    n = sum(r[0] for r in runs); t = sum(r[1] for r in runs)
    out = {"value": float(np.median(rates)), "unit": "QP solves/s", "cores": 1, "kind": "port", "host_cpu": _host_info(),
           "host_nproc": os.cpu_count(), "runs": rates,
           "sample": "%d cold solves of the first %d QPs of the rank-0 batch in %.1f s (3 runs, median; in-repo C oracle, "
                     "gcc -O3 -march=native built on this host, 1 thread pinned to a core, solve loop in C; qpOASES 3.2.1 "
                     "is not available)" % (n, len(probs), t)}
    out["all_cores"] = _all_cores(probs, max(1.0, seconds / 3))
    return out


def cpu_batch_baseline(probs, seconds):
    """the same for the 512-QP mixed batch of BASELINE configs[4] (whole batches, cold)"""
    n, t = _run_worker_in_child(probs, seconds)
    out = {"value": n / t, "unit": "QP solves/s", "cores": 1, "kind": "port",
           "sample": "%d cold solves (%.1f passes over the 512 mixed hs0xx QPs) in %.1f s, 1 pinned core" % (n, n / len(probs), t)}
    out["all_cores"] = _all_cores(probs, max(1.0, seconds / 2))
    return out


def _run_worker_in_child(probs, seconds):
    """time the one-core leg in a forked child (pinning stays local to it)"""
    import multiprocessing as mp
    with mp.get_context("fork").Pool(1) as pool:
        return pool.map(_cpu_worker, [(probs, seconds, 0)])[0]


def spawn_ranks(args):
    """--gpus N without a torchrun environment: start the N rank processes as a child job"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    global NO_BUILD
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch-per-gpu", type=int, default=65536)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--spmv-batch", type=int, default=256)
    ap.add_argument("--stat-launches", type=int, default=200, help="individually timed launches for median / IQR")
    ap.add_argument("--no-extras", action="store_true", help="skip cpu_baseline and the secondary measurements")
    ap.add_argument("--no-large", action="store_true", help="skip the dense 2048x4096 / sparse 10k configurations")
    ap.add_argument("--no-build", action="store_true",
                    help="never run hipcc / make (profiled runs: build first, outside rocprofv3 -- tools/*.sh)")
    ap.add_argument("--keep-state", type=int, default=0,
                    help="1: every cold solve also writes the hot-start image (1.7 KB/QP) back to HBM")
    args = ap.parse_args()
    NO_BUILD = args.no_build

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        import torch
        have = torch.cuda.device_count()          # does not initialise the GPU
        if have < args.gpus and os.environ.get("RSQP_BENCH_REHEARSAL") != "1":
            raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible" % (args.gpus, have))
        raise SystemExit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ONE JSON line on stdout, whatever the libraries print: RCCL writes a five-line version banner to file descriptor 1 when its
    # first communicator is created (seen in profiles/r04_a: it landed in front of the JSON line). Everything that is not the result
    # goes to stderr -- descriptor 1 is pointed there for the run, the line is written to the saved descriptor at the end
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    # rehearsal of the N > 1 path on a box with ONE GPU (RSQP_BENCH_REHEARSAL=1): all ranks share device 0 and the
    # collectives run over gloo on host tensors -- checks the logic (spawn, barriers, record packing, gather), not RCCL
    rehearsal = os.environ.get("RSQP_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch
    # fail loudly BEFORE any GPU call when the job cannot be the N-GPU job that was asked for (device_count() does not
    # initialise the GPU): every rank needs a device of its own
    if not rehearsal:
        have = torch.cuda.device_count()
        if have < world or local_rank >= have:
            raise SystemExit("bench.py: --gpus %d but rank %d (local rank %d) sees %d GPU(s): refusing to run a smaller job "
                             "under the name of a larger one" % (args.gpus, rank, local_rank, have))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        import datetime
        tmo = datetime.timedelta(seconds=180)
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, timeout=tmo, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    cdev = "cpu" if rehearsal else "cuda"          # where the tensors of the collectives live
    from restartsqp_amd import build, capi, parallel, problems
    if not NO_BUILD:
        if rank == 0:
            build.build_lib()      # (a no-op when the library is up to date; never N ranks compiling into one file)
        if dist is not None:
            dist.barrier()
    if capi.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the HIP engine has no CPU fallback")

    B = args.batch_per_gpu
    probs = problems.hs071_scale_batch(B, seed=20260103 + rank)
    batch = capi.Batch(probs, device=local_rank)
    batch.set_keep_state(bool(args.keep_state))

    def sync():
        capi.check(capi.lib().rsqp_batch_sync(batch._h))
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
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # launch-by-launch statistics (each launch bracketed by its own HIP events; outside the timed region)
    per_launch = []
    for _ in range(args.stat_launches):
        batch.solve(capi.MODE_COLD, 1000, sync=True)
        per_launch.append(batch.last_solve_ms())

    # the one collective of the path: all-gather of the fixed-stride result records (RCCL), N > 1 only
    gather = None
    if dist is not None:
        # no try / except here: a rank that failed inside this block would fall through to the next collective while its
        # peers are still in all_gather -- mismatched collectives hang until the timeout; an exception ends the job instead
        stride = batch.record_stride
        rec = torch.zeros(B * stride, dtype=torch.float64, device="cuda")
        allrec = torch.zeros(world * B * stride, dtype=torch.float64, device=cdev)
        ksteps = max(5, min(args.steps, 50))
        for timed in (False, True):
            sync()
            tg0, only = time.perf_counter(), 0.0
            for _ in range(ksteps):
                batch.solve(capi.MODE_COLD, 1000, sync=False)
                batch.pack_records_dev(rec.data_ptr())
                capi.check(capi.lib().rsqp_batch_sync(batch._h))
                tc = time.perf_counter()
                if rehearsal:
                    parts = [torch.zeros(B * stride, dtype=torch.float64) for _ in range(world)]
                    dist.all_gather(parts, rec.cpu())
                    allrec = torch.cat(parts)
                else:
                    dist.all_gather_into_tensor(allrec, rec)
                torch.cuda.synchronize()
                only += time.perf_counter() - tc
            sync()
            tg = time.perf_counter() - tg0
        tt = torch.tensor([tg, only], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        tg, only = (float(v) for v in tt.tolist())
        got = allrec.view(world * B, stride)
        gather = {"steps": ksteps, "ms_per_step": 1e3 * tg / ksteps, "value": world * B * ksteps / tg, "unit": "QP solves/s",
                  "all_gather_ms": 1e3 * only / ksteps, "record_bytes": 8 * stride, "bytes_gathered_per_rank": 8 * world * B * stride,
                  "ranks_seen": int((got[::B, 0] == 20).sum().item()),
                  "note": "solve + device-side record packing + all_gather_into_tensor over RCCL; not part of `value`"}

        if gather["ranks_seen"] != world:
            # every rank's first record carries Exitflag 20: fewer means a rank's shard never arrived -- not a result to report
            raise SystemExit("bench.py: the all-gather returned the records of %d ranks, %d expected" % (gather["ranks_seen"], world))
        if not rehearsal:
            gather["native_rccl"] = native_rccl_gather(capi, torch, dist, batch, B, rank, world, local_rank, max(3, ksteps // 5))

    # BASELINE configs[4] (512 hs0xx QPs) as a weak- and a strong-scaling figure, on every rank, for every N
    scaling5 = hs_batch_scaling(capi, problems, parallel, torch, dist, rank, world, local_rank, cdev)

    # correctness guard: every QP solved, certificate green (outside the timed region)
    res = batch.results()
    ok, kkt = batch.test_optimality()
    n_bad = sum(1 for r, o in zip(res, ok) if r["status"] != 20 or o != 1)
    if dist is not None:
        tb = torch.tensor([n_bad], dtype=torch.int64, device=cdev)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        n_bad = int(tb.item())

    if rank == 0:
        total = world * B * args.steps
        k_ms = kernel_ms_total / args.steps   # average launch duration over the timed region
        bytes_launch = float(sum(qp_algorithmic_bytes(q) for q in probs))
        achieved = bytes_launch / (k_ms * 1e-3) / 1e9
        fcal, fcal_note = pmc_fetch_calibration()
        ffac = 2.0 if (fcal is not None and fcal > 1.5) else 1.0     # (the guide's x2 or nothing: the calibration decides which)
        kname, kkey, kdesc, knote = HEADLINE_KERNELS.get(batch.last_kernel(), HEADLINE_KERNELS[1])
        if batch.last_kernel() == 2 and args.keep_state:      # (the build that writes the state blocks back)
            kname = kkey = "lane_qp_kernel<2, true, true>"
        traffic, tfile = pmc_traffic(kkey, ffac) if B == 65536 else (None, None)
        line = {
            "metric": "QP-subproblem solves/sec", "value": total / elapsed, "unit": "QP solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "hs071-scale QP batch (derived hs071 first QP + seeded 1 %% perturbations, "
                                   "nV=8 x nC=2 via QPhandler [J I -I]), cold start, %d QPs/GPU per step" % B,
                       "qps_per_gpu": B, "engine": kdesc,
                       "keep_state": bool(args.keep_state),
                       "mean_nWSR": float(np.mean([r["nWSR"] for r in res])), "unsolved_or_kkt_fail": n_bad},
            "roofline": {"kernel": kname, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tfile,
                         "traffic_note": "from the committed rocprofv3 --pmc passes of this command, not measured in this run: "
                                         "FETCH_SIZE x %.0f + WRITE_SIZE. The x2 of MI355X_MICROARCH.md (FETCH_SIZE tallies 128-B requests at "
                                         "64 B) is calibrated there for 16-B-per-lane loads only; this kernel stages its inputs with 4- and "
                                         "8-byte-per-lane coalesced loads, so the factor is calibrated on a kernel of the same file with a "
                                         "known byte count: %s" % (ffac, fcal_note or "no calibration entry found: FETCH_SIZE taken as is"),
                         "fetch_size_factor": ffac,
                         "kernel_ms": k_ms, "algorithmic_bytes_per_launch": bytes_launch,
                         "note": knote},
            "kernel_ms_stats": quartiles(per_launch),
        }
        issue = pmc_issue_roofline(kkey) if B == 65536 else None
        if issue:
            line["roofline_lds"] = issue
        if gather is not None:
            line["with_gather"] = gather
        if scaling5 is not None:
            line["hs0xx_batch_scaling"] = scaling5
        if not args.no_extras and world == 1:   # extras (CPU baseline, secondary rooflines, large configs): N = 1 only
            import oracle as O
            if not NO_BUILD:
                O.use_native_build()
            line["cpu_baseline"] = cpu_baseline(probs[:256], args.cpu_seconds)
            line["speedup_vs_cpu_baseline"] = line["value"] / line["cpu_baseline"]["value"]
            try:
                line["native_rccl_1_rank"] = native_rccl_gather(capi, torch, None, batch, B, 0, 1, local_rank, 5)
            except capi.RsqpError as e:     # informational at N = 1 (no librccl on the host): say so, never hide it
                line["native_rccl_1_rank"] = {"error": str(e)}
            line["roofline_spmv"] = spmv_roofline(capi, problems, args.spmv_batch, 40)
            line["roofline_value_refresh"] = value_refresh_roofline(capi, problems)
            line["hs071_single_qp"] = hs071_single_qp_latency(problems)
            traj = hs071_trajectory_latency()
            if traj:
                line["hs071_trajectory_latency"] = traj
                # (the driver keeps `config` whole and drops other extras: the two per-SQP-iteration figures of BASELINE's
                #  metric -- "wall-clock per SQP iteration (hs071; n=10k sparse)" -- go there)
                line["config"]["hs071_us_per_sqp_iteration"] = {k: traj.get(k) for k in ("gpu", "cpu_oracle", "sqp_iterations",
                                                                                         "gpu_later_iterations", "cpu_oracle_later_iterations")}
            line["hs071_trajectory_batch"] = trajectory_batch(capi, problems)
            line["hs0xx_batch_512"] = hs_batch_config(capi, problems, parallel)
            if not args.no_large:
                line["large_engine"] = large_configs(capi, problems)
                le = line["large_engine"]
                ref, y0 = le.get("sparse_10000x20000_warm_sequence_reference_rule"), le.get("sparse_10000x20000_warm_sequence_y0_rule")
                if ref:
                    line["config"]["sparse10k_s_per_sqp_iteration_reference_rule"] = {
                        "mean": 1e-3 * ref["wall_ms_per_sqp_iteration_mean"],
                        "fixed": 1e-3 * ref["fixed_matrix_steps"]["wall_ms_mean"] if ref.get("fixed_matrix_steps") else None,
                        "varied": 1e-3 * ref["varied_matrix_steps"]["wall_ms_mean"] if ref.get("varied_matrix_steps") else None,
                        "steps": ref["qps"], "all_certified": ref["all_certified"]}
                if y0:
                    line["config"]["sparse10k_s_per_sqp_iteration_y0_rule_opt_in"] = 1e-3 * y0["wall_ms_per_sqp_iteration_mean"]
                d2 = le.get("dense_2048x4096_cold")
                if d2:
                    line["config"]["dense_2048x4096_cold_s"] = d2["seconds"]
                if "roofline_mfma" in le:
                    line["config"]["roofline_mfma_frac"] = le["roofline_mfma"]["frac"]
            line["config"]["roofline_spmv_frac"] = line["roofline_spmv"]["frac"]
            line["config"]["hs0xx_batch_512_ms"] = line["hs0xx_batch_512"]["512_qps"]["ms_per_batch"]
            line["config"]["hs0xx_batch_64_shard_ms"] = line["hs0xx_batch_512"]["64_qps_shard_of_8_gpus"]["ms_per_batch"]
        text = compact_line(line)
        write_extras(line)
        print("[bench extras] " + json.dumps(line), file=sys.stderr)
        sys.stdout.flush()
        sys.stderr.flush()
        os.write(result_fd, (text + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
