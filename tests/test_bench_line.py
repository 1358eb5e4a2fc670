"""The line bench.py prints must survive the driver's record: one JSON object, below 4 KB, with the contract keys, `roofline`
and `cpu_baseline` (the 20 KB line of round 4 was cut and parsed as nothing). Built here from canned numbers --
the full record of round 4's last builder run (tests/golden/bench_full_record_r04.json) -- no GPU involved."""
import json
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def canned():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "bench_full_record_r04.json")))


def test_line_is_small_and_carries_the_contract():
    full = canned()
    text = bench.compact_line(full)
    assert "\n" not in text and len(text) < bench.MAX_LINE_BYTES <= 4096
    line = json.loads(text)
    for k in CONTRACT:
        assert k in line, k
    assert line["value"] == pytest.approx(full["value"], rel=1e-6)
    assert line["steps"] == full["steps"] and line["warmup"] == full["warmup"] and line["n_gpus"] == 1
    assert isinstance(line["config"]["workload"], str) and "model" not in line["config"]
    rf = line["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-4)
    assert rf["achieved"] == pytest.approx(1e-6 * rf["algorithmic_bytes_per_launch"] / rf["kernel_ms"], rel=1e-4)
    cb = line["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] == 1 and cb["value"] > 0 and cb["sample"]
    # every value of config is a scalar (the driver keeps config whole: prose and nested samples belong in the sidecar)
    assert all(not isinstance(v, (dict, list)) for v in line["config"].values())
    for k in ("hs071_us_per_sqp_iteration_gpu", "sparse10k_s_per_sqp_iteration", "dense_2048x4096_cold_s", "hs0xx_batch_512_ms",
              "roofline_mfma_frac", "roofline_spmv_frac"):
        assert isinstance(line["config"][k], float), k


def test_line_without_extras_and_with_gather_stays_small():
    full = canned()
    for k in ("cpu_baseline", "speedup_vs_cpu_baseline", "large_engine", "roofline_spmv", "hs071_single_qp", "hs071_trajectory_latency"):
        full.pop(k, None)
    full["n_gpus"] = 8
    full["with_gather"] = {"steps": 50, "ms_per_step": 0.1, "value": 5e9, "unit": "QP solves/s", "all_gather_ms": 0.03,
                           "ranks_seen": 8, "note": "x" * 5000, "native_rccl": {"a": "y" * 5000}}
    line = json.loads(bench.compact_line(full))
    assert "cpu_baseline" not in line and line["with_gather"]["ranks_seen"] == 8
    assert len(json.dumps(line)) < 2048


def test_oversized_line_is_refused():
    full = canned()
    full["config"]["workload"] = "w" * 5000
    with pytest.raises(AssertionError):
        bench.compact_line(full)
