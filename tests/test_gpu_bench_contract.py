"""GPU: bench.py keeps the driver's contract — one JSON line on stdout with the agreed keys, the roofline and (at N=1) the
cpu_baseline objects, sane values.  A short run (3 timed steps); the numbers themselves are not asserted."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", *extra],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines            # exactly ONE line on stdout
    return json.loads(lines[0])


def test_bench_line_has_the_contract_keys():
    d = _run("--ttfa-requests", "1", "--serving-ttfa-requests", "3", "--no-cpu-baseline", "--no-other-configs")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["ttfa_ms_p50"] > 0 and d["ttfa_ms_p50_detokenize_interval_2"] > 0
    assert "Scheduler" in d["ttfa_path"] and d["ttfa_ms_p50_under_32way_load"] > 0 and d["ttfa_ms_p50_engine"] > 0
    assert d["ranks_seen"] == 1 and d["roofline"]["traffic_source"] is None or d["roofline"]["traffic_source"].startswith("recorded")
    for b in (8, 32):      # the headline metric is quoted at batch 1 / 8 / 32: sub-results timed in the same run
        sub = d[f"batch{b}"]
        assert sub["batch_per_gpu"] == b and sub["value"] > d["value"] and 0 < sub["roofline"]["frac"] < 1
        assert abs(sub["value"] - b * 1920 * 3 / (sub["ms_per_step"] * 3e-3)) / sub["value"] < 1e-6
    assert abs(d["value"] - 1920 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6      # samples of exactly K steps / their time


def test_bench_batched_line_and_exact_rows_flag():
    d = _run("--batch", "8", "--ttfa-requests", "0", "--no-cpu-baseline", "--exact-rows", "8")
    assert "batch32" not in d          # an explicit --batch runs that batch size only
    assert d["config"]["batch_per_gpu"] == 8 and "exact_rows 8" in d["config"]["workload"]
    assert abs(d["value"] - 8 * 1920 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6
