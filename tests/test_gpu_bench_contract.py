"""GPU: bench.py keeps the driver's contract — one JSON line on stdout with the agreed keys, the roofline object, sane values —
and its multi-GPU code path really runs on RCCL: ONE short run, started the way the driver starts an N > 1 run
(`python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1 --force-dist`), so the `nccl` process group is
initialised, the weight arena goes through `broadcast_weights`, and the barrier / max-reduce around the timed region are the
collective ones.  The numbers themselves are not asserted."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.slow]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract_with_the_rccl_path_exercised_at_world_1():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "2", "--force-dist",
           "--ttfa-requests", "1", "--serving-ttfa-requests", "3", "--serving-modes", "ttfa", "--no-cpu-baseline",
           "--no-other-configs", "--no-kv-sweep", "--sub-batches", "8", "--exact-rows", "2"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]            # exactly ONE JSON line on stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 10 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"] and "exact_rows 2" in d["config"]["workload"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic_source"] is None or r["traffic_source"].startswith("recorded")
    assert d["value"] > 1e5 and d["ms_per_step"] > 0 and d["ttfa_ms_p50"] > 0 and d["ttfa_ms_p50_engine"] > 0
    assert "Scheduler" in d["ttfa_path"]
    assert abs(d["value"] - 1920 * 10 / (d["ms_per_step"] * 10e-3)) / d["value"] < 1e-6       # samples of exactly K steps / their time
    sub = d["batch8"]                                      # sub-result timed in the same run
    assert sub["batch_per_gpu"] == 8 and sub["value"] > d["value"] and 0 < sub["roofline"]["frac"] < 1
    assert abs(sub["value"] - 8 * 1920 * 10 / (sub["ms_per_step"] * 10e-3)) / sub["value"] < 1e-6
    # the RCCL path: group up, every collective of the multi-GPU run executed
    assert d["ranks_seen"] == 1
    b = d["weight_broadcast_rccl"]
    assert b["replicas_identical"] is True and b["bytes"] > 3e9 and b["GBps"] > 0


def test_bench_explicit_batch_32_line():
    """`--batch 32` (BASELINE config 3's batch size as the headline of the line): value = 32 requests x 1920 samples per step over
    the timed steps, no sub-results, roofline present."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "32", "--steps", "10", "--warmup", "2", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["config"]["batch_per_gpu"] == 32 and "batch8" not in d and "batch32" not in d and "kv_sweep" not in d
    assert abs(d["value"] - 32 * 1920 * 10 / (d["ms_per_step"] * 10e-3)) / d["value"] < 1e-6
    assert 0 < d["roofline"]["frac"] < 1 and d["roofline"]["algorithmic_bytes_per_launch"] > 3.5e9
