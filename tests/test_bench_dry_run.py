"""`bench.py --gpus 2` launched the way the driver launches N > 1 (torch.distributed.run, one rank per GPU), with no GPU: the
--dry-run switch swaps the engine loop for a stub and RCCL for gloo, everything else — rank launch, weight broadcast,
barriers, max-over-ranks timing, rank-0 JSON line — is the code the first 8-GPU lease will execute."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


@pytest.mark.timeout(300)
def test_bench_two_ranks_gloo_dry_run_under_the_drivers_launcher():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
           "--dry-run"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=ROOT,
                       env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    lines = [ln for ln in p.stdout.strip().split("\n") if ln.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, p.stdout[-1500:] + p.stderr[-1500:]      # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out["dry_run"] is True and "not a measurement" in out["data"]
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["steps"] == 5 and out["warmup"] == 2
    assert out["scaling"] == "weak" and out["higher_is_better"] is True and out["vs_baseline"] is None
    assert out["weight_broadcast_rccl"]["replicas_identical"] is True          # rank 1 started from other weights
    assert out["ms_per_step"] >= 1.0                                           # max over ranks of >= 1 ms stub steps
    for k in ("batch8", "batch32", "roofline"):
        assert k in out
    # the serving mode of an N > 1 run goes through the online pool: two daemons, requests alternate between them
    pool = out["serving_pool_dp"]
    assert "error" not in pool, pool
    assert pool["dp_size"] == 2 and pool["ranks_used"] == [0, 1] and pool["requests"] == 16 and pool["value"] > 0
    # whole-job aggregate: 2 ranks x 1 request x 1920 samples per step
    assert abs(out["value"] - 2 * 1920 / (out["ms_per_step"] * 1e-3)) / out["value"] < 1e-6


@pytest.mark.timeout(300)
def test_bench_self_launch_path_dry_run():
    """`python bench.py --gpus 2` with no launcher re-executes itself under torch.distributed.run."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run",
                        "--sub-batches", ""], capture_output=True, text=True, timeout=280, cwd=ROOT)
    lines = [ln for ln in p.stdout.strip().split("\n") if ln.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, p.stdout[-1500:] + p.stderr[-1500:]
    assert json.loads(lines[0])["ranks_seen"] == 2
