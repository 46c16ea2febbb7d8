"""GPU tests that need TWO devices.  Leases of this project have been one GPU, so these have never executed (DESIGN.md §4 lists
what has and has not run at N > 1); they are written so that they run unattended the first time `torch.cuda.device_count() >= 2`:
  * the online serving pool with two GPU daemons — request i on rank i % 2, each daemon pinned to its own GPU, DP-2 PCM ==
    single-GPU PCM byte for byte (launch.py:183-279, 355-415, 460-474 of the reference);
  * RCCL between two GPUs: worker/dp_pool.broadcast_weights under torch.distributed.run, checksums equal on both ranks;
  * bench.py --gpus 2 under the driver's own launcher: ONE JSON line, n_gpus 2, weak scaling;
  * the detokenizer on cuda:1 while the LM runs on cuda:0, across a slot reset (worker/base.py:641-644 of the reference).
"""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs two GPUs")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(nproc, script_args, timeout):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", "29531"] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(900)
def test_two_gpu_daemons_serve_the_same_pcm_as_one_gpu():
    from tests.test_gpu_worker import build, serve
    from vox_serve_amd.launch import ServingPool
    prompt = [1, 2, 3, 40, 41, 42, 43, 7, 8, 9, 10, 11]
    prompts = {f"r{i}": prompt[:3] + [40 + i, 50 + i] + prompt[-5:] for i in range(6)}
    # the reference result: every request ALONE on one GPU (a request's bits depend on the rows it shares a launch with; rank r of
    # the pool sees requests r, r + 2, r + 4 — served here with the same compositions by replaying each rank's share in process)
    want = {}
    for r in range(2):
        m, _ = build(torch.device("cuda:0"), max_tokens=30)
        share = {rid: ids for i, (rid, ids) in enumerate(prompts.items()) if i % 2 == r}
        out, _ = serve(m, share)
        want.update({rid: o["pcm"] for rid, o in out.items()})
        m.engine.close(); m.audio_decoder.close()
    pool = ServingPool("tiny", dp_size=2, max_batch_size=4, page_size=16, max_num_pages=64,
                       worker_factory="tests.dp_tiny_qwen3_worker:make", ready_timeout_s=600.0)
    try:
        assert sorted(pool.ready) == [0, 1]
        assert [pool.ready[r]["visible_devices"] for r in (0, 1)] == ["0", "1"]          # one GPU each, set before torch was imported
        assert all(pool.ready[r]["torch_devices"] == 1 and pool.ready[r]["device"] == "cuda:0" for r in (0, 1))
        rids = []
        import time
        for rid, ids in prompts.items():          # one at a time: the router's counter order is the submission order
            rids.append(pool.start_streaming_request("", model_kwargs={"prompt_token_ids": ids, "speaker": "a"}, request_id=rid, block=True))
            t0 = time.time()
            while pool.request_info(rid)["rank"] is None and time.time() - t0 < 10:
                time.sleep(0.002)
        got = {rid: b"".join(pool.stream(rid, timeout_s=300)) for rid in rids}
        assert [pool.request_info(r)["rank"] for r in rids] == [i % 2 for i in range(len(rids))]
        assert all(pool.completion(r)["status"] == "completed" for r in rids)
    finally:
        pool.cleanup()
    assert got == want


@pytest.mark.timeout(600)
def test_rccl_broadcast_weights_between_two_gpus():
    p = _torchrun(2, ["tests/dp_rccl_check.py"], 500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["ok"] and line["world"] == 2


@pytest.mark.timeout(1500)
def test_bench_line_on_two_gpus_under_the_drivers_launcher():
    p = _torchrun(2, ["bench.py", "--gpus", "2", "--steps", "6", "--warmup", "2", "--sub-batches", "", "--no-cpu-baseline", "--ttfa-requests", "0",
                      "--serving-ttfa-requests", "0", "--serving-modes", "", "--no-other-configs", "--no-kv-sweep"], 1400)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                                       # rank 0 prints ONE line
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["value"] > 0 and r["steps"] == 6


@pytest.mark.timeout(900)
def test_detokenizer_on_the_second_gpu_across_a_slot_reset():
    """The LM on cuda:0, the codec on cuda:1: two waves of requests through ONE model, so that the second wave's chunks run in
    streaming slots that were reset after the first wave (the fence of the reset waits on the detokenizer device's stream)."""
    from tests.test_gpu_worker import build, serve
    prompt = [1, 2, 3, 40, 41, 42, 43, 7, 8, 9, 10, 11]
    waves = [{f"a{i}": prompt[:3] + [40 + i] + prompt[-5:] for i in range(4)}, {f"b{i}": prompt[:3] + [60 + i, 61] + prompt[-5:] for i in range(4)}]
    res = []
    for dec in (None, "cuda:1"):
        m, _ = build(torch.device("cuda:0"), max_tokens=30, audio_decoder_device=dec)
        out = {}
        for w in waves:
            o, wk = serve(m, w)
            out.update({k: (v["pcm"], v["done"]) for k, v in o.items()})
        if dec:
            assert str(wk.detokenizer_device) == "cuda:1" and wk._detok_stream.device == torch.device("cuda:1")
        res.append(out)
        m.engine.close(); m.audio_decoder.close()
    assert res[0] == res[1]
    assert torch.cuda.current_device() == 0
