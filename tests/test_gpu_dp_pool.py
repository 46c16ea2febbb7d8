"""GPU: the online serving pool with a real daemon — a fresh interpreter pinned to the GPU through HIP_VISIBLE_DEVICES before
it imports torch, serving the tiny Qwen3-TTS + codec over the AF_UNIX transports; the PCM a request gets through the pool
equals, byte for byte, what the in-process scheduler gives it (launch.py / scheduler_entry.py of the reference:
183-279, 355-415, 460-474; 1-105)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_one_gpu_daemon_serves_the_same_pcm_as_the_in_process_scheduler():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from tests.test_gpu_worker import build, serve
    from vox_serve_amd.launch import ServingPool
    prompt = [1, 2, 3, 40, 41, 42, 43, 7, 8, 9, 10, 11]
    prompts = {"r0": prompt, "r1": prompt[:3] + [50, 51] + prompt[-5:], "r2": prompt[:3] + [60] + prompt[-5:]}
    m, _ = build(torch.device("cuda:0"), max_tokens=30)
    want, _ = serve(m, prompts)
    m.engine.close(); m.audio_decoder.close()
    pool = ServingPool("tiny", dp_size=1, max_batch_size=4, page_size=16, max_num_pages=64,
                       worker_factory="tests.dp_tiny_qwen3_worker:make", extra_env={"VOX_TRANSPORT": "ipc"}, ready_timeout_s=300.0)
    try:
        info = pool.ready[0]
        assert info["device"] == "cuda:0" and info["torch_devices"] == 1 and info["visible_devices"] == "0"
        rids = [pool.start_streaming_request("", model_kwargs={"prompt_token_ids": ids, "speaker": "a"}, request_id=rid, block=True)
                for rid, ids in prompts.items()]
        got = {rid: b"".join(pool.stream(rid, timeout_s=240)) for rid in rids}
        for rid in rids:
            assert pool.completion(rid)["status"] == "completed" and pool.request_info(rid)["rank"] == 0
    finally:
        pool.cleanup()
    # one prefill per step in arrival order on both sides: the same batch compositions, hence the same bits
    assert {r: got[r] for r in got} == {r: want[r]["pcm"] for r in want}
