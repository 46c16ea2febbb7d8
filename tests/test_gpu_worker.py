"""GPU: the whole serving loop — Scheduler -> ModelWorker -> Qwen3TTSModel (native engine + native codec) — on a tiny
synthetic model: wire format, streaming chunks, trim rule, resource release, determinism, and agreement of the
worker-driven token stream with a hand-driven engine (the path the oracle parity tests pin)."""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def build(dev, max_tokens=None):
    from oracle import qwen3_ref as QR, voxref as vr           # weights recipe only (test side)
    from tests.test_gpu_codec import small_cfg
    from oracle import qwen3_codec_ref as CR
    from tests.test_gpu_qwen3 import to_engine_cfg
    from vox_serve_amd.model.qwen3_tts import Qwen3TTSModel, Qwen3TTSTokens
    from vox_serve_amd.sampling import SamplingConfig
    from vox_serve_amd.tokenizer.qwen3_codec import Qwen3CodecConfig
    cfg = QR.tiny_cfg()
    cfg.n_groups = 4
    W = {k: vr.to_torch(v).to(dev) for k, v in QR.random_weights(cfg, 7, 0.08).items()}
    cc = small_cfg()
    pc = Qwen3CodecConfig(**{k: getattr(cc, k) for k in Qwen3CodecConfig.__dataclass_fields__})
    toks = Qwen3TTSTokens(tts_bos=5, tts_eos=6, tts_pad=cfg.tts_pad_id, codec_bos=10, codec_eos=cfg.eos_id, codec_pad=11,
                          codec_think=12, codec_nothink=13, codec_think_bos=14, codec_think_eos=15,
                          codec_language_id={"english": 16}, spk_id={"a": 17})
    m = Qwen3TTSModel("tiny", W, CR.random_codec_weights(cc, 3), config=to_engine_cfg(cfg), codec_config=pc, tokens=toks,
                      device=str(dev), detokenize_interval=4, max_batch_size=4, page_size=16, max_num_pages=64,
                      max_seq_len=512, max_prefill_tokens=64)
    m.default_sampling_config = SamplingConfig(greedy=True, max_tokens=max_tokens, repetition_penalty=1.05, repetition_window=-1)
    return m, cfg


def serve(m, prompts):
    from vox_serve_amd.scheduler import QueueTransport, Scheduler, encode_request
    from vox_serve_amd.worker import ModelWorker
    t = QueueTransport()
    w = ModelWorker(model=m, max_batch_size=4, max_num_pages=64, page_size=16, device=m.device)
    s = Scheduler(w, max_batch_size=4, transport=t)
    for rid, ids in prompts.items():
        t.requests.put(encode_request(rid, "", model_kwargs={"prompt_token_ids": ids, "speaker": "a"}))
    s.run_until_idle(2000)
    out = {rid: {"pcm": b"", "done": None} for rid in prompts}
    while not t.results.empty():
        msg = t.results.get()
        rid, kind, body = msg.split(b"|", 2)
        if kind == b"AUDIO":
            out[rid.decode()]["pcm"] += body
        else:
            out[rid.decode()]["done"] = json.loads(body)
    return out, w


def test_scheduler_worker_engine_codec_end_to_end():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = torch.device("cuda:0")
    m, cfg = build(dev, max_tokens=30)
    prompt = [1, 2, 3, 40, 41, 42, 43, 7, 8, 9, 10, 11]           # 3 role + 4 text + 5 template tail
    out, w = serve(m, {"r1": prompt, "r2": prompt[:3] + [50, 51] + prompt[-5:]})
    for rid, o in out.items():
        assert o["done"]["status"] == "completed" and o["done"]["reason"] in ("max_tokens_reached", "stop_id_encountered")
        assert len(o["pcm"]) % 2 == 0 and len(o["pcm"]) > 0
        pcm = np.frombuffer(o["pcm"], dtype=np.int16)
        assert np.abs(pcm).max() > 50                                # audible signal, int16 range
    assert w.empty_pages.qsize() == 64                               # KV pages returned
    assert len(m.audio_decoder._free_slots) == m.audio_decoder.max_slots   # codec slots returned
    # determinism: the same arrivals give byte-identical audio
    out2, _ = serve(m, {"r1": prompt, "r2": prompt[:3] + [50, 51] + prompt[-5:]})
    assert all(out[r]["pcm"] == out2[r]["pcm"] for r in out)
    # a request served alone: frames generated == max_tokens rule, samples = frames*hop with the reference's trim
    solo, w2 = serve(m, {"r1": prompt})
    hop = m.audio_decoder.hop
    n_prompt = 3 + 3 + 1 + 1 + 4 + 2
    frames_audio = len(solo["r1"]["pcm"]) // 2 / hop
    assert 1 <= frames_audio <= 30 - n_prompt + 1
    m.engine.close(); m.audio_decoder.close()
