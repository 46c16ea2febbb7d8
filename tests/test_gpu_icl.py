"""GPU: voice cloning through the Qwen3-TTS plugin — the prompt layout + input_features of x-vector-only and ICL cloning
equal the reference's `preprocess` (g14) bit for bit, and an ICL request driven through ModelWorker produces the oracle's
token stream."""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def build_base(dev, golden_ids, **override):
    from oracle import qwen3_codec_ref as CR, qwen3_ref as QR, voxref as vr
    from tests.test_gpu_codec import small_cfg
    from tests.test_gpu_qwen3 import to_engine_cfg
    from vox_serve_amd.model.qwen3_tts import Qwen3TTSModel, Qwen3TTSTokens
    from vox_serve_amd.sampling import SamplingConfig
    from vox_serve_amd.tokenizer.qwen3_codec import Qwen3CodecConfig
    ids = dict(golden_ids, **override)
    spk_id, dialect, lang = ids.pop("spk_id"), ids.pop("spk_is_dialect"), ids.pop("codec_language_id")
    cfg = QR.tiny_cfg()
    Wn = QR.random_weights(cfg, seed=0, std=0.08)
    W = {k: vr.to_torch(v).to(dev) for k, v in Wn.items()}
    cc = small_cfg()
    pc = Qwen3CodecConfig(**{k: getattr(cc, k) for k in Qwen3CodecConfig.__dataclass_fields__})
    toks = Qwen3TTSTokens(**ids, codec_eos=cfg.eos_id, codec_language_id=lang, spk_id=spk_id, spk_is_dialect=dialect)
    m = Qwen3TTSModel("tiny-base", W, CR.random_codec_weights(cc, 3), config=to_engine_cfg(cfg), codec_config=pc, tokens=toks,
                      device=str(dev), detokenize_interval=4, max_batch_size=4, page_size=16, max_num_pages=64,
                      max_seq_len=512, max_prefill_tokens=64, tts_model_type="base")
    m.default_sampling_config = SamplingConfig(greedy=True, max_tokens=None)
    return m, cfg, Wn


def test_clone_prompts_equal_the_reference_preprocess(golden):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import voxref as vr
    dev = torch.device("cuda:0")
    g = golden("g14_qwen3_preprocess")
    m, cfg, _ = build_base(dev, json.loads(str(g["special_ids"])))
    spk = vr.to_torch(g["spk_embedding"])
    codes = torch.from_numpy(g["ref_codes"])
    n = 0
    for tag, kind, kw in json.loads(str(g["cases"])):
        if kind != "base":
            continue
        icl = not kw.get("x_vector_only_mode")
        po = m.preprocess(prompt_token_ids=g[f"{tag}_prompt_ids"].tolist(), language=kw["language"],
                          instruct_token_ids=g[f"{tag}_instruct_ids"].tolist() if f"{tag}_instruct_ids" in g else None,
                          is_input_streaming=bool(kw.get("is_input_streaming")), x_vector_only_mode=not icl,
                          speaker_embedding=spk, ref_codes=codes if icl else None,
                          ref_text_token_ids=g[f"{tag}_ref_text_ids"].tolist() if icl else None)
        assert torch.equal(po.input_tokens, torch.from_numpy(g[f"{tag}_tokens"])), tag
        assert torch.equal(po.input_masks, torch.from_numpy(g[f"{tag}_masks"])), tag
        assert np.array_equal(vr.from_torch(po.input_features), g[f"{tag}_features"]), tag
        n += 1
    assert n == 4
    with pytest.raises(ValueError):
        m.preprocess(prompt_token_ids=[1, 2, 3, 4, 5, 6, 7, 8, 9])          # cloning without a reference voice
    m.engine.close()


def test_icl_request_through_the_worker_equals_the_oracle(golden):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import qwen3_ref as QR, voxref as vr
    from tests.test_gpu_worker import _drive_worker
    dev = torch.device("cuda:0")
    g = golden("g14_qwen3_preprocess")
    # text-side ids inside the tiny text vocabulary (512 rows); the layout itself is pinned by the test above
    m, cfg, Wn = build_base(dev, json.loads(str(g["special_ids"])), tts_bos=5, tts_eos=6, tts_pad=7)
    kw = {"prompt_token_ids": [11, 12, 13] + list(range(100, 109)) + [21, 22, 23, 24, 25], "language": "english",
          "speaker_embedding": vr.to_torch(g["spk_embedding"]).float().tolist(), "ref_codes": g["ref_codes"].tolist(),
          "ref_text_token_ids": [11, 12, 13] + list(range(200, 206)) + [21, 22]}
    po = m.preprocess(**kw)
    assert po.input_tokens.shape[0] == 3 + 4 + 1 + 1 + 6 + 9 + 1 + 1 + 7
    reqs, w = _drive_worker(m, [kw], 6)
    ref = QR.Qwen3Ref(cfg, Wn, page_size=16, max_pages=64, max_batch=1)
    q = QR.RefRequest()
    toks = po.input_tokens.numpy().astype(np.int32)
    lg, hid = ref.prefill(q, toks, po.input_masks[:, -1].numpy().astype(np.uint8), vr.from_torch(po.input_features))
    frames = [ref.frame([q], lg, hid)[0][0]]
    for _ in range(6):
        frames.append(ref.frame([q])[0][0])
    got = [t[0, :cfg.n_groups].tolist() for t in reqs[0].lm_output_tokens]
    want = [f[:cfg.n_groups].tolist() for f in frames[: len(got)]]
    assert len(got) >= 5 and got == want, (got, want)
    m.engine.close()


def test_clone_request_with_a_reference_clip_runs_both_encoders_and_is_served(golden, tmp_path):
    """audio_path -> speaker encoder + codec encoder -> ICL prompt: equals the prompt built from the oracle encoders' outputs
    (codes equal; the bf16 speaker row within one bf16 ulp), and the request is served end to end through the scheduler."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import dataclasses
    import wave
    from oracle import codec_enc_ref as ER, spk_ref as SR, voxref as vr
    from tests.test_gpu_codec_encoder import _plugin as cenc_plugin
    from tests.test_gpu_spkenc import _plugin as spk_plugin
    dev = torch.device("cuda:0")
    g = golden("g14_qwen3_preprocess")
    m, cfg, _ = build_base(dev, json.loads(str(g["special_ids"])), tts_bos=5, tts_eos=6, tts_pad=7)
    scfg = dataclasses.replace(SR.tiny_spk_cfg(), enc_dim=cfg.talker.hidden)
    ecfg = ER.tiny_codec_enc_cfg()
    assert ecfg.valid_quantizers == cfg.n_groups and ecfg.codebook_size <= cfg.depth_vocab
    Ws, We = SR.random_spk_weights(scfg, seed=11), ER.random_codec_enc_weights(ecfg, seed=12)
    m.speaker_encoder, m.audio_encoder = spk_plugin(scfg, Ws, dev), cenc_plugin(ecfg, We, dev)
    # the clip as a 16-bit wav file (what a client uploads); both sides read the same file
    pcm = (SR.test_audio(21, 9000) * 32767.0).astype("<i2")
    path = str(tmp_path / "ref.wav")
    with wave.open(path, "wb") as f:
        f.setnchannels(1), f.setsampwidth(2), f.setframerate(24000), f.writeframes(pcm.tobytes())
    audio, sr = m._load_audio_to_np(path)
    assert sr == 24000 and np.array_equal(audio, pcm.astype(np.float32) / 32768.0)
    kw = {"prompt_token_ids": [11, 12, 13] + list(range(100, 109)) + [21, 22, 23, 24, 25], "language": "english",
          "ref_text_token_ids": [11, 12, 13] + list(range(200, 206)) + [21, 22]}
    po = m.preprocess(audio_path=path, **kw)
    o_spk = torch.from_numpy(SR.SpkRef(scfg, Ws).embed(audio)).to(torch.bfloat16)
    o_codes = ER.CodecEncRef(ecfg, We).encode(torch.from_numpy(audio))
    assert o_codes.shape == (-(-9000 // ecfg.hop), cfg.n_groups)
    po2 = m.preprocess(speaker_embedding=o_spk, ref_codes=o_codes, **kw)
    assert torch.equal(po.input_tokens, po2.input_tokens) and torch.equal(po.input_masks, po2.input_masks)
    a, b = vr.from_torch(po.input_features).astype(np.int32), vr.from_torch(po2.input_features).astype(np.int32)
    spk_row = 3 + 4
    other = np.ones(a.shape[0], bool)
    other[spk_row] = False
    assert np.array_equal(a[other], b[other])                         # ICL rows: codes equal -> bit-equal sums
    assert np.abs(a[spk_row] - b[spk_row]).max() <= 1 and (a[spk_row] == b[spk_row]).mean() > 0.97
    # x-vector-only cloning from the same clip: no reference-code rows
    po3 = m.preprocess(audio_path=path, x_vector_only_mode=True, **kw)
    assert po3.input_tokens.shape[0] == po.input_tokens.shape[0] - o_codes.shape[0] - 6
    # served end to end: JSON request with the clip's path
    from vox_serve_amd.sampling import SamplingConfig
    m.default_sampling_config = SamplingConfig(greedy=True, max_tokens=po.input_tokens.shape[0] + 9)
    from vox_serve_amd.scheduler import QueueTransport, Scheduler, encode_request
    from vox_serve_amd.worker import ModelWorker
    t = QueueTransport()
    w = ModelWorker(model=m, max_batch_size=4, max_num_pages=64, page_size=16, device=m.device)
    s = Scheduler(w, max_batch_size=4, transport=t)
    t.requests.put(encode_request("clone", "", model_kwargs=kw, audio_path=path))
    s.run_until_idle(500)
    pcm_out, done = b"", None
    while not t.results.empty():
        rid, kind, body = t.results.get().split(b"|", 2)
        if kind == b"AUDIO":
            pcm_out += body
        else:
            done = json.loads(body)
    assert done is not None and done["status"] == "completed" and len(pcm_out) > 0
    m.engine.close()
