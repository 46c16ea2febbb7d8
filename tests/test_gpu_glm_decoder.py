"""GPU: the GLM-4-Voice detokenizer (speech tokens -> mel -> waveform) through the C ABI against the reference modules' output
(tests/golden/g13_glm_decoder.npz) and the CPU oracle (oracle/glm_dec_ref.py, pinned to the same fixture).
Tolerances: mel RMS error <= 1e-4 (mel RMS ~ 1.2).  Waveform: <= 3e-4 — this vocoder's SineGen v1 accumulates f0 * h / sr over all
44 032 samples of the window, so an f0 difference of 1e-4 Hz moves the phase of the 9th harmonic by 1e-2 rad at the window's end: two
fp32 implementations of the reference itself (oracle vs module) already sit 8e-5 apart."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rms(x):
    return float(np.sqrt(np.mean(np.asarray(x, np.float64) ** 2)))


def glm_flow_plugin_cfg(c):
    from vox_serve_amd.tokenizer.glm import GLMFlowConfig
    return GLMFlowConfig(vocab_size=c.vocab, dim=c.dim, mel=c.mel, spk_embed_dim=c.spk_dim, enc_layers=c.enc_layers, enc_heads=c.enc_heads,
                         enc_ffn=c.enc_ffn, block_size=c.block_size, est_channels=c.est_ch, est_heads=c.est_heads, est_head_dim=c.est_head_dim,
                         est_blocks=c.est_blocks, est_mid_blocks=c.est_mid, n_timesteps=c.n_steps, inference_cfg_rate=c.cfg_rate)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_glm_decoder_matches_reference_modules(dev, golden, tag):
    from oracle import glm_dec_ref as GR, hift_ref as HR
    from tests.test_gpu_hift import to_plugin_cfg
    from vox_serve_amd.tokenizer.glm import GLMAudioDecoder
    g = golden("g13_glm_decoder")
    fc, hc = (GR.tiny_glm_flow_cfg(), GR.glm_hift_cfg(base_channels=128, f0_channels=64)) if tag == "tiny" else (GR.GlmFlowCfg(), GR.glm_hift_cfg())
    Wf, Wh = GR.random_glm_flow_weights(fc, seed=5), HR.random_hift_weights(hc, seed=6)
    pc = to_plugin_cfg(hc)
    pc.sine_gen_v1 = True
    seed = int(g["noise_seed"])
    dec = GLMAudioDecoder(Wf, Wh, device=dev, flow_config=glm_flow_plugin_cfg(fc), hift_config=pc, max_batch=2, seed=seed)
    tok = torch.from_numpy(g[f"{tag}_token"]).long()
    B, T = tok.shape
    Tm = fc.mel_len(T)
    z = GR.glm_cfm_noise(seed, 0, B, fc.mel, Tm)
    ini, nz = HR.make_noise(hc, B, Tm, seed=seed, first_stream=8)
    mel = dec.flow.inference(tok, noise=z).cpu().numpy()
    mel_s = dec.flow.inference(tok, first_stream=0).cpu().numpy()                      # the seeded device streams == the noise handed in
    three = dec.flow.inference(torch.cat([tok, tok[:1]]), noise=torch.cat([z, z[:1]])).cpu().numpy()      # 3 requests, max_batch 2
    assert mel.shape == (B, fc.mel, 172) and rms(g[f"{tag}_mel"]) > 0.5
    assert rms(mel - g[f"{tag}_mel"]) < 1e-4 and rms(mel_s - mel) < 1e-5, (rms(mel - g[f"{tag}_mel"]), rms(mel_s - mel))
    assert np.array_equal(three[:2], mel) and np.array_equal(three[2], mel[0])
    with torch.no_grad():
        mel_o = GR.GlmFlowRef(fc, Wf).inference(tok, z).numpy() if tag == "tiny" else g[f"{tag}_mel"]
    assert rms(mel - mel_o) < 1e-4
    wav = dec.forward(tok, flow_noise=z, hift_noise=nz, hift_rand_ini=ini).cpu().numpy()
    wav_s = dec.forward(tok, first_stream=0, hift_stream_base=8 + 2 * torch.arange(B, dtype=torch.int32)).cpu().numpy()
    assert wav.shape == (B, 44032) and rms(g[f"{tag}_wav"]) > 0.05
    assert rms(wav - g[f"{tag}_wav"]) < 3e-4 and rms(wav_s - g[f"{tag}_wav"]) < 3e-4, (rms(wav - g[f"{tag}_wav"]), rms(wav_s - g[f"{tag}_wav"]))
    dec.close()


def test_glm_served_end_to_end_with_flow_and_hift(dev):
    """Scheduler -> ModelWorker -> GLMVoiceModel (native LM engine) -> GLMAudioDecoder: 25-token windows of audio tokens become
    44032-sample AUDIO messages; two service runs give the same bytes."""
    import json
    from oracle import glm_dec_ref as GR, hift_ref as HR, lm_ref as LR, voxref as vr
    from tests.test_gpu_hift import to_plugin_cfg
    from vox_serve_amd.model.glm_voice import GLMVoiceConfig, GLMVoiceModel
    from vox_serve_amd.sampling import SamplingConfig
    from vox_serve_amd.scheduler import QueueTransport, Scheduler, encode_request
    from vox_serve_amd.worker import ModelWorker
    cfg = LR.tiny_glm_cfg()
    c = cfg.stack
    S = {k: vr.to_torch(v).to(dev) for k, v in LR.random_glm_state_dict(cfg, seed=3, std=0.08).items()}
    fc, hc = GR.tiny_glm_flow_cfg(), GR.glm_hift_cfg(base_channels=128, f0_channels=64)
    audio_offset = 40                    # most ids of the tiny vocabulary are audio tokens, so windows fill up quickly
    fc.vocab = cfg.vocab_out - audio_offset
    pc = to_plugin_cfg(hc)
    pc.sine_gen_v1 = True
    gcfg = GLMVoiceConfig(ffn_hidden_size=c.ffn, hidden_size=c.hidden, multi_query_group_num=c.kv_heads, num_attention_heads=c.heads,
                          num_layers=c.layers, padded_vocab_size=cfg.vocab_out, vocab_size=cfg.vocab_out, eos_token_id=[1, 2, 3],
                          audio_offset=audio_offset)

    def serve():
        m = GLMVoiceModel("tiny-glm", S, config=gcfg, sampling=SamplingConfig(greedy=True, max_tokens=70), max_pos=512,
                          codec_weights={"flow": GR.random_glm_flow_weights(fc, seed=5), "hift": HR.random_hift_weights(hc, seed=6)},
                          codec_config={"flow": glm_flow_plugin_cfg(fc), "hift": pc}, codec_seed=4, device=str(dev), max_batch_size=4, page_size=16,
                          max_num_pages=64, max_seq_len=512, max_prefill_tokens=64)
        t = QueueTransport()
        w = ModelWorker(model=m, max_batch_size=4, max_num_pages=64, page_size=16, device=str(dev))
        s = Scheduler(w, max_batch_size=4, transport=t)
        for rid, ids in (("a", [5, 9, 20, 31]), ("b", [6, 8, 30])):
            t.requests.put(encode_request(rid, "", model_kwargs={"prompt_token_ids": ids}))
        s.run_until_idle(3000)
        out, done = {"a": [], "b": []}, {}
        while not t.results.empty():
            rid, kind, body = t.results.get().split(b"|", 2)
            (out[rid.decode()].append(body) if kind == b"AUDIO" else done.__setitem__(rid.decode(), json.loads(body)))
        free = w.empty_pages.qsize()
        m.engine.close(); m.audio_decoder.close()
        return out, done, free

    out, done, free = serve()
    assert free == 64 and set(done) == {"a", "b"} and all(d["status"] == "completed" for d in done.values())
    n_audio = sum(len(v) for v in out.values())
    for chunks in out.values():
        for ch in chunks[:-1] or chunks[:1]:
            assert len(ch) == 2 * 44032
    if n_audio:
        assert max(np.abs(np.frombuffer(ch, np.int16)).max() for v in out.values() for ch in v) > 500
    out2, done2, _ = serve()
    assert out2 == out and done2 == done
