"""GPU: the CosyVoice2 detokenizer — flow (tokens -> mel) and the whole decode_chunk (tokens -> waveform) — through the C ABI against the
CPU oracle (oracle/flow_ref.py + oracle/hift_ref.py, pinned to the reference CosyVoice2Decoder by tests/test_oracle_goldens.py::test_flow_*)
and against the reference's own output (tests/golden/g12_flow.npz).
Tolerances (floating point path): mel RMS error <= 1e-4 (mel RMS ~ 1.2); waveform RMS error <= 1e-4 against the fp32 computation for the
vocoder given the same mels (tests/test_gpu_hift.py) and <= 2e-4 end to end — the vocoder's harmonic source multiplies an f0 error by
2 pi * 480 * frames, so a 1e-6 mel difference is a 3e-5 waveform difference (the oracle itself sits 2e-5 from the reference)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rms(x):
    return float(np.sqrt(np.mean(np.asarray(x, np.float64) ** 2)))


def flow_plugin_cfg(fc):
    from vox_serve_amd.tokenizer.cosyvoice_flow import FlowConfig
    return FlowConfig(vocab_size=fc.vocab, dim=fc.dim, mel=fc.mel, spk_embed_dim=fc.spk_dim, enc_layers=fc.enc_layers, up_layers=fc.up_layers,
                      enc_heads=fc.enc_heads, enc_ffn=fc.enc_ffn, pre_lookahead_len=fc.pre_lookahead, est_channels=fc.est_ch,
                      est_heads=fc.est_heads, est_head_dim=fc.est_head_dim, est_blocks=fc.est_blocks, est_mid_blocks=fc.est_mid,
                      n_timesteps=fc.n_steps, inference_cfg_rate=fc.cfg_rate, max_cache_len=fc.max_cache, prefix_len=fc.prefix)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _setup(tag, golden):
    from oracle import flow_ref as FR, hift_ref as HR
    g = golden("g12_flow")
    fc, hc = (FR.tiny_flow_cfg(), HR.HiftCfg(base_channels=256, f0_channels=64)) if tag == "tiny" else (FR.FlowCfg(), HR.HiftCfg())
    ptok, pfeat, spk = (torch.from_numpy(g[f"{tag}_{k}"]) for k in ("prompt_token", "prompt_feat", "spk"))
    tok = torch.from_numpy(g[f"{tag}_token"]).long()
    return g, fc, hc, FR.random_flow_weights(fc, seed=3), HR.random_hift_weights(hc, seed=2), ptok, pfeat, spk, tok


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_flow_mels_match_oracle_and_reference(dev, golden, tag):
    """init_cache on the prompt (native), then one 28-token chunk of two requests against the static caches: prompt mels, encoder
    output and chunk mels vs the oracle; chunk mels vs the reference module's; the seeded device noise == the same noise handed in."""
    from oracle import flow_ref as FR
    from vox_serve_amd.tokenizer.cosyvoice_flow import CosyVoice2Flow
    g, fc, hc, Wf, Wh, ptok, pfeat, spk, tok = _setup(tag, golden)
    seed = int(g["noise_seed"])
    Np, (B, T) = ptok.shape[1], tok.shape
    fr = FR.FlowRef(fc, Wf)
    z0, z1 = FR.cfm_noise(seed, 0, fc.mel, 2 * (Np + 3)), FR.cfm_noise(seed, 1, fc.mel, 2 * T)
    flow = CosyVoice2Flow(Wf, flow_plugin_cfg(fc), device=dev, max_batch=2, max_T=28, max_prompt_T=48, seed=seed)
    with pytest.raises(Exception):
        flow.forward_chunk(tok)                                    # no prompt yet: fails loudly
    pm = flow.set_prompt(ptok, pfeat, spk, noise=z0).cpu().numpy()
    with torch.no_grad():
        pm_o, cache = fr.init_cache(ptok.long(), pfeat, spk, z0)
        mel_o, _ = fr.flow_chunk(tok, torch.zeros(1, 0, fc.mel), spk, z1, cache)
    assert list(g[f"{tag}_cache_lens"]) == [cache["enc"].shape[3], cache["up"].shape[3], cache["att"].shape[5]]
    mel = flow.forward_chunk(tok, noise=z1).cpu().numpy()
    mel_s = flow.forward_chunk(tok, noise_stream=1).cpu().numpy()
    three = flow.forward_chunk(torch.cat([tok, tok[:1]]), noise=z1).cpu().numpy()      # 3 requests through a max_batch-2 engine
    assert rms(pm - pm_o.numpy()) < 1e-4 and rms(mel - mel_o.numpy()) < 1e-4, (rms(pm - pm_o.numpy()), rms(mel - mel_o.numpy()))
    assert rms(mel - g[f"{tag}_mel"]) < 1e-4 and rms(g[f"{tag}_mel"]) > 0.5
    assert rms(mel_s - mel) < 1e-5
    assert np.array_equal(three[:2], mel) and np.array_equal(three[2], mel[0])            # a request's mels do not depend on its batch
    flow.close()


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_cosyvoice2_decode_chunk_matches_reference_audio(dev, golden, tag):
    """CosyVoice2Decoder.init_cache + decode_chunk (shared prompt cache mode) end to end: tokens -> 24000 samples per request."""
    from oracle import flow_ref as FR, hift_ref as HR
    from tests.test_gpu_hift import to_plugin_cfg
    from vox_serve_amd.tokenizer.cosyvoice2 import CosyVoice2Decoder
    g, fc, hc, Wf, Wh, ptok, pfeat, spk, tok = _setup(tag, golden)
    seed = int(g["noise_seed"])
    Np, (B, T) = ptok.shape[1], tok.shape
    dec = CosyVoice2Decoder(Wf, Wh, device=dev, flow_config=flow_plugin_cfg(fc), hift_config=to_plugin_cfg(hc), max_batch=2, max_prompt_tokens=48,
                            seed=seed)
    ref = {"prompt_speech_token": ptok, "prompt_feat": pfeat, "embedding": spk}
    cache = dec.init_cache(ref, noise=FR.cfm_noise(seed, 0, fc.mel, 2 * (Np + 3)))
    assert cache.prompt_tokens == Np and tuple(cache.prompt_mels.shape) == (1, fc.mel, 2 * (Np + 3))
    ini, nz = HR.make_noise(hc, B, 2 * T, seed=seed, first_stream=16)
    audio, _ = dec.decode_chunk(tok, T, cache, ref_dict=ref, flow_noise=FR.cfm_noise(seed, 1, fc.mel, 2 * T), hift_noise=nz)
    audio_s, _ = dec.decode_chunk(tok, T, cache, ref_dict=ref, flow_noise_stream=1, hift_stream_base=16 + 2 * torch.arange(B, dtype=torch.int32))
    audio, audio_s = audio.cpu().numpy(), audio_s.cpu().numpy()
    want = g[f"{tag}_audio"]
    assert audio.shape == want.shape == (B, 24000) and rms(want) > 0.05
    assert rms(audio - want) < 2e-4, rms(audio - want)
    assert rms(audio_s - want) < 2e-4, rms(audio_s - want)
    assert np.abs(audio[:, :8]).max() < 0.02                       # faded in from silence (cosyvoice2.py:1040-1046)
    dec.close()


def test_cosyvoice2_per_request_evolving_caches(dev, golden):
    """use_detokenizer_cache=True (CosyVoice2Decoder with shared_prompt_cache_mode=False, cosyvoice2.py:1010-1083): one request, three
    consecutive 28-token chunks — the caches grow (40 / 80 / 80 rows), then slide (64 / 128 / 128: the third chunk runs against
    truncated caches held as a ring), the fade-in blends against the previous chunk's tail.  HIP vs the reference's own output (g17)
    and vs the oracle (decode_chunk_evolving) chunk by chunk; then two requests that started one chunk apart in ONE call (different
    cache states: decoded in groups) equal the same requests decoded alone."""
    from oracle import flow_ref as FR, hift_ref as HR
    from tests.test_gpu_hift import to_plugin_cfg
    from vox_serve_amd.tokenizer.base import DecoderCache
    from vox_serve_amd.tokenizer.cosyvoice2 import CosyVoice2Decoder
    g = golden("g17_flow_evolving")
    fc, hc = FR.tiny_flow_cfg(), HR.HiftCfg(base_channels=256, f0_channels=64)
    Wf, Wh = FR.random_flow_weights(fc, seed=3), HR.random_hift_weights(hc, seed=2)
    seed, T, K = int(g["noise_seed"]), g["tokens"].shape[2], g["tokens"].shape[0]
    ptok, pfeat, spk = torch.from_numpy(g["prompt_token"]).long(), torch.from_numpy(g["prompt_feat"]), torch.from_numpy(g["spk"])
    ref = {"prompt_speech_token": ptok, "prompt_feat": pfeat, "embedding": spk}
    z0 = FR.cfm_noise(seed, 0, fc.mel, 2 * (ptok.shape[1] + 3))

    def make():
        d = CosyVoice2Decoder(Wf, Wh, device=dev, flow_config=flow_plugin_cfg(fc), hift_config=to_plugin_cfg(hc), max_batch=2, max_prompt_tokens=48,
                              seed=seed, shared_prompt_cache_mode=False, max_slots=4)
        d.init_cache(ref, noise=z0)
        return d
    dec = make()
    fr, hr = FR.FlowRef(fc, Wf), HR.HiftRef(hc, Wh)
    with torch.no_grad():
        _, ocache = fr.init_cache(ptok, pfeat, spk, z0)
    ospeech = torch.zeros(1, 6 * hc.upsample_scale)
    cache = dec.new_request_cache(1)
    assert dec.flow.slot_lens(int(cache.slot[0])) == [ptok.shape[1] + 3, 2 * (ptok.shape[1] + 3), 2 * (ptok.shape[1] + 3)]
    audios = []
    for k in range(K):
        tok = torch.from_numpy(g["tokens"][k]).long()
        ini, nz = HR.make_noise(hc, 1, 2 * T, seed=seed, first_stream=16 + 2 * k)
        zk = FR.cfm_noise(seed, 1 + k, fc.mel, 2 * T)
        audio, cache = dec.decode_chunk(tok, T, cache, ref_dict=ref, flow_noise=zk, hift_noise=nz)
        with torch.no_grad():
            oa, _, ocache, ospeech = FR.decode_chunk_evolving(fr, hr, tok, spk, ocache, ospeech, zk, ini, nz)
        a = audio.cpu().numpy()
        audios.append(a)
        assert a.shape == g[f"audio_{k}"].shape
        assert rms(a - g[f"audio_{k}"]) < 2e-4, (k, rms(a - g[f"audio_{k}"]))            # vs the reference
        assert rms(a - oa.numpy()) < 2e-4, (k, rms(a - oa.numpy()))                       # vs the oracle
        assert dec.flow.slot_lens(int(cache.slot[0])) == g["cache_lens"][k].tolist(), k
        assert rms(cache.speech_cache.cpu().numpy() - g[f"speech_cache_{k}"]) < 4e-4, k
    assert rms(audios[1] - audios[0]) > 1e-2
    # two requests one chunk apart, decoded together: request A at its chunk k + 1 beside request B at its chunk k
    ca, cb = dec.new_request_cache(1), dec.new_request_cache(1)
    toks = [torch.from_numpy(g["tokens"][k]).long() for k in range(K)]
    nzs = [HR.make_noise(hc, 1, 2 * T, seed=seed, first_stream=16 + 2 * k)[1] for k in range(K)]
    a0, ca = dec.decode_chunk(toks[0], T, ca, ref_dict=ref, flow_noise=FR.cfm_noise(seed, 1, fc.mel, 2 * T), hift_noise=nzs[0])
    assert rms(a0.cpu().numpy() - audios[0]) < 1e-6                                      # a fresh slot reproduces the first run
    both = DecoderCache.cat([ca, cb])
    zk = FR.cfm_noise(seed, 2, fc.mel, 2 * T)
    out, both = dec.decode_chunk(torch.cat([toks[1], toks[1]]), T, both, ref_dict=ref, flow_noise=zk, hift_noise=torch.cat([nzs[1], nzs[1]]))
    assert rms(out[0].cpu().numpy() - audios[1][0]) < 1e-6                               # A: its second chunk, as alone
    assert dec.flow.slot_lens(int(ca.slot[0])) == g["cache_lens"][1].tolist() and dec.flow.slot_lens(int(cb.slot[0])) == g["cache_lens"][0].tolist()
    assert rms(out[1].cpu().numpy() - audios[1][0]) > 1e-3                               # B saw the same tokens against a shorter history
    dec.release_cache(ca), dec.release_cache(cb), dec.release_cache(cache)
    assert sorted(dec._free_slots) == [0, 1, 2, 3]
    dec.close()
