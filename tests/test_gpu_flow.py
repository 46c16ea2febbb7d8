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
