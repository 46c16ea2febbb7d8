"""GPU: the SNAC decoder (libvoxhip vox_snac_* through the C ABI) against the CPU oracle and the reference-module fixtures
(g10), tiny and snac_24khz size: waveform RMS error < 1e-4 with the NoiseBlock noise injected on both sides; the device's
own seeded Philox noise stream against the oracle's restatement of it; Orpheus' token layout and output window."""
import numpy as np
import pytest
import torch

from oracle import snac_ref as SR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda")


def build(dev, cfg, seed=1, **kw):
    from vox_serve_amd.tokenizer.snac import SNACConfig, SNACDecoder
    pc = SNACConfig(latent_dim=cfg.latent_dim, decoder_dim=cfg.decoder_dim, decoder_rates=list(cfg.rates), codebook_size=cfg.codebook_size,
                    codebook_dim=cfg.codebook_dim, vq_strides=list(cfg.vq_strides))
    W = SR.random_snac_weights(cfg, seed=seed)
    return SNACDecoder(W, pc, device=dev, **kw), SR.SnacRef(cfg, W)


def rms(a):
    return float(np.sqrt(np.mean(np.asarray(a, np.float64) ** 2)))


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_snac_decode_vs_oracle_and_reference_fixture(dev, golden, tag):
    g = golden("g10_snac")
    cfg = SR.tiny_snac_cfg() if tag == "tiny" else SR.SnacCfg()
    dec, ref = build(dev, cfg, max_batch=2)
    codes = [torch.from_numpy(g[f"{tag}_codes{i}"].astype(np.int64)) for i in range(3)]
    noise = SR.make_noise(cfg, 2, 16, seed=int(g["noise_seed"]))
    wav = dec.decode(codes, noise).cpu().numpy()
    want_oracle, want_ref = ref.decode(codes, noise).numpy(), g[f"{tag}_wav"]
    assert wav.shape == want_ref.shape == (2, 1, 16 * cfg.hop)
    assert rms(wav - want_oracle) < 1e-4 and rms(wav - want_ref) < 1e-4, (rms(wav - want_oracle), rms(wav - want_ref), rms(want_ref))
    assert np.abs(wav - want_ref).max() < 1e-3
    dec.close()


@pytest.mark.parametrize("tag", ["dense_attn", "dw_attn", "dense"])
def test_snac_dense_conv_and_local_attention_variants(dev, golden, tag):
    """The module's other forms (snac.py:20-90 LocalMHA, :119-176 dense k7 convs; the 32 / 44 kHz checkpoints' structure), tiny size,
    against the oracle and the reference module's output (g20); a frame count that is not a multiple of the window is refused as in
    the reference (its `T // window_size` reshape fails)."""
    import dataclasses
    from vox_serve_amd.tokenizer.snac import SNACConfig, SNACDecoder
    g = golden("g20_snac_variants")
    base = SR.tiny_snac_cfg()
    cfg = {"dense_attn": dataclasses.replace(base, depthwise=False, attn_window_size=4),
           "dw_attn": dataclasses.replace(base, depthwise=True, attn_window_size=4),
           "dense": dataclasses.replace(base, depthwise=False, attn_window_size=None)}[tag]
    W = SR.random_snac_weights(cfg, seed=2, final_gain=0.3)
    pc = SNACConfig(latent_dim=cfg.latent_dim, decoder_dim=cfg.decoder_dim, decoder_rates=list(cfg.rates), codebook_size=cfg.codebook_size,
                    codebook_dim=cfg.codebook_dim, vq_strides=list(cfg.vq_strides), depthwise=cfg.depthwise, attn_window_size=cfg.attn_window_size)
    dec, ref = SNACDecoder(W, pc, device=dev, max_batch=2, max_T=24), SR.SnacRef(cfg, W)
    codes = [torch.from_numpy(g[f"{tag}_codes{i}"].astype(np.int64)) for i in range(3)]
    noise = SR.make_noise(cfg, 2, 16, seed=int(g["noise_seed"]))
    wav = dec.decode(codes, noise).cpu().numpy()
    want_oracle, want_ref = ref.decode(codes, noise).numpy(), g[f"{tag}_wav"]
    assert wav.shape == want_ref.shape == (2, 1, 16 * cfg.hop)
    assert rms(wav - want_oracle) < 1e-4 and rms(wav - want_ref) < 1e-4, (rms(wav - want_oracle), rms(wav - want_ref), rms(want_ref))
    assert np.abs(wav - want_ref).max() < 1e-3
    if cfg.attn_window_size is not None:
        odd = [c[:, : 12 // s] for c, s in zip(codes, cfg.vq_strides)]          # 12 frames: the window is 4 -> fine; 20 would be too
        assert dec.decode(odd, SR.make_noise(cfg, 2, 12, seed=1)).shape == (2, 1, 12 * cfg.hop)
        bad_cfg = dataclasses.replace(pc, attn_window_size=8)
        bad = SNACDecoder(W, bad_cfg, device=dev, max_batch=2, max_T=24)
        with pytest.raises(Exception, match="multiple of the attention window"):
            bad.decode(odd, SR.make_noise(cfg, 2, 12, seed=1))                 # 12 % 8 != 0
        bad.close()
    dec.close()


def test_snac_device_noise_stream_and_windowing(dev):
    """No noise handed in: the kernel draws it (Philox4x32-10 + Box-Muller, stream = stream_base[b] + stage).  The oracle run
    with philox_noise of the same seed / streams reproduces the waveform; an output window equals the slice of the full decode;
    a batch larger than max_batch is split without changing results."""
    cfg = SR.SnacCfg()
    dec, ref = build(dev, cfg, max_batch=2, seed=3)
    dec.seed = 1234
    g = torch.Generator().manual_seed(9)
    B, T = 3, 16
    codes = [torch.randint(0, cfg.codebook_size, (B, T // s), generator=g) for s in cfg.vq_strides]
    base = torch.tensor([40, 7, 1000], dtype=torch.int32)
    full = dec.decode(codes, stream_base=base).cpu().numpy()
    ns = len(cfg.rates)
    noise = [torch.from_numpy(np.stack([SR.philox_noise(1234, int(base[b]) + i, L) for b in range(B)])[:, None, :])
             for i, L in enumerate(SR.stage_lengths(cfg, T))]
    want = ref.decode(codes, noise).numpy()
    assert rms(full - want) < 1e-4, rms(full - want)
    win = dec.decode(codes, stream_base=base, out_off=2048, out_len=2048).cpu().numpy()
    assert np.array_equal(win, full[:, :, 2048:4096])
    again = dec.decode(codes, stream_base=base).cpu().numpy()
    assert np.array_equal(again, full)                         # deterministic
    other = dec.decode(codes, stream_base=base + 1).cpu().numpy()
    assert rms(other - full) > 1e-3                            # the noise branch is live
    dec.close()


def test_orpheus_postprocess_fixture(dev, golden):
    """OrpheusModel.postprocess (model/orpheus.py:479-507) on raw LM ids vs the reference's output (g10)."""
    from vox_serve_amd.model.orpheus import orpheus_codes
    g = golden("g10_snac")
    cfg = SR.SnacCfg()
    dec, _ = build(dev, cfg, max_batch=2)
    tok = torch.from_numpy(g["orpheus_tokens"].astype(np.int64))
    noise = SR.make_noise(cfg, 2, 16, seed=int(g["noise_seed"]))
    audio = dec.decode(orpheus_codes(tok, cfg.codebook_size), noise, out_off=2048, out_len=2048).cpu().numpy()
    assert audio.shape == (2, 1, 2048)
    assert rms(audio - g["orpheus_audio"]) < 1e-4
    dec.close()
