"""GPU: the HiFT vocoder (mel -> waveform of the CosyVoice2 / GLM-4-Voice detokenizers) through the C ABI against the CPU oracle
(oracle/hift_ref.py, pinned to the reference HiFTGenerator by tests/test_oracle_goldens.py::test_hift_*) and against the
reference module's own output (tests/golden/g11_hift.npz).
Tolerance (floating point path): waveform RMS error <= 1e-4 against the fp32 computation (signal RMS ~ 0.1)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rms(x):
    return float(np.sqrt(np.mean(np.asarray(x, np.float64) ** 2)))


def to_plugin_cfg(cfg):
    from vox_serve_amd.tokenizer.hifigan import HiFTConfig
    return HiFTConfig(in_channels=cfg.in_channels, base_channels=cfg.base_channels, nb_harmonics=cfg.nb_harmonics,
                      sampling_rate=cfg.sampling_rate, nsf_alpha=cfg.nsf_alpha, nsf_sigma=cfg.nsf_sigma,
                      nsf_voiced_threshold=cfg.voiced_threshold, upsample_rates=list(cfg.upsample_rates),
                      upsample_kernel_sizes=list(cfg.upsample_kernel_sizes), istft_n_fft=cfg.n_fft, istft_hop_len=cfg.hop_len,
                      resblock_kernel_sizes=list(cfg.resblock_kernel_sizes), resblock_dilation_sizes=list(cfg.resblock_dilations),
                      source_resblock_kernel_sizes=list(cfg.source_resblock_kernel_sizes), lrelu_slope=cfg.lrelu_slope,
                      audio_limit=cfg.audio_limit, f0_channels=cfg.f0_channels)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_hift_matches_oracle_and_reference_fixture(dev, golden, tag):
    from oracle import hift_ref as HR
    from vox_serve_amd.tokenizer.hifigan import HiFTGenerator
    g = golden("g11_hift")
    cfg = HR.tiny_hift_cfg() if tag == "tiny" else HR.HiftCfg()
    W = HR.random_hift_weights(cfg, seed=2)
    mel = torch.from_numpy(g[f"{tag}_mel"])
    B, _, T = mel.shape
    ini, nz = HR.make_noise(cfg, B, T, seed=int(g["noise_seed"]))
    voc = HiFTGenerator(W, to_plugin_cfg(cfg), device=dev, max_batch=4, max_T=16, seed=int(g["noise_seed"]))
    # (a) the noise handed in, (b) the seeded device stream: the same tensors, so the same audio
    wav_a, src_a = voc.forward_chunk(mel, noise=nz)
    wav_b, src_b = voc.forward_chunk(mel, stream_base=torch.arange(B, dtype=torch.int32) * 2)
    wav_a, src_a, wav_b, src_b = (t.cpu().numpy() for t in (wav_a, src_a, wav_b, src_b))
    assert wav_a.shape == (B, T * cfg.upsample_scale) and src_a.shape == (B, 1, T * cfg.upsample_scale)
    ref = HR.HiftRef(cfg, W)
    wav_o, src_o = ref.forward_chunk(mel, ini, nz)
    sig = rms(g[f"{tag}_wav"])
    assert sig > 0.05
    for name, wav, src in (("given", wav_a, src_a), ("device stream", wav_b, src_b)):
        assert rms(src - src_o.numpy()) < 1e-4, (name, rms(src - src_o.numpy()))
        assert rms(wav - wav_o.numpy()) < 1e-4, (name, rms(wav - wav_o.numpy()), sig)
        assert rms(src - g[f"{tag}_source"]) < 1e-4 and rms(wav - g[f"{tag}_wav"]) < 1e-4, (name, rms(wav - g[f"{tag}_wav"]))
    # the initial phases SineGen2 draws never reach the output (see include/voxhip.h): any other draw gives the oracle's same source
    wav_o2, src_o2 = ref.forward_chunk(mel, torch.rand_like(ini), nz)
    assert torch.equal(src_o2, src_o)
    voc.close()


def test_hift_batching_and_chunk_streams(dev):
    """A request's audio does not depend on the batch it shares a call with (same stream base), and the default stream bases advance
    from call to call (two chunks never reuse noise)."""
    from oracle import hift_ref as HR
    from vox_serve_amd.tokenizer.hifigan import HiFTGenerator
    cfg = HR.tiny_hift_cfg()
    W = HR.random_hift_weights(cfg, seed=4)
    voc = HiFTGenerator(W, to_plugin_cfg(cfg), device=dev, max_batch=2, max_T=12, seed=7)
    g = torch.Generator().manual_seed(1)
    mel = (0.8 * torch.randn(5, cfg.in_channels, 9, generator=g))
    sb = torch.tensor([10, 12, 14, 16, 18], dtype=torch.int32)
    all_wav, _ = voc.forward_chunk(mel, stream_base=sb)                      # five requests through a max_batch-2 engine: three calls
    one_wav, _ = voc.forward_chunk(mel[3:4], stream_base=sb[3:4])
    assert torch.equal(all_wav[3], one_wav[0])
    w1, _ = voc.forward_chunk(mel[:2])
    w2, _ = voc.forward_chunk(mel[:2])
    assert not torch.equal(w1, w2) and rms((w1 - w2).cpu().numpy()) < 0.05      # different noise, same speech
    with pytest.raises(Exception):
        voc.forward_chunk(torch.zeros(1, cfg.in_channels, 13))                   # T > max_T fails loudly
    voc.close()
