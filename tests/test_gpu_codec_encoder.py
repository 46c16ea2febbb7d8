"""GPU: the speech tokenizer's encoder (vox_codecenc_*) against the oracle and against the reference wiring over transformers'
MimiModel (g16), tiny and full size: pre-quantisation frames within 1e-5 relative RMS, every code equal on the fixtures; on other
clips a code may differ only at a near tie of the nearest-centroid search (relative distance margin < 1e-4)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _plugin(cfg, W, dev, max_seconds=4.0):
    from vox_serve_amd.tokenizer.qwen3_codec_encoder import Qwen3TTSTokenizerV2Encoder, Qwen3TTSTokenizerV2EncoderConfig
    pc = Qwen3TTSTokenizerV2EncoderConfig(
        num_filters=cfg.num_filters, upsampling_ratios=list(reversed(cfg.ratios)), kernel_size=cfg.kernel_size, compress=cfg.compress,
        hidden_size=cfg.hidden_size, num_attention_heads=cfg.num_heads, head_dim=cfg.head_dim, num_hidden_layers=cfg.num_layers,
        intermediate_size=cfg.intermediate_size, rope_theta=cfg.rope_theta, sliding_window=cfg.sliding_window, norm_eps=cfg.norm_eps,
        codebook_size=cfg.codebook_size, codebook_dim=cfg.codebook_dim, num_quantizers=cfg.num_quantizers,
        num_semantic_quantizers=cfg.num_semantic_quantizers, encoder_valid_num_quantizers=cfg.valid_quantizers)
    return Qwen3TTSTokenizerV2Encoder(W, pc, device=dev, max_seconds=max_seconds)


def _check_clip(enc, ref, wav, cfg, strict):
    codes, lat = enc.encode(wav, return_latents=True)
    codes, lat = codes.cpu(), lat.cpu()
    o_lat = ref.latents(wav)
    assert lat.shape == o_lat.shape
    e = float((lat - o_lat).pow(2).mean().sqrt() / o_lat.pow(2).mean().sqrt())
    o_codes, margins = ref.quantize(o_lat, return_margins=True)
    o_codes, margins = o_codes[: codes.shape[0]], margins[: codes.shape[0]]
    assert codes.shape == (-(-wav.numel() // cfg.hop), cfg.valid_quantizers)
    diff = codes != o_codes
    print("latent rel err", e, "codes differing", int(diff.sum()), "of", diff.numel(), "min margin", float(margins.min()))
    assert e < 1e-5
    if strict:
        assert not diff.any()
    else:       # a differing code sits at a near tie, and everything before it in that frame agreed
        for t, q in zip(*np.nonzero(diff.numpy())):
            first = int(np.nonzero(diff[t].numpy())[0][0])
            assert float(margins[t, first]) < 1e-4, (t, q, float(margins[t, first]))
    return codes, lat


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_codec_encoder_matches_oracle_and_reference(golden, tag):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import codec_enc_ref as ER, spk_ref as SR
    dev = torch.device("cuda:0")
    g = golden("g16_codec_encoder")
    cfg = ER.tiny_codec_enc_cfg() if tag == "tiny" else ER.CodecEncCfg()
    seed, n = int(g[f"{tag}_seed"]), int(g[f"{tag}_n"])
    W = ER.random_codec_enc_weights(cfg, seed=seed)
    ref = ER.CodecEncRef(cfg, W)
    enc = _plugin(cfg, W, dev)
    wav = torch.from_numpy(SR.test_audio(seed, n))
    codes, lat = _check_clip(enc, ref, wav, cfg, strict=True)
    want = torch.from_numpy(g[f"{tag}_latents"])
    assert float((lat - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()) < 1e-5
    assert np.array_equal(codes.numpy(), g[f"{tag}_codes"])
    # other lengths through the same object: a whole number of hops, an odd number of 25 Hz frames, a single frame
    for k, n2 in enumerate((cfg.hop * 5, cfg.hop * 4 + cfg.hop // 2 - 3, cfg.hop // 3)):
        _check_clip(enc, ref, torch.from_numpy(SR.test_audio(seed + 1 + k, n2)), cfg, strict=False)
    c1, c2 = enc.encode(wav), enc.encode(wav)
    assert torch.equal(c1, c2)
    with pytest.raises(Exception):
        enc.encode(torch.zeros(enc.max_samples + 1))
    enc.close()
