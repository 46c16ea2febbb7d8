"""GPU: the Qwen3-TTS speaker encoder (log-mel front end + ECAPA-TDNN) through vox_spkenc_* against the oracle and against the
reference modules' outputs (g15), tiny and full size.  Tolerances: log-mel 1e-4 absolute, x-vector 1e-5 relative RMS (fp32
activations over bf16-valued weights; the reference's own bf16 serving run is ~1e-2 from its fp32 run)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _plugin(cfg, W, dev):
    from vox_serve_amd.model.qwen3_tts_speaker import Qwen3TTSSpeakerEncoder, Qwen3TTSSpeakerEncoderConfig
    pc = Qwen3TTSSpeakerEncoderConfig(enc_dim=cfg.enc_dim, mel_dim=cfg.mel_dim, enc_channels=cfg.enc_channels,
                                      enc_kernel_sizes=cfg.enc_kernel_sizes, enc_dilations=cfg.enc_dilations,
                                      enc_res2net_scale=cfg.enc_res2net_scale, enc_se_channels=cfg.enc_se_channels,
                                      enc_attention_channels=cfg.enc_attention_channels)
    return Qwen3TTSSpeakerEncoder({k: torch.from_numpy(v) for k, v in W.items()}, pc, device=dev, max_seconds=4.0)


@pytest.mark.parametrize("tag", ["tiny", "full"])
def test_speaker_encoder_matches_oracle_and_reference(golden, tag):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import spk_ref as SR
    dev = torch.device("cuda:0")
    g = golden("g15_speaker_encoder")
    cfg = SR.tiny_spk_cfg() if tag == "tiny" else SR.SpkCfg()
    seed, n = int(g[f"{tag}_seed"]), int(g[f"{tag}_n"])
    W = SR.random_spk_weights(cfg, seed=seed)
    audio = SR.test_audio(seed, n)
    enc = _plugin(cfg, W, dev)
    emb, mel = enc(torch.from_numpy(audio), return_mel=True)
    emb, mel = emb.cpu().numpy(), mel.cpu().numpy()
    o_mel = SR.mel_spectrogram(audio, cfg)
    assert mel.shape == o_mel.shape == g[f"{tag}_mel"].shape
    assert np.abs(mel - o_mel).max() < 1e-4 and np.abs(mel - g[f"{tag}_mel"]).max() < 1e-4
    o_emb = SR.SpkRef(cfg, W).forward(o_mel)
    rms = np.sqrt((o_emb ** 2).mean())
    e_or = np.sqrt(((emb - o_emb) ** 2).mean()) / rms
    e_ref = np.sqrt(((emb - g[f"{tag}_emb"]) ** 2).mean()) / rms
    print(tag, "x-vector rel err vs oracle", e_or, "vs reference", e_ref)
    assert e_or < 1e-5 and e_ref < 1e-5
    # a second clip of another length through the same object; determinism
    a2 = SR.test_audio(seed + 1, n - 1234)
    e1, e2 = enc(torch.from_numpy(a2)).cpu().numpy(), enc(torch.from_numpy(a2)).cpu().numpy()
    assert np.array_equal(e1, e2)
    o2 = SR.SpkRef(cfg, W).embed(a2)
    assert np.sqrt(((e1 - o2) ** 2).mean()) / np.sqrt((o2 ** 2).mean()) < 1e-5
    with pytest.raises(Exception):
        enc(torch.zeros(200))           # shorter than the STFT's reflect padding
    enc.close()
