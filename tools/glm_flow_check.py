"""GPU: GLM-4-Voice detokenizer error margins vs the oracle / reference fixture (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import glm_dec_ref as GR, hift_ref as HR
from tests.test_gpu_hift import to_plugin_cfg, rms
from vox_serve_amd.tokenizer.glm import GLMAudioDecoder, GLMFlowConfig
g = dict(np.load("tests/golden/g13_glm_decoder.npz"))
def fcfg(c):
    return GLMFlowConfig(vocab_size=c.vocab, dim=c.dim, mel=c.mel, spk_embed_dim=c.spk_dim, enc_layers=c.enc_layers, enc_heads=c.enc_heads, enc_ffn=c.enc_ffn,
                         block_size=c.block_size, est_channels=c.est_ch, est_heads=c.est_heads, est_head_dim=c.est_head_dim, est_blocks=c.est_blocks,
                         est_mid_blocks=c.est_mid, n_timesteps=c.n_steps, inference_cfg_rate=c.cfg_rate)
for tag in (sys.argv[1:] or ["tiny", "full"]):
    fc, hc = (GR.tiny_glm_flow_cfg(), GR.glm_hift_cfg(base_channels=128, f0_channels=64)) if tag == "tiny" else (GR.GlmFlowCfg(), GR.glm_hift_cfg())
    Wf, Wh = GR.random_glm_flow_weights(fc, seed=5), HR.random_hift_weights(hc, seed=6)
    pc = to_plugin_cfg(hc); pc.sine_gen_v1 = True
    dec = GLMAudioDecoder(Wf, Wh, device="cuda:0", flow_config=fcfg(fc), hift_config=pc, max_batch=2, seed=47)
    tok = torch.from_numpy(g[f"{tag}_token"]).long(); B, T = tok.shape; Tm = 172
    z = GR.glm_cfm_noise(47, 0, B, fc.mel, Tm)
    ini, nz = HR.make_noise(hc, B, Tm, seed=47, first_stream=8)
    mel = dec.flow.inference(tok, noise=z).cpu().numpy()
    mel_s = dec.flow.inference(tok, first_stream=0).cpu().numpy()
    wav = dec.forward(tok, flow_noise=z, hift_noise=nz, hift_rand_ini=ini).cpu().numpy()
    print(tag, "mel err vs reference", rms(mel - g[f"{tag}_mel"]), "/", rms(g[f"{tag}_mel"]), "device noise vs given", rms(mel - mel_s),
          "wav err vs reference", rms(wav - g[f"{tag}_wav"]), "/", rms(g[f"{tag}_wav"]))
    if tag == "full":
        for _ in range(2): dec.forward(tok[:1])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): dec.forward(tok[:1])
        torch.cuda.synchronize()
        print(f"  B=1, 25 tokens -> 44032 samples (2.0 s): {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per call (eager)")
    dec.close()
