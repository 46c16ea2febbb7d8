"""GPU: CosyVoice2 flow (tokens -> mel) error margins vs the oracle / reference fixture and chunk timing (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import flow_ref as FR
from vox_serve_amd.tokenizer.cosyvoice_flow import CosyVoice2Flow, FlowConfig
dev = torch.device("cuda:0")
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "g12_flow.npz")))
rms = lambda x: float(np.sqrt(np.mean(np.asarray(x, np.float64) ** 2)))
def plugin_cfg(fc):
    return FlowConfig(vocab_size=fc.vocab, dim=fc.dim, mel=fc.mel, spk_embed_dim=fc.spk_dim, enc_layers=fc.enc_layers, up_layers=fc.up_layers,
                      enc_heads=fc.enc_heads, enc_ffn=fc.enc_ffn, pre_lookahead_len=fc.pre_lookahead, est_channels=fc.est_ch,
                      est_heads=fc.est_heads, est_head_dim=fc.est_head_dim, est_blocks=fc.est_blocks, est_mid_blocks=fc.est_mid,
                      n_timesteps=fc.n_steps, inference_cfg_rate=fc.cfg_rate, max_cache_len=fc.max_cache, prefix_len=fc.prefix)
for tag in (sys.argv[1:] or ["tiny", "full"]):
    fc = FR.tiny_flow_cfg() if tag == "tiny" else FR.FlowCfg()
    W = FR.random_flow_weights(fc, seed=3)
    fr = FR.FlowRef(fc, W)
    ptok, pfeat, spk = (torch.from_numpy(g[f"{tag}_{k}"]) for k in ("prompt_token", "prompt_feat", "spk"))
    tok = torch.from_numpy(g[f"{tag}_token"]).long()
    Np, (B, T) = ptok.shape[1], tok.shape
    flow = CosyVoice2Flow(W, plugin_cfg(fc), device=dev, max_batch=4, max_T=32, max_prompt_T=64, seed=33)
    z0, z1 = FR.cfm_noise(33, 0, fc.mel, 2 * (Np + 3)), FR.cfm_noise(33, 1, fc.mel, 2 * T)
    pm = flow.set_prompt(ptok, pfeat, spk, noise=z0).cpu()
    with torch.no_grad():
        pm_o, cache = fr.init_cache(ptok.long(), pfeat, spk, z0)
        mel_o, _ = fr.flow_chunk(tok, torch.zeros(1, 0, fc.mel), spk, z1, cache)
        emb = fr._lin(torch.nn.functional.normalize(spk, dim=1), "spk_embed_affine_layer")
        h_o, _, _ = fr.encoder_chunk(torch.nn.functional.embedding(tok, fr.W["input_embedding.weight"]), cache["enc"], cache["up"])
        mu_o = fr._lin(h_o, "encoder_proj")
    mel, mu = flow.forward_chunk(tok, noise=z1, return_mu=True)
    mel2 = flow.forward_chunk(tok, noise_stream=1)
    print(tag, "prompt mel err", rms(pm.numpy() - pm_o.numpy()), "/", rms(pm_o.numpy()))
    print(tag, "mu err", rms(mu.cpu().numpy() - mu_o.numpy()), "/", rms(mu_o.numpy()))
    print(tag, "chunk mel err vs oracle", rms(mel.cpu().numpy() - mel_o.numpy()), "vs reference", rms(mel.cpu().numpy() - g[f"{tag}_mel"]), "/", rms(g[f"{tag}_mel"]),
          "device-stream noise vs given", rms((mel - mel2).cpu().numpy()))
    if tag == "full":
        for Bt in (1, 8):
            tk = torch.randint(0, fc.vocab, (min(Bt, 4), 28))
            for _ in range(2): flow.forward_chunk(tk)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3): flow.forward_chunk(tk)
            torch.cuda.synchronize()
            print(f"  B={min(Bt,4)} T=28 tokens (1.12 s audio): {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per chunk (eager launches)")
    flow.close()
