"""GPU: HiFT vocoder error margins vs the oracle / reference fixture and chunk timing (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import hift_ref as HR
from tests.test_gpu_hift import to_plugin_cfg, rms
from vox_serve_amd.tokenizer.hifigan import HiFTGenerator
dev = torch.device("cuda:0")
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "g11_hift.npz")))
for tag, cfg in (("tiny", HR.tiny_hift_cfg()), ("full", HR.HiftCfg())):
    W = HR.random_hift_weights(cfg, seed=2)
    mel = torch.from_numpy(g[f"{tag}_mel"]); B, _, T = mel.shape
    ini, nz = HR.make_noise(cfg, B, T, seed=91)
    voc = HiFTGenerator(W, to_plugin_cfg(cfg), device=dev, max_batch=8, max_T=64, seed=91)
    wav, src = voc.forward_chunk(mel, noise=nz)
    wo, so = HR.HiftRef(cfg, W).forward_chunk(mel, ini, nz)
    print(tag, "src err", rms(src.cpu().numpy() - so.numpy()), "wav err vs oracle", rms(wav.cpu().numpy() - wo.numpy()),
          "vs reference", rms(wav.cpu().numpy() - g[f"{tag}_wav"]), "signal", rms(g[f"{tag}_wav"]))
    if tag == "full":
        for Bt, Tt in ((1, 30), (8, 30), (1, 60)):
            m = torch.randn(Bt, cfg.in_channels, Tt)
            for _ in range(3): voc.forward_chunk(m)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): voc.forward_chunk(m)
            torch.cuda.synchronize()
            print(f"  B={Bt} T={Tt} mel frames ({Tt * 480 / 24000:.2f} s audio): {(time.perf_counter() - t0) * 100:.2f} ms per chunk (eager launches)")
    voc.close()
