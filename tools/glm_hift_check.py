import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from oracle import glm_dec_ref as GR, hift_ref as HR
from tests.test_gpu_hift import to_plugin_cfg, rms
from vox_serve_amd.tokenizer.hifigan import HiFTGenerator
g = dict(np.load("tests/golden/g13_glm_decoder.npz"))
for tag in ("tiny", "full"):
    hc = GR.glm_hift_cfg(base_channels=128, f0_channels=64) if tag == "tiny" else GR.glm_hift_cfg()
    W = HR.random_hift_weights(hc, seed=6)
    pc = to_plugin_cfg(hc); pc.sine_gen_v1 = True
    voc = HiFTGenerator(W, pc, device="cuda:0", max_batch=2, max_T=172, seed=47)
    mel = torch.from_numpy(g[f"{tag}_mel"]); B, _, Tm = mel.shape
    ini, nz = HR.make_noise(hc, B, Tm, seed=47, first_stream=8)
    wav, src = voc.forward_chunk(mel, noise=nz, rand_ini=ini)
    wav2, src2 = voc.forward_chunk(mel, stream_base=8 + 2 * torch.arange(B, dtype=torch.int32))
    hr = GR.GlmHiftRef(hc, W)
    with torch.no_grad():
        wo, so = hr.forward_chunk(mel, ini, nz)
    print(tag, "src err", rms(src.cpu().numpy() - so.numpy()), "wav err vs oracle", rms(wav.cpu().numpy() - wo.numpy()), "vs reference", rms(wav.cpu().numpy() - g[f"{tag}_wav"]),
          "device streams vs given", rms((wav - wav2).cpu().numpy()), "signal", rms(g[f"{tag}_wav"]))
    voc.close()
