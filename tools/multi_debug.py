"""Development aid: one-request frames at a given kv length, ids / hidden / status row dumped to a file for comparison between builds or
environment settings (VOX_TALKER_MULTI=0 / 1).  python tools/multi_debug.py <out.npz> [kv] [frames]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vox_serve_amd.engine import Qwen3Cfg, Qwen3Engine
from vox_serve_amd.synth import synth_qwen3_weights
out = sys.argv[1]; kv0 = int(sys.argv[2]) if len(sys.argv) > 2 else 200; frames = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda")
cfg = Qwen3Cfg()
eng = Qwen3Engine(cfg, synth_qwen3_weights(cfg, dev, seed=0), max_batch=1, page_size=128, max_pages=64, max_seq_len=2304, max_prefill_rows=128)
eng.keep_hidden = True
g = torch.Generator(device="cpu").manual_seed(1)
eng.kv[:, :3].copy_((torch.randn(eng.kv[:, :3].shape, generator=g) * 0.5).to(eng.kv.dtype))
sc = eng.sampling_cfg(greedy=True)
eng.input_ids.zero_(); eng.input_ids[:, -1] = cfg.tts_pad_id
res = {}
for f in range(frames):
    kv = kv0 + f
    pages = list(range((kv + 127) // 128))
    eng.upload_plan(pos=[kv], kvlen=[kv], page=[pages[-1]], slot=[(kv - 1) % 128], indptr=[0, len(pages)], indices=pages)
    eng.frame(1, kv, sc, use_graph=False, feedback=False)
    torch.cuda.synchronize()
    res[f"ids{f}"] = eng.out_ids[:1].cpu().numpy()
    res[f"hid{f}"] = eng.out_hidden[:1].float().cpu().numpy()
    res[f"status{f}"] = eng.status_row.cpu().numpy() if hasattr(eng, "status_row") else np.zeros(1)
    print(f, "status", res[f"status{f}"][:4], "ids", res[f"ids{f}"][0][:6], "hid", res[f"hid{f}"][0][:4])
np.savez(out, **res)
