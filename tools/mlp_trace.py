"""Phase stamps of the one-request talker layer launch (k_talker_mlp; development build, VOX_LIB=tools/bin/libvoxhip_dev.so): block 0
(an attention block when the attention runs inside the launch) and block 100 (a plain block).  Stamps, relative to the block's entry:
1 attention done + published (attention blocks) | 2 attention row gathered | 3 x' published, second gate/up pair requested |
4 x' gathered | 5 h published, down rows requested | 6 h gathered | 7 x'' gathered (next layer's q/k/v rows requested before) | end.
python tools/mlp_trace.py [kv]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vox_serve_amd import _native as N
from vox_serve_amd.engine import Qwen3Cfg, Qwen3Engine
from vox_serve_amd.synth import synth_qwen3_weights
kvlen0 = int(sys.argv[1]) if len(sys.argv) > 1 else 200
CAP = 6000
dev = torch.device("cuda")
cfg = Qwen3Cfg()
eng = Qwen3Engine(cfg, synth_qwen3_weights(cfg, dev, seed=0), max_batch=1, page_size=128, max_pages=64, max_seq_len=2304, max_prefill_rows=128)
eng.keep_hidden = False
eng.kv[:, :3].normal_(0, 0.5)
sc = eng.sampling_cfg(greedy=True)
eng.input_ids.zero_(); eng.input_ids[:, -1] = cfg.tts_pad_id
def plan(kv):
    pages = list(range((kv + 127) // 128))
    eng.upload_plan(pos=[kv], kvlen=[kv], page=[pages[-1]], slot=[(kv - 1) % 128], indptr=[0, len(pages)], indices=pages)
for w in range(5):
    plan(kvlen0 + w); eng.frame(1, kvlen0 + w, sc, use_graph=True)
torch.cuda.synchronize()
buf = torch.zeros(16 * (CAP + 1) + 2 * CAP, dtype=torch.int64, device=dev)
fn = N.lib().vox_dev_set_trace
fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p]
for f in range(4):
    plan(kvlen0 + 5 + f); buf.zero_(); torch.cuda.synchronize(); assert fn(buf.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.frame(1, kvlen0 + 5 + f, sc, use_graph=True); e1.record(); torch.cuda.synchronize()
fn(None)
a = buf.cpu().numpy(); n = int(a[0]); rec = a[16:16 * (n + 1)].reshape(n, 16)
print(f"kv~{kvlen0}: frame {e0.elapsed_time(e1):.3f} ms, {n} records")
for kind, name in ((4, "block 0  "), (5, "block 100")):
    r = rec[rec[:, 0] == kind]
    if not len(r):
        continue
    t = r[:, 3:12].astype(np.float64) * 0.01
    rel = t - t[:, :1]
    rel[r[:, 3:12] == 0] = np.nan
    med = np.nanmedian(rel, axis=0)
    print(f"{name} ATTN={int(r[0, 1])} n={len(r)}: " + " ".join(f"{x:6.2f}" for x in med[1:]) + "  (stamps 1..7, end; us from entry)")
    ent = np.sort(t[:, 0]); print(f"   entry-to-entry of consecutive layers: median {np.median(np.diff(ent)):.2f} us")
