"""Phase time stamps of the one-launch decode attention inside real frames (development build only:
tools/build_dev.sh, VOX_LIB=tools/bin/libvoxhip_dev.so).  Block (0, 0) of every k_attn_decode8 launch writes
s_memrealtime (100 MHz) at: 0 entry, 1 K/V loads issued, 2 q prologue done, 3 tiles parked (first barrier), 4 new token
placed, 5 scores, 6 softmax, 7 P.V, 8 merged + stored.  Prints the mean time between consecutive stamps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vox_serve_amd import _native as N
from vox_serve_amd.engine import Qwen3Cfg, Qwen3Engine
from vox_serve_amd.synth import synth_qwen3_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
kvlen0 = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = torch.device("cuda")
cfg = Qwen3Cfg()
W = synth_qwen3_weights(cfg, dev, seed=0)
eng = Qwen3Engine(cfg, W, max_batch=B, page_size=128, max_pages=max(64, 4 * B), max_seq_len=2304, max_prefill_rows=128)
eng.keep_hidden = False
ps = 128
for b in range(B):
    eng.kv[:, b * 3:(b + 1) * 3].normal_(0, 0.5)
sc = eng.sampling_cfg(greedy=True)
eng.input_ids.zero_(); eng.input_ids[:, -1] = cfg.tts_pad_id


def plan(kvlen):
    pages = [[b * 3 + j for j in range((kvlen + ps - 1) // ps)] for b in range(B)]
    indptr = np.cumsum([0] + [len(p) for p in pages]); indices = sum(pages, [])
    eng.upload_plan(pos=[kvlen] * B, kvlen=[kvlen] * B, page=[p[-1] for p in pages], slot=[(kvlen - 1) % ps] * B,
                    indptr=indptr, indices=indices)


for w in range(5):
    plan(kvlen0 + w); eng.frame(B, kvlen0 + w, sc, use_graph=True)
torch.cuda.synchronize()
stamps = torch.zeros(16 * 4002, dtype=torch.int64, device=dev)
import ctypes
lib = N.lib()
fn = lib.vox_dev_set_stamps
fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p]
assert fn(stamps.data_ptr()) == 0
for f in range(frames):
    plan(kvlen0 + 5 + f)
    eng.frame(B, kvlen0 + 5 + f, sc, use_graph=True)
    torch.cuda.synchronize()
fn(None)
s = stamps.cpu().numpy().reshape(-1, 16)
n = int(s[0, 0])
s = s[1:1 + min(n, 4000), :9].astype(np.float64) * 10.0 / 1000.0     # us
d = np.diff(s, axis=1)
names = ["issue K/V loads", "q prologue", "wait K/V + park (barrier)", "new token (barrier)", "scores (barrier)",
         "softmax (barrier)", "P.V", "merge + store"]
print(f"B={B} kv~{kvlen0}: {n} launches stamped; per phase mean / median us")
for i, nm in enumerate(names):
    print(f"  {nm:28s} {d[:, i].mean():6.2f} {np.median(d[:, i]):6.2f}")
print(f"  {'entry -> end':28s} {(s[:, 8] - s[:, 0]).mean():6.2f}")
# gap between consecutive launches' entries (one talker layer) for context
if n > 2:
    g = np.diff(s[:, 0])
    g = g[(g > 0) & (g < 200)]
    print(f"  entry-to-entry of consecutive layers: median {np.median(g):.2f} us")
