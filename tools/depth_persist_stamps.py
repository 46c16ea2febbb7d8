"""Development build: per-stage time stamps of the persistent depth step (block 0, thread 0): per layer A publish, B qkv gathered,
B attention done, B publish, C publish, D publish (+ entry, end).  VOX_LIB=tools/bin/libvoxhip_dev.so VOX_DEPTH_PERSIST=1."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VOX_DEPTH_PERSIST"] = "1"
import numpy as np, torch
from vox_serve_amd import _native as N
from vox_serve_amd.engine import Qwen3Cfg, Qwen3Engine
from vox_serve_amd.synth import synth_qwen3_weights
dev = torch.device("cuda")
cfg = Qwen3Cfg()
W = synth_qwen3_weights(cfg, dev, seed=0)
e = Qwen3Engine(cfg, W, max_batch=1, page_size=128, max_pages=64, max_seq_len=2304, max_prefill_rows=128)
e.keep_hidden = False
e.kv[:, :3].normal_(0, 0.5)
e.input_ids.zero_(); e.input_ids[:, -1] = cfg.tts_pad_id
sc = e.sampling_cfg(greedy=True)
def plan(kvlen):
    pages = list(range((kvlen + 127) // 128))
    e.upload_plan(pos=[kvlen], kvlen=[kvlen], page=[pages[-1]], slot=[(kvlen - 1) % 128], indptr=[0, len(pages)], indices=pages)
for w_ in range(5):
    plan(200 + w_); e.frame(1, 200 + w_, sc)
torch.cuda.synchronize()
st = torch.zeros(32 * 2002, dtype=torch.int64, device=dev)
fn = N.lib().vox_dev_set_stamps2; fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p]
assert fn(st.data_ptr()) == 0
for f in range(20):
    plan(205 + f); e.frame(1, 205 + f, sc); torch.cuda.synchronize()
fn(None)
s = st.cpu().numpy().reshape(-1, 32)
n = int(s[0, 0]); s = s[1:1 + min(n, 2000)].astype(np.float64) * 0.01     # us
print(f"{n} persistent launches stamped; status {e.depth_persist_status()}")
names = ["A: x gathered -> qkv published", "B: qkv gathered", "B: attention done", "B: x' published", "C: h published", "D: x'' published (+sync)"]
tot = s[:, 31] - s[:, 0]
print(f"launch entry -> end: mean {tot.mean():.2f} us (min {tot.min():.2f})")
seq = [0] + [k + 6 * l for l in range(5) for k in range(1, 7)] + [31]
d = np.diff(s[:, seq], axis=1)
for k in range(6):
    cols = [k + 6 * l for l in range(5)]
    print(f"  {names[k]:34s} mean over layers {d[:, cols].mean():6.2f} us   per layer {np.round(d[:, cols].mean(axis=0), 2).tolist()}")
print(f"  {'head':34s} {d[:, 30].mean():6.2f} us")
g = np.diff(s[:, 0]); g = g[(g > 0) & (g < 400)]
print(f"  entry-to-entry of consecutive steps: median {np.median(g):.2f} us")
# per visible-token count: the frame's 14 persistent launches are steps i = 2 .. 15, NT = i + 1 visible tokens
if n >= 14 and os.environ.get("DS_PER_NT", "1") != "0":
    k = np.arange(len(s)) % 14
    print("  NT : entry->end |  A     B-gather  B-attn  B-pub   C      D      (means over the five layers, us)")
    for j in range(14):
        m = k == j
        row = [d[m][:, [c + 6 * l for l in range(5)]].mean() for c in range(6)]
        print(f"  {j + 3:2d} : {tot[m].mean():9.2f} | " + "  ".join(f"{v:5.2f}" for v in row))
