"""Development build: phase stamps of the decode attention INSIDE the one-request talker layer launch (block 0, thread 0 of
k_talker_mlp<true>): 0 entry, 1 K/V requested, 2 q / new-k prologue done, 3 tiles parked, 4 scores, 5 softmax, 6 P.V, 7 merge weights,
8 published.  VOX_LIB=tools/bin/libvoxhip_dev.so; the persistent depth step (same stamp buffer) is switched off."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VOX_DEPTH_PERSIST"] = "0"
import numpy as np, torch
from vox_serve_amd import _native as N
from vox_serve_amd.engine import Qwen3Cfg, Qwen3Engine
from vox_serve_amd.synth import synth_qwen3_weights
kv0 = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda")
cfg = Qwen3Cfg()
e = Qwen3Engine(cfg, synth_qwen3_weights(cfg, dev, seed=0), max_batch=1, page_size=128, max_pages=64, max_seq_len=2304, max_prefill_rows=128)
e.keep_hidden = False
e.kv[:, :3].normal_(0, 0.5)
e.input_ids.zero_(); e.input_ids[:, -1] = cfg.tts_pad_id
sc = e.sampling_cfg(greedy=True)
def plan(kvlen):
    pages = list(range((kvlen + 127) // 128))
    e.upload_plan(pos=[kvlen], kvlen=[kvlen], page=[pages[-1]], slot=[(kvlen - 1) % 128], indptr=[0, len(pages)], indices=pages)
for w_ in range(5):
    plan(kv0 + w_); e.frame(1, kv0 + w_, sc)
torch.cuda.synchronize()
st = torch.zeros(32 * 2002, dtype=torch.int64, device=dev)
fn = N.lib().vox_dev_set_stamps2; fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p]
assert fn(st.data_ptr()) == 0
for f in range(10):
    plan(kv0 + 5 + f); e.frame(1, kv0 + 5 + f, sc); torch.cuda.synchronize()
fn(None)
s = st.cpu().numpy().reshape(-1, 32)
n = int(s[0, 0]); full = s[1:1 + min(n, 2000)].astype(np.float64) * 0.01; s = full[:, :9]
pq = full[:, 14:16] - full[:, :1]
print("page ids + row length requested at %.2f us, returned at %.2f us from entry" % tuple(np.median(pq, axis=0)))
pr = full[:, 11:14] - full[:, :1]
print("probes (us from entry): plain load of kvlen returned %.2f | epoch word (L1-bypassing load) returned %.2f | page-table word returned %.2f" % tuple(np.median(pr, axis=0)))
names = ["K/V requested", "q / new k prologue", "tiles parked (barrier)", "scores (barrier)", "softmax (barrier)", "P.V (barrier)", "merge weights (barrier)", "merged + published"]
d = np.diff(s, axis=1)
print(f"kv~{kv0}: {n} launches stamped (persist status {e.depth_persist_status()}); per phase mean / median us")
for i, nm in enumerate(names):
    print(f"  {nm:26s} {d[:, i].mean():6.2f} {np.median(d[:, i]):6.2f}")
print(f"  {'entry -> published':26s} {(s[:, 8] - s[:, 0]).mean():6.2f}")
if (full[:, 10] > 0).any():      # all-layer launch: the new token's q | k | v gathered from the previous layer's granules (stamp 10), behind the tile (stamp 1)
    m = full[:, 10] > 0
    print(f"  all-layer form: tile landed -> q|k|v gathered {np.median(full[m, 10] - full[m, 1]):6.2f} (median, layers 1..)")
