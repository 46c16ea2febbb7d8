"""Soak of the persistent kernels: N back-to-back one-request frames (graph replays, lock-step status read every frame), kv cycling over
200..1200 tokens; reports frames, hand-off timeouts (must be 0) and the mean frame time.  python tools/persist_soak.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vox_serve_amd.engine import Qwen3Cfg, Qwen3Engine
from vox_serve_amd.synth import synth_qwen3_weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dev = torch.device("cuda")
cfg = Qwen3Cfg()
e = Qwen3Engine(cfg, synth_qwen3_weights(cfg, dev, seed=0), max_batch=1, page_size=128, max_pages=64, max_seq_len=2304, max_prefill_rows=128)
e.keep_hidden = False
e.kv[:, :12].normal_(0, 0.5)
e.input_ids.zero_(); e.input_ids[:, -1] = cfg.tts_pad_id
sc = e.sampling_cfg(greedy=True)
assert e.depth_persist_status() == (3, 0), e.depth_persist_status()
pages = list(range(12))
t0 = time.perf_counter()
for f in range(n):
    kv = 200 + f % 1000
    npg = (kv + 127) // 128
    e.upload_plan(pos=[kv], kvlen=[kv], page=[pages[npg - 1]], slot=[(kv - 1) % 128], indptr=[0, npg], indices=pages[:npg])
    e.frame(1, kv, sc)
    e.read_ids(1)                      # the status row travels with the ids: a timeout would be recovered (and counted) here
dt = time.perf_counter() - t0
print(f"{n} frames, {len(e.persist_failures)} hand-off timeouts, persist status {e.depth_persist_status()}, {dt / n * 1e3:.3f} ms per frame (lock-step)")
assert not e.persist_failures and e.depth_persist_status() == (3, 0)
