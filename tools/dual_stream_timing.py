"""Two independent engines (sub-batches) on two HIP streams, frames issued alternately without joining the streams: does the
latency-bound token loop of one sub-batch fill the idle CUs of the other?  (development experiment)
usage: dual_stream_timing.py <B per engine> [frames] [n_engines]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vox_serve_amd.engine import Qwen3Cfg, Qwen3Engine
from vox_serve_amd.synth import synth_qwen3_weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 50
NE = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda")
cfg = Qwen3Cfg()
W = synth_qwen3_weights(cfg, dev, seed=0)
ps, kv0 = 128, 200
engs = []
for e in range(NE):
    eng = Qwen3Engine(cfg, W, max_batch=B, page_size=ps, max_pages=max(64, 4 * B), max_seq_len=2304, max_prefill_rows=16)
    eng.keep_hidden = False
    for b in range(B):
        eng.kv[:, b * 3:(b + 1) * 3].normal_(0, 0.5)
    eng.input_ids.zero_(); eng.input_ids[:, -1] = cfg.tts_pad_id
    engs.append(eng)
sc = engs[0].sampling_cfg(greedy=True)
def plan(eng, kvlen):
    pages = [[b * 3 + j for j in range((kvlen + ps - 1) // ps)] for b in range(B)]
    indptr = np.cumsum([0] + [len(p) for p in pages]); indices = sum(pages, [])
    eng.upload_plan(pos=[kvlen] * B, kvlen=[kvlen] * B, page=[p[-1] for p in pages], slot=[(kvlen - 1) % ps] * B, indptr=indptr, indices=indices)
cs = [torch.cuda.Stream(device=dev) for _ in engs]      # one caller stream per engine: the engines' entry/exit fences must not chain through a common stream
for w in range(5):
    for eng, c in zip(engs, cs):
        with torch.cuda.stream(c):
            plan(eng, kv0 + w); eng.frame(B, kv0 + w, sc)
torch.cuda.synchronize()
t0 = time.perf_counter()
for f in range(frames):
    for eng, c in zip(engs, cs):
        with torch.cuda.stream(c):
            plan(eng, kv0 + 5 + f)
            eng.frame(B, kv0 + 5 + f, sc)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{NE} engines x B={B}: {dt / frames * 1e3:.3f} ms per round of {NE * B} requests -> {NE * B * 1920 * frames / dt:.0f} samples/s")
