"""Quick LM-only frame timing (development aid; bench.py is the contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vox_serve_amd.engine import Qwen3Cfg, Qwen3Engine
from vox_serve_amd.synth import synth_qwen3_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 50
use_graph = (sys.argv[3] != "eager") if len(sys.argv) > 3 else True
dev = torch.device("cuda")
if os.environ.get("VOX_EXACT_ROWS"):
    from vox_serve_amd import _native as _N
    _N.set_exact_rows(int(os.environ["VOX_EXACT_ROWS"]))
cfg = Qwen3Cfg()
W = synth_qwen3_weights(cfg, dev, seed=0)
eng = Qwen3Engine(cfg, W, max_batch=B, page_size=128, max_pages=max(64, 4 * B), max_seq_len=2304, max_prefill_rows=128)
eng.keep_hidden = False
n = 75
ps = 128
for b in range(B):
    eng.kv[:, b * 3:(b + 1) * 3].normal_(0, 0.5)
kvlen0 = int(os.environ.get('LM_KV', '200'))
sc = eng.sampling_cfg(greedy=True) if os.environ.get('LM_SAMPLING', 'greedy') == 'greedy' else eng.sampling_cfg(greedy=False, top_k=50, temperature=0.9)      # LM_SAMPLING=topk: the reference's default for Qwen3-TTS
eng.input_ids.zero_(); eng.input_ids[:, -1] = cfg.tts_pad_id
def plan(kvlen):
    pages = [[b * 3 + j for j in range((kvlen + ps - 1) // ps)] for b in range(B)]
    indptr = np.cumsum([0] + [len(p) for p in pages]); indices = sum(pages, [])
    eng.upload_plan(pos=[kvlen] * B, kvlen=[kvlen] * B, page=[p[-1] for p in pages], slot=[(kvlen - 1) % ps] * B,
                    indptr=indptr, indices=indices)
for w in range(5):
    plan(kvlen0 + w); eng.frame(B, kvlen0 + w, sc, use_graph=use_graph)
torch.cuda.synchronize()
if os.environ.get("LM_MODE"):
    # where do the 0.09 ms between back-to-back replays (2.56) and the bench's frames (2.65) go?  a: fixed plan, no host sync;
    # b: + plan upload (H2D in stream order) per frame; c: b + stream-ordered D2H snapshot of the ids; d: c + host wait per frame (pipelined
    # one deep); e: lock-step (upload, frame, blocking .cpu()).  Frame time = HIP events around every replay.
    mode = os.environ["LM_MODE"]
    pin = [torch.zeros_like(eng.out_ids[:B], device="cpu").pin_memory() for _ in range(2)]
    evs = [torch.cuda.Event(), torch.cuda.Event()]
    times = []
    plan(kvlen0 + 5)
    pairs = []
    for f in range(frames):
        if mode in "bcde":
            plan(kvlen0 + 5 + (f % 40))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(eng.stream)
        eng.frame(B, kvlen0 + 5 + (f % 40 if mode in "bcde" else 0), sc, use_graph=True)
        e1.record(eng.stream)
        pairs.append((e0, e1))
        if mode in "cd":
            pin[f & 1].copy_(eng.out_ids[:B], non_blocking=True); evs[f & 1].record(eng.stream)
            if mode == "d" and f > 0:
                evs[(f - 1) & 1].synchronize()
        if mode == "e":
            eng.out_ids[:B].cpu()
    torch.cuda.synchronize()
    ms = np.array([a.elapsed_time(b) for a, b in pairs[20:]])
    print(f"B={B} LM_MODE={mode}: frame graph mean {ms.mean():.4f} ms, median {np.median(ms):.4f}")
    sys.exit(0)
if os.environ.get("LM_ASYNC") == "1":
    # frames enqueued back to back, no host synchronisation in between (GPU never idles): plans of a fixed kv length
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    plan(kvlen0 + 5)
    with eng._OnStream(eng):
        ev0.record()
    for f in range(frames):
        eng.frame(B, kvlen0 + 5, sc, use_graph=use_graph)
    with eng._OnStream(eng):
        ev1.record()
    eng.stream.synchronize()
    print(f"B={B} back-to-back: gpu {ev0.elapsed_time(ev1)/frames:.3f} ms/frame; persistent kernels (enabled bits, hand-off timeouts): {eng.depth_persist_status()}")
    sys.exit(0)
import contextlib
_one = torch.cuda.stream(eng.stream) if os.environ.get("LM_ONE_STREAM") == "1" else contextlib.nullcontext()
_one.__enter__()
t0 = time.perf_counter()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
gpu_ms = 0.0
for f in range(frames):
    plan(kvlen0 + 5 + f)
    with eng._OnStream(eng):
        ev0.record()
    eng.frame(B, kvlen0 + 5 + f, sc, use_graph=use_graph)
    with eng._OnStream(eng):
        ev1.record()
    eng.stream.synchronize()
    ids = eng.out_ids[:B].cpu()
    gpu_ms += ev0.elapsed_time(ev1)
t1 = time.perf_counter()
print(f"B={B} graph={use_graph} wall {(t1-t0)/frames*1e3:.3f} ms/frame, gpu {gpu_ms/frames:.3f} ms/frame, "
      f"samples/s {B*1920*frames/(t1-t0):.0f}; tokens {ids[0,:4].tolist()}")
