"""GPU: the persistent depth step (VOX_DEPTH_PERSIST=1: one launch per depth step of a one-request frame) against the launch
chain, same process, same weights: every frame's ids, codec logits, depth logits and fed-back features must be bit-identical
over a free-running stream; then the frame time of both."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vox_serve_amd.engine import Qwen3Cfg, Qwen3Engine
from vox_serve_amd.synth import synth_qwen3_weights

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda")
cfg = Qwen3Cfg()
W = synth_qwen3_weights(cfg, dev, seed=0)
ps = 128


def make(persist):
    os.environ["VOX_DEPTH_PERSIST"] = "1" if persist & 1 else "0"
    os.environ["VOX_TALKER_PERSIST"] = "1" if persist & 2 else "0"
    e = Qwen3Engine(cfg, W, max_batch=1, page_size=ps, max_pages=64, max_seq_len=2304, max_prefill_rows=128, keep_depth_logits=True)
    e.keep_hidden = False
    g = torch.Generator(device=dev).manual_seed(5)
    e.kv[:, :3] = (torch.randn(e.kv[:, :3].shape, generator=g, device=dev) * 0.5).to(e.kv.dtype)
    e.input_ids.zero_(); e.input_ids[:, -1] = cfg.tts_pad_id; e.input_ids[:, 0] = 17
    e.input_masks[:1] = 1
    e.input_features.zero_()
    return e


def plan(e, kvlen):
    pages = [list(range((kvlen + ps - 1) // ps))]
    e.upload_plan(pos=[kvlen], kvlen=[kvlen], page=[pages[0][-1]], slot=[(kvlen - 1) % ps], indptr=[0, len(pages[0])], indices=pages[0])


MODE = int(os.environ.get('PERSIST_MODE', '3'))       # bit 0: depth steps, bit 1: talker MLP halves
ea, eb = make(0), make(MODE)
print("persist enabled:", ea.depth_persist_status(), eb.depth_persist_status())
assert eb.depth_persist_status()[0] == MODE and not ea.depth_persist_status()[0]
bad = 0
for use_graph in (False, True):
    sc = ea.sampling_cfg(greedy=True) if use_graph else ea.sampling_cfg(greedy=False, top_k=50, temperature=0.9)
    for f in range(frames):
        for e in (ea, eb):
            plan(e, 200 + f)
            e.frame(1, 200 + f, sc, seed=3, feedback=True, use_graph=use_graph)
        torch.cuda.synchronize()
        same = (torch.equal(ea.out_ids[:1], eb.out_ids[:1]) and torch.equal(ea.out_logits[:1], eb.out_logits[:1])
                and torch.equal(ea.out_depth_logits[:, :1], eb.out_depth_logits[:, :1]) and torch.equal(ea.next_features[:1], eb.next_features[:1])
                and torch.equal(ea.input_features[:1], eb.input_features[:1]))
        if not same:
            bad += 1
            if bad < 4:
                d = (ea.out_depth_logits[:, 0].float() - eb.out_depth_logits[:, 0].float()).abs().amax(dim=-1)
                print(f"graph={use_graph} frame {f}: MISMATCH ids {ea.out_ids[0].tolist()} vs {eb.out_ids[0].tolist()}; max |dlogit diff| per step {d.tolist()}")
    print(f"graph={use_graph}: {frames} frames compared, mismatching frames so far {bad}; status {eb.depth_persist_status()}")
# depth KV caches of both engines must agree too
print("depth-side state equal:", torch.equal(ea.kv, eb.kv))
# timing
sc = ea.sampling_cfg(greedy=True)
for name, e in (("launch chain", ea), ("persistent  ", eb)):
    for w_ in range(5):
        plan(e, 300 + w_); e.frame(1, 300 + w_, sc, feedback=True)
    torch.cuda.synchronize()
    ms = []
    for f in range(100):
        plan(e, 305 + f)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(e.stream); e.frame(1, 305 + f, sc, feedback=True); ev1.record(e.stream)
        e.out_ids[:1].cpu()
        ms.append(ev0.elapsed_time(ev1))
    print(f"{name}: frame {np.mean(ms):.4f} ms (median {np.median(ms):.4f})")
print("final status", eb.depth_persist_status(), "RESULT", "OK" if bad == 0 and eb.depth_persist_status()[1] == 0 else "FAIL")
