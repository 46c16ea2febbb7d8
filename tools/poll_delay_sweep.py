"""Coordinate descent over the first-pass hold-back of the persistent kernels' gathers (VOX_DS_POLL_DELAY / VOX_MLP_POLL_DELAY: one byte per
gather site, x 128 clocks), one-request frames, one engine per setting on one box.  python tools/poll_delay_sweep.py [rounds]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vox_serve_amd.engine import Qwen3Cfg, Qwen3Engine
from vox_serve_amd.synth import synth_qwen3_weights

dev = torch.device("cuda")
cfg = Qwen3Cfg()
W = synth_qwen3_weights(cfg, dev, seed=0)
ps, kv0 = 128, 200


def measure(ds, mlp, frames=80, reps=3):
    os.environ["VOX_DS_POLL_DELAY"] = hex(ds)
    os.environ["VOX_MLP_POLL_DELAY"] = hex(mlp)
    eng = Qwen3Engine(cfg, W, max_batch=1, page_size=ps, max_pages=64, max_seq_len=2304, max_prefill_rows=128)
    eng.keep_hidden = False
    eng.kv[:, :3].normal_(0, 0.5)
    eng.input_ids.zero_(); eng.input_ids[:, -1] = cfg.tts_pad_id
    sc = eng.sampling_cfg(greedy=True)
    pages = list(range((kv0 + 40 + ps - 1) // ps))

    def plan(kv):
        eng.upload_plan(pos=[kv], kvlen=[kv], page=[pages[(kv - 1) // ps]], slot=[(kv - 1) % ps], indptr=[0, (kv + ps - 1) // ps], indices=pages[:(kv + ps - 1) // ps])
    for w in range(5):
        plan(kv0 + w); eng.frame(1, kv0 + w, sc)
    torch.cuda.synchronize()
    out = []
    for r in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for f in range(frames):
            plan(kv0 + 5 + f % 30); eng.frame(1, kv0 + 5 + f % 30, sc)
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / frames)
    assert eng.depth_persist_status() == (3, 0), eng.depth_persist_status()
    eng.close()
    return statistics.median(out)


def setb(word, site, v):
    return (word & ~(0xff << (8 * site))) | (v << (8 * site))


ds, mlp = (int(os.environ.get("SWEEP_DS", "0x04040404"), 0), int(os.environ.get("SWEEP_MLP", "0x040404"), 0))
VALUES = [int(v) for v in os.environ.get("SWEEP_VALUES", "0,2,4,6,8,12,16,24").split(",")]
SITES = os.environ.get("SWEEP_SITES")
EXTRA = [e for e in os.environ.get("SWEEP_ENV", "").split(",") if e]      # whole-value integer knobs swept like a site, e.g. VOX_TALKER_ATTN_DELAY
names = [("env", e, e) for e in EXTRA] + [("ds", 0, "x D->A"), ("ds", 1, "qkv A->B"), ("ds", 2, "x B->C"), ("ds", 3, "h C->D"), ("mlp", 0, "x' O->C"), ("mlp", 1, "h C->D"), ("mlp", 2, "x D->qkv"), ("mlp", 3, "attention row (VOX_TALKER_ATTN=1)")]
print(f"zero delays: {measure(0, 0):.4f} ms/frame")
print(f"start ds={ds:#x} mlp={mlp:#x}: {measure(ds, mlp):.4f} ms/frame")
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    for which, site, label in names:
        if SITES and f"{which}{site}" not in SITES.split(",") and which != "env":
            continue
        res = {}
        for v in VALUES:
            if which == "env":
                os.environ[site] = str(v)
                res[v] = measure(ds, mlp)
                continue
            d, m = (setb(ds, site, v), mlp) if which == "ds" else (ds, setb(mlp, site, v))
            res[v] = measure(d, m)
        best = min(res, key=res.get)
        if which == "env":
            os.environ[site] = str(best)
        print(f"round {rnd} {which}[{site}] {label}: " + " ".join(f"{v}:{t:.4f}" for v, t in res.items()) + f"  -> {best}", flush=True)
        if which == "ds":
            ds = setb(ds, site, best)
        elif which == "mlp":
            mlp = setb(mlp, site, best)
print(f"final ds={ds:#x} mlp={mlp:#x} " + " ".join(f"{e}={os.environ.get(e)}" for e in EXTRA) + f": {measure(ds, mlp):.4f} ms/frame")
