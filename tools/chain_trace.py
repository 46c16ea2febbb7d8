"""Chain trace of a Qwen3-TTS frame (development build only: tools/build_dev.sh, VOX_LIB=tools/bin/libvoxhip_dev.so).
Thread 0 of block 0 of every instrumented launch (k_gemm_fullk, k_attn_short, k_attn_decode8) writes s_memrealtime stamps
(100 MHz); thread 0 of the last block its entry / exit.  Prints, for the last frame: per launch the gap since the previous
instrumented launch's block-0 exit, the in-kernel phases, and the last block's entry / exit relative to block 0's entry;
then a per-kind summary.  python tools/chain_trace.py [B] [frames] [kv]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vox_serve_amd import _native as N
from vox_serve_amd.engine import Qwen3Cfg, Qwen3Engine
from vox_serve_amd.synth import synth_qwen3_weights

CAP = 6000


def summarize(a, CAP=6000):
    """a: the trace buffer as a numpy int64 array (layout: vox_dev_set_trace in csrc/kernels_lm.hip)."""
    n1, n2 = int(a[0]), int(a[1])
    rec = a[16:16 * (n1 + 1)].reshape(n1, 16)
    last = a[16 * (CAP + 1):16 * (CAP + 1) + 2 * n2].reshape(n2, 2)
    us = lambda t: t * 0.01
    KIND = {1: "fullk", 2: "attn_short", 3: "attn_decode8"}
    rows = []
    prev_exit = None
    t_first = rec[0, 3]
    for i in range(n1):
        kind, n, grid = int(rec[i, 0]), int(rec[i, 1]), int(rec[i, 2])
        t = rec[i, 3:11].astype(np.int64); te = int(rec[i, 11])
        gx, gy = grid >> 32, grid & 0xffffffff
        if kind == 1:
            name = f"fullk<MT{n >> 24},KS{(n >> 16) & 255},P{(n >> 12) & 15},E{(n >> 8) & 15},{'RS' if n & 16 else '--'},CT{n & 15}> g{gx}"
        elif kind == 2:
            name = f"attn_short<{n}> g{gx}"
        else:
            name = f"attn_decode8<G{n >> 8},NCH{n & 255}> g{gx}x{gy}"
        gap = us(t[0] - prev_exit) if prev_exit is not None else 0.0
        ph = [us(x - t[0]) if x else None for x in t[1:]]
        lb = (us(last[i, 0] - t[0]), us(last[i, 1] - t[0])) if i < n2 else (None, None)
        rows.append((name, gap, ph, us(te - t[0]), lb, us(t[0] - t_first)))
        prev_exit = te
    print(f"{n1} instrumented launches; columns: start(us) | gap since previous instrumented exit | phases rel. entry | exit | last block entry/exit")
    show = rows if os.environ.get("TRACE_ALL") else rows[:12] + rows[len(rows) // 2: len(rows) // 2 + 30]
    for name, gap, ph, ex, lb, st in show:
        phs = " ".join("  -  " if p is None else f"{p:5.2f}" for p in ph[:5])
        print(f"{st:8.2f} {name:44s} gap {gap:6.2f} | {phs} | exit {ex:5.2f} | last {lb[0]:5.2f} {lb[1]:5.2f}")
    from collections import defaultdict
    agg = defaultdict(list)
    for name, gap, ph, ex, lb, st in rows[1:]:
        agg[name].append((gap, ex, lb[0], lb[1], *[p if p is not None else np.nan for p in ph[:4]]))
    print("\nper kind: count | median gap before | block-0 entry->exit | last block entry, exit (rel. block-0 entry) | phase stamps 1..4 (rel. entry)")
    tot = 0.0
    for name, v in sorted(agg.items(), key=lambda kv: -len(kv[1]) * np.median([x[0] + x[1] for x in kv[1]])):
        v = np.array(v, dtype=np.float64)
        med = np.nanmedian(v, axis=0)
        pitch = med[0] + med[1]
        tot += pitch * len(v)
        print(f"{name:44s} n={len(v):4d} gap {med[0]:5.2f} body {med[1]:5.2f} last {med[2]:5.2f} {med[3]:5.2f} | " + " ".join(f"{x:5.2f}" for x in med[4:]) +
              f" | pitch*n {pitch * len(v) / 1000:6.3f} ms")
    print(f"sum of (gap + body) over instrumented launches: {tot / 1000:.3f} ms (gaps include the un-instrumented launches in between)")


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4
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
    buf = torch.zeros(16 * (CAP + 1) + 2 * CAP, dtype=torch.int64, device=dev)
    lib = N.lib()
    fn = lib.vox_dev_set_trace
    fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for f in range(frames):
        plan(kvlen0 + 5 + f)
        buf.zero_()
        torch.cuda.synchronize()
        assert fn(buf.data_ptr()) == 0
        e0.record(eng.stream)
        eng.frame(B, kvlen0 + 5 + f, sc, use_graph=True)
        e1.record(eng.stream)
        torch.cuda.synchronize()
    fn(None)
    print(f"B={B} kv~{kvlen0}: traced frame graph time {e0.elapsed_time(e1):.3f} ms (stamps cost a little)")
    summarize(buf.cpu().numpy(), CAP)
