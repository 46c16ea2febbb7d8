"""BASELINE config 3 on one MI355X: CSM-1B (random-init weights of the named architecture) + Mimi, B concurrent requests.
One step = one audio frame for the batch: backbone decode + codebook-0 sampling + 31 depth steps (one hipGraph), host plan
upload + token read-back, and every 10th step one stateless Mimi chunk (10 frames -> 19200 samples per request) with
PCM16 packing.  Development measurement (bench.py is the contract); prints one JSON line."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vox_serve_amd.engine import CSMCfg, CSMEngine
from vox_serve_amd.synth import synth_csm_weights

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--warmup", type=int, default=20)
ap.add_argument("--topk", action="store_true", help="the reference's default sampling for CSM-1B (top-k 50 at temperature 0.9, csm.py:355-360) instead of greedy")
args = ap.parse_args()
B, dev = args.batch, torch.device("cuda")
cfg = CSMCfg()
W = synth_csm_weights(cfg, dev)
# algorithmic bytes of one frame: every weight once (backbone, depth decoder, heads) + K/V of the visible context; activations are
# negligible.  As scheduled, the depth decoder's weights are read 31 times (one pass per codebook 1..31, Infinity-Cache hits)
nb = lambda pred: sum(v.numel() * v.element_size() for k, v in W.items() if pred(k) and "embed" not in k)
bytes_backbone = nb(lambda k: k.startswith("backbone_model.") or k == "lm_head.weight")
bytes_depth = nb(lambda k: k.startswith("depth_decoder.model.layers.") or k.startswith("depth_decoder.model.norm") or "inputs_embeds_projector" in k)
bytes_heads = nb(lambda k: k.startswith("depth_decoder.codebooks_head"))
eng = CSMEngine(cfg, W, max_batch=B, page_size=128, max_pages=4 * B + 1, max_seq_len=2304, max_prefill_rows=128)
eng.keep_hidden = False
# Mimi with random-init weights of the reference configuration
from vox_serve_amd.synth import synth_mimi_weights
from vox_serve_amd.tokenizer.mimi import MimiConfig, MimiDecoder
mimi = MimiDecoder(synth_mimi_weights(MimiConfig(), seed=0), MimiConfig(), device=dev, max_batch=B, max_frames=10)
C1, ps, n0 = cfg.n_codebooks + 1, 128, 64
rng = np.random.default_rng(1)
pages = [[b * 4 + j for j in range(4)] for b in range(B)]
sc = eng.sampling_cfg(greedy=False, top_k=50, temperature=0.9) if args.topk else eng.sampling_cfg(greedy=True)
for b in range(B):                       # one prefill per request: 64 text rows
    ids = np.zeros((n0, C1), np.int32); ids[:, -1] = rng.integers(0, 128000, n0)
    masks = np.zeros((n0, C1), np.uint8); masks[:, -1] = 1
    eng.row_ids[:n0], eng.row_masks[:n0] = torch.from_numpy(ids).to(dev), torch.from_numpy(masks).to(dev)
    eng.upload_plan(pos=np.arange(n0), kvlen=np.arange(1, n0 + 1), page=[pages[b][0]] * n0, slot=np.arange(n0), q_req=np.zeros(n0),
                    last_rows=[n0 - 1], indptr=[0, 1], indices=pages[b][:1])
    eng.prefill(n0, 1, n0, sc, feedback=True)
    st = (eng.input_ids[0].clone(), eng.input_masks[0].clone())
    if b == 0:
        states = []
    states.append(st)
for b, (i_, m_) in enumerate(states):
    eng.input_ids[b], eng.input_masks[b] = i_, m_
kv, pos = [n0] * B, [n0 + 1] * B
ring = torch.zeros(B, 10, C1, dtype=torch.int32, device=dev)
ev = []

def step(i, timed):
    global kv, pos
    kv = [k + 1 for k in kv]
    npg = [(k + ps - 1) // ps for k in kv]
    indptr = np.concatenate([[0], np.cumsum(npg)])
    eng.upload_plan(pos=pos, kvlen=kv, page=[pages[b][npg[b] - 1] for b in range(B)], slot=[(k - 1) % ps for k in kv],
                    indptr=indptr, indices=sum([pages[b][:npg[b]] for b in range(B)], []))
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(eng.stream)
    eng.frame(B, max(kv), sc, feedback=True)
    if timed:
        e1.record(eng.stream); ev.append((e0, e1))
    ring[:, i % 10] = eng.out_ids[:B]
    ids = eng.out_ids[:B].cpu()
    pos = [p + 1 for p in pos]
    if i % 10 == 9:
        # the chunk runs on its own stream beside the following frames (like bench.py); its PCM is fetched one chunk later
        global pend
        done = None
        if pend is not None:
            pend[1].synchronize(); done = pend[0].numpy().copy()
        snap = ring.clone()
        codec_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(codec_stream):
            wav = mimi.decode(snap, code_layout="BTQ")
            pcm_host.copy_((wav[:, 0] * 32767).to(torch.int16), non_blocking=True)
            ev_ = torch.cuda.Event(); ev_.record(codec_stream)
        snap.record_stream(codec_stream)
        pend = (pcm_host, ev_)
        return done

codec_stream = torch.cuda.Stream(device=dev)
pcm_host = torch.zeros(B, 19200, dtype=torch.int16).pin_memory()
pend = None
for i in range(args.warmup):
    step(i, False)
torch.cuda.synchronize()
if os.environ.get("VOX_TRACE"):          # development build (VOX_LIB=tools/bin/libvoxhip_dev.so): chain trace of one frame, then exit
    import ctypes
    from vox_serve_amd import _native as N
    from tools.chain_trace import summarize, CAP
    buf = torch.zeros(16 * (CAP + 1) + 2 * CAP, dtype=torch.int64, device=dev)
    fn = N.lib().vox_dev_set_trace
    fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p]
    for i in range(3):
        buf.zero_(); torch.cuda.synchronize(); assert fn(buf.data_ptr()) == 0
        step(1 + i, False); torch.cuda.synchronize()
    fn(None)
    summarize(buf.cpu().numpy(), CAP)
    sys.exit(0)
t0 = time.perf_counter()
for i in range(args.steps):
    step(i, True)
if pend is not None:
    pend[1].synchronize()                 # the last chunk's audio is on the host inside the timed region
torch.cuda.synchronize()
dt = time.perf_counter() - t0
t1 = time.perf_counter(); mimi.decode(ring, code_layout="BTQ"); torch.cuda.synchronize(); t_codec = time.perf_counter() - t1
frame_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
kv_bytes = B * (n0 + args.warmup + args.steps / 2) * cfg.backbone.layers * 2 * cfg.backbone.kv_heads * cfg.backbone.head_dim * 2
print(json.dumps({"workload": f"CSM-1B bf16 + Mimi, batch={B}, {'top-k 50 T 0.9' if args.topk else 'greedy'}, 64-token prompt, detokenize_interval 10",
                  "audio_samples_per_s": B * 1920 * args.steps / dt, "ms_per_step": dt / args.steps * 1e3,
                  "lm_frame_graph_ms": frame_ms, "mimi_chunk_ms": t_codec * 1e3, "realtime_factor_per_request": 1920 * args.steps / dt / 24000,
                  "roofline": (lambda alg, sched: {"bound": "hbm", "achieved": alg / (frame_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                                   "frac": alg / (frame_ms * 1e-3) / 8e12, "traffic": None, "algorithmic_bytes_per_launch": alg,
                                                   "as_scheduled_bytes_per_launch": sched, "as_scheduled_frac": sched / (frame_ms * 1e-3) / 8e12,
                                                   "launch": "one hipGraph replay = one frame (backbone + 31 depth passes + samplers)"})(
                      # algorithmic = every weight byte once per frame (unique bytes; the depth decoder's 0.2 GB are re-read
                      # by each of its 31 passes, served by the Infinity Cache: that figure is reported beside it as `as_scheduled`)
                      bytes_backbone + bytes_depth + bytes_heads + kv_bytes,
                      bytes_backbone + (cfg.n_codebooks - 1) * bytes_depth + bytes_heads + kv_bytes)}))
