"""BASELINE config 4, one GPU's share: GLM-4-Voice-9B (random-init weights of the named architecture), B = 8 of the 64
data-parallel requests, the model's default top-p 0.8 / T 0.8 sampling over the 168 960-entry vocabulary.  One step = one
LM token for the batch (GLM emits 12.5 audio tokens/s; 25 tokens -> 44 032 samples at 22.05 kHz through the flow + HiFT
detokenizer).  GLM interleaves 13 text tokens with 26 audio tokens: 2/3 of the LM steps yield audio tokens, so one 25-token window (one
hipGraph: block conformer encoder, length regulator, 10-step CFM, HiFT) is decoded every 37.5 steps.  Development measurement; prints one
JSON line with the LM roofline block and the end-to-end audio rate."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vox_serve_amd.engine import LMEngine
from vox_serve_amd.model.glm_voice import GLMVoiceConfig, pack_glm_weights
from vox_serve_amd.synth import synth_glm_weights

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--warmup", type=int, default=20)
ap.add_argument("--greedy", action="store_true")
ap.add_argument("--exact-rows", type=int, default=2)
ap.add_argument("--gpus", type=int, default=1, help="BASELINE config 5: data-parallel replicas, one process per GPU (8 requests each = 64 "
                "concurrent requests on 8 GPUs); without a launcher the script starts its ranks itself under torch.distributed.run")
args = ap.parse_args()
if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    import socket, subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
                              "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:], env=env))
rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
if world != args.gpus:
    raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
torch.cuda.set_device(local)
B, dev = args.batch, torch.device("cuda", local)
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", device_id=dev)       # replicas only: a barrier and a max-reduce of the time, nothing on the token path
if args.exact_rows != 2:
    from vox_serve_amd import _native as _N
    _N.set_exact_rows(args.exact_rows)
gc = GLMVoiceConfig()
layers, norm, emb, head = pack_glm_weights(synth_glm_weights(gc, dev), gc)
eng = LMEngine(gc.lm_cfg(4096), layers, norm, emb, head, None, max_batch=B, page_size=128, max_pages=4 * B + 1, max_seq_len=1024, max_prefill_rows=64)
ps, n0 = 128, 64
rng = np.random.default_rng(1)
pages = [[b * 4 + j for j in range(4)] for b in range(B)]
sc = eng.sampling_cfg(greedy=True) if args.greedy else eng.sampling_cfg(greedy=False, top_k=0, top_p=0.8, temperature=0.8)
for b in range(B):                       # 64-token prompt per request
    eng.kv[:, pages[b][0], :, :n0].normal_(0, 0.5)
eng.input_ids[:B, 0] = torch.from_numpy(rng.integers(152353, 168000, B).astype(np.int32)).to(dev)
kv, pos = [n0] * B, [n0 + 1] * B
from vox_serve_amd.synth import synth_glm_codec_weights
from vox_serve_amd.tokenizer.glm import GLMAudioDecoder
cw = synth_glm_codec_weights()
dec = GLMAudioDecoder(cw["flow"], cw["hift"], device=dev, max_batch=B)
ev, chunk_ms, samples, audio_tok = [], [], [0], [0.0]
win = torch.randint(0, 16384, (B, 25))

def step(timed):
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
    ids = eng.out_ids[:B].cpu()
    pos = [p + 1 for p in pos]
    audio_tok[0] += 26.0 / 39.0                      # audio tokens per LM step in the interleaved stream
    if audio_tok[0] >= 25.0:
        audio_tok[0] -= 25.0
        t0_ = time.perf_counter()
        pcm = (dec(win) * 32767).to(torch.int16).cpu()
        if timed:
            chunk_ms.append((time.perf_counter() - t0_) * 1e3); samples[0] += pcm.numel()

for i in range(max(args.warmup, 80)):
    step(False)
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
t0 = time.perf_counter()
for i in range(args.steps):
    step(True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
if world > 1:
    dist.barrier()
    t_ = torch.tensor([dt, float(samples[0])], device=dev, dtype=torch.float64)
    tmax, ssum = t_.clone(), t_.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX); dist.all_reduce(ssum, op=dist.ReduceOp.SUM)
    dt, samples[0] = float(tmax[0].item()), float(ssum[1].item())
    dist.destroy_process_group()
    if rank != 0:
        sys.exit(0)
frame_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
wbytes = sum(t.numel() * 2 for l in layers for t in l.values()) + head.numel() * 2
print(json.dumps({"workload": f"GLM-4-Voice-9B bf16 LM, batch={B}/GPU x {world} GPU(s) (data-parallel replicas), {'greedy' if args.greedy else 'top-p 0.8 T 0.8 over 168960 ids'}, 64-token context",
                  "n_gpus": world, "concurrent_requests": B * world, "scaling": "weak",
                  "lm_tokens_per_s": world * B * args.steps / dt, "audio_seconds_per_s": world * B * args.steps / dt / 12.5, "ms_per_step": dt / args.steps * 1e3,
                  "lm_graph_ms": frame_ms, "weight_bytes_streamed": wbytes, "exact_rows": args.exact_rows,
                  "audio_samples_per_s": samples[0] / dt, "detokenizer_window_ms": float(np.mean(chunk_ms)) if chunk_ms else None,
                  "roofline": {"bound": "hbm", "achieved": wbytes / (frame_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                               "frac": wbytes / (frame_ms * 1e-3) / 8e12, "traffic": None, "algorithmic_bytes_per_launch": wbytes,
                               "launch": "one hipGraph replay = one token step (40 layers + head + sampler)"}}))
