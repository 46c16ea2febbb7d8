"""BASELINE config: CosyVoice2-0.5B (random-init weights of the named architecture) on one MI355X, B concurrent requests: the speech-LM step
(one hipGraph per token, top-k 25) and, every 25 tokens, one detokenizer chunk — 28-token window -> flow (conformer encoder + 10-step CFM with
classifier-free guidance against the speaker prompt's static caches) -> HiFT -> 24 000 samples per request (one hipGraph per chunk).
Development measurement (bench.py is the contract); prints one JSON line with a roofline block for the LM step."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vox_serve_amd.engine import LMEngine
from vox_serve_amd.model.cosyvoice2 import CosyVoice2Config, pack_cosyvoice2_weights
from vox_serve_amd.synth import synth_cosyvoice2_codec_weights, synth_cosyvoice2_weights
from vox_serve_amd.tokenizer.cosyvoice2 import CosyVoice2Decoder

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--warmup", type=int, default=25)
args = ap.parse_args()
B, dev = args.batch, torch.device("cuda")
cc = CosyVoice2Config()
W = synth_cosyvoice2_weights(cc, dev)
layers, norm, emb, head, head_b = pack_cosyvoice2_weights(W, cc)
eng = LMEngine(cc.lm_cfg(4096), layers, norm, emb, head, head_b, max_batch=B, page_size=128, max_pages=4 * B + 1, max_seq_len=1024, max_prefill_rows=64)
cw, prompt = synth_cosyvoice2_codec_weights()
dec = CosyVoice2Decoder(cw["flow"], cw["hift"], device=dev, max_batch=B, max_prompt_tokens=64)
dec.init_cache(prompt)
ps, n0 = 128, 64
rng = np.random.default_rng(1)
pages = [[b * 4 + j for j in range(4)] for b in range(B)]
sc = eng.sampling_cfg(greedy=False, top_k=25, top_p=1.0, temperature=1.0)
for b in range(B):
    eng.kv[:, pages[b][0], :, :n0].normal_(0, 0.5)
eng.input_ids[:B, 0] = torch.from_numpy(rng.integers(0, 6561, B).astype(np.int32)).to(dev)
kv, pos = [n0] * B, [n0 + 1] * B
ring = torch.zeros(B, 28, dtype=torch.int32, device=dev)
ev, chunk_ms, samples = [], [], 0

def step(i, timed):
    global kv, pos, samples
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
    ring[:, i % 28] = eng.out_ids[:B].reshape(B, -1)[:, 0].clamp(0, 6560)
    ids = eng.out_ids[:B].cpu()
    pos = [p + 1 for p in pos]
    if i % 25 == 24:                       # a 28-token window every 25 tokens (detokenize_interval 28, overlap 3)
        t0 = time.perf_counter()
        audio, _ = dec.decode_chunk(ring, 28, None)
        pcm = (audio * 32767).to(torch.int16).cpu()
        if timed:
            chunk_ms.append((time.perf_counter() - t0) * 1e3); samples += pcm.numel()

for i in range(args.warmup + 50):
    step(i, False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(args.steps):
    step(i, True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
frame_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
wbytes = sum(t.numel() * 2 for l in layers for t in l.values()) + head.numel() * 2
print(json.dumps({"workload": f"CosyVoice2-0.5B bf16 LM + flow/HiFT detokenizer, batch={B}, top-k 25, 64-token context, 28-token windows every 25 tokens",
                  "audio_samples_per_s": samples / dt, "realtime_factor_per_request": samples / dt / 24000 / B, "ms_per_token_step": dt / args.steps * 1e3,
                  "lm_graph_ms": frame_ms, "detokenizer_chunk_ms": float(np.mean(chunk_ms)) if chunk_ms else None,
                  "roofline": {"bound": "hbm", "achieved": wbytes / (frame_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                               "frac": wbytes / (frame_ms * 1e-3) / 8e12, "traffic": None, "algorithmic_bytes_per_launch": wbytes,
                               "launch": "one hipGraph replay = one LM token step (24 layers + head + sampler)"}}))
