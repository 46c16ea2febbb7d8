"""GPU: the fused residual unit (k_res_unit, VOX_RES_UNIT=1) against the two-launch form (=0): run once per setting; the second run
compares its waveform with the first run's file bit for bit and prints the chunk time of both."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vox_serve_amd.synth import synth_qwen3_codec_weights
from vox_serve_amd.tokenizer.qwen3_codec import Qwen3TTSDecoder
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
path = sys.argv[2] if len(sys.argv) > 2 else "/tmp/res_unit_ref.pt"
dec = Qwen3TTSDecoder(synth_qwen3_codec_weights(seed=0), device=torch.device("cuda"), max_batch=B, max_slots=B + 2, detokenize_interval=10)
cache = dec.init_cache(B)
g = torch.Generator().manual_seed(1)
outs = []
for k in range(3):
    c = torch.randint(0, 2048, (B, 16, 10), generator=g)
    outs.append(dec.decode_chunk(c, cache)[0].clone())
torch.cuda.synchronize()
wav = torch.cat(outs, -1).cpu()
c = torch.randint(0, 2048, (B, 16, 10), generator=g)
for _ in range(3):
    dec.decode_chunk(c, cache)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    dec.decode_chunk(c, cache)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 10 * 1e3
mode = os.environ.get("VOX_RES_UNIT", "1")
if os.path.exists(path):
    ref = torch.load(path)
    d = (wav - ref).abs().max().item()
    print(f"B={B} VOX_RES_UNIT={mode}: chunk {ms:.3f} ms; vs first run: max |diff| {d:.3e}, bit-identical {torch.equal(wav, ref)}, rms {wav.pow(2).mean().sqrt().item():.4f}")
else:
    torch.save(wav, path)
    print(f"B={B} VOX_RES_UNIT={mode}: chunk {ms:.3f} ms; saved reference waveform (rms {wav.pow(2).mean().sqrt().item():.4f})")
