"""Qwen3 codec chunk time alone (development aid): python tools/codec_timing.py [B] [frames]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vox_serve_amd.synth import synth_qwen3_codec_weights
from vox_serve_amd.tokenizer.qwen3_codec import Qwen3TTSDecoder
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 10
prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"
dev = torch.device("cuda")
dec = Qwen3TTSDecoder(synth_qwen3_codec_weights(seed=0), device=dev, max_batch=B, max_slots=B, detokenize_interval=T, operand_precision=prec)
codes = torch.randint(0, 2048, (B, 16, T))
cache = dec.init_cache(B)
for _ in range(3):
    dec.decode_chunk(codes, cache)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    dec.decode_chunk(codes, cache)
torch.cuda.synchronize()
print(f"B={B} T={T} operands {prec}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per chunk")
