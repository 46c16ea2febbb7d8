"""GPU: a few Qwen3 codec chunks at B requests with P operand planes (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vox_serve_amd import _native as N
from vox_serve_amd.synth import synth_qwen3_codec_weights
from vox_serve_amd.tokenizer.qwen3_codec import Qwen3TTSDecoder
B, P = int(sys.argv[1]), int(sys.argv[2])
graph = len(sys.argv) > 3 and sys.argv[3] == "graph"
dec = Qwen3TTSDecoder(synth_qwen3_codec_weights(seed=0), device=torch.device("cuda"), max_batch=B, max_slots=B + 2, detokenize_interval=10)
dec.use_graph = graph
N.check(dec.L.vox_codec_set_operand_planes(dec.h, P))
cache = dec.init_cache(B)
c = torch.randint(0, 2048, (B, 16, 10))
for _ in range(3):
    dec.decode_chunk(c, cache)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(8):
    dec.decode_chunk(c, cache)
torch.cuda.synchronize()
print(f"B={B} planes={P} graph={graph}: chunk {(time.perf_counter() - t0) / 8 * 1e3:.2f} ms")
