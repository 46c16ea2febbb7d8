"""GPU: Qwen3 codec decoder with 3 / 2 / 1 operand planes — waveform RMS against the reference module's fp32 run (fixture g4,
full-size decoder, 2 requests x 30 frames) and the time of one 10-frame chunk at 1 and 32 requests."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import qwen3_codec_ref as CR            # (development probe: the fixture's weights come from the oracle's generator)
from vox_serve_amd import _native as N
from vox_serve_amd.tokenizer.qwen3_codec import Qwen3CodecConfig, Qwen3TTSDecoder

rms = lambda a: float(np.sqrt(np.mean(np.square(np.asarray(a, dtype=np.float64)))))
dev = torch.device("cuda")
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "g4_qwen3_codec.npz"))
cfg = CR.CodecCfg()
W = CR.random_codec_weights(cfg, seed=0)
pc = Qwen3CodecConfig(**{k: getattr(cfg, k) for k in Qwen3CodecConfig.__dataclass_fields__})
codes = torch.from_numpy(g["full_codes"].astype(np.int64))
ref32 = g["full_fp32_c10"].astype(np.float32)
for planes in (3, 2, 1):
    dec = Qwen3TTSDecoder(W, pc, device=dev, max_batch=32, max_slots=40, detokenize_interval=10)
    N.check(dec.L.vox_codec_set_operand_planes(dec.h, planes))
    cache = dec.init_cache(2)
    got = torch.cat([dec.decode_chunk(codes[:, :, t:t + 10], cache)[0].cpu().clone() for t in range(0, 30, 10)], -1).numpy()
    dec.release_cache(cache)
    line = f"planes {planes}: rms vs reference fp32 {rms(got - ref32):.3e} (signal rms {rms(ref32):.3e})"
    for B in (1, 32):
        cache = dec.init_cache(B)
        c = torch.randint(0, 2048, (B, 16, 10))
        for _ in range(4):
            dec.decode_chunk(c, cache)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            dec.decode_chunk(c, cache)
        torch.cuda.synchronize()
        line += f"; B={B} chunk {(time.perf_counter() - t0) * 100:.2f} ms"
        dec.release_cache(cache)
    print(line, flush=True)
    dec.close()
