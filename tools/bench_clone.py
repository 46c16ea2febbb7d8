"""Prompt-side cost of a voice-clone request at full size (synthetic weights): speaker encoder (mel 128 -> 2048) and speech-tokenizer
encoder (SEANet + 8-layer transformer + RVQ 16) over a reference clip, plus the ICL prompt features.  Prints one JSON line.

  python tools/bench_clone.py [--seconds 5] [--reps 20]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, nargs="*", default=[3.0, 10.0])
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    from vox_serve_amd.model.qwen3_tts_speaker import Qwen3TTSSpeakerEncoder
    from vox_serve_amd.synth import synth_qwen3_codec_encoder_weights, synth_qwen3_speaker_encoder_weights
    from vox_serve_amd.tokenizer.qwen3_codec_encoder import Qwen3TTSTokenizerV2Encoder
    dev = torch.device("cuda:0")
    spk = Qwen3TTSSpeakerEncoder(synth_qwen3_speaker_encoder_weights(), device=dev, max_seconds=max(a.seconds) + 1)
    enc = Qwen3TTSTokenizerV2Encoder(synth_qwen3_codec_encoder_weights(), device=dev, max_seconds=max(a.seconds) + 1)
    out = {"workload": "voice-clone prompt side, Qwen3-TTS-1.7B-Base shapes, synthetic weights, one request", "clips": []}
    rng = np.random.default_rng(0)
    for sec in a.seconds:
        n = int(sec * 24000)
        wav = torch.from_numpy((0.3 * rng.standard_normal(n)).clip(-1, 1).astype(np.float32)).to(dev)
        rec = {"seconds": sec}
        for name, fn in (("speaker_encoder_ms", lambda: spk(wav)), ("codec_encoder_ms", lambda: enc.encode(wav))):
            fn(); fn()
            ts = []
            for _ in range(a.reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            rec[name] = float(np.median(ts))
        rec["frames"] = -(-n // 1920)
        out["clips"].append(rec)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
