# A/B: RoPE + ring write + attention of the codec transformer as one launch (VOX_CODEC_ATTN2=0: two launches), chunk time at 1 / 8 / 32 requests,
# then the codec suite (waveforms must not move: the fused kernel is bit-identical)
for b in 1 8 32; do
  for m in 1 0 1 0; do echo -n "VOX_CODEC_ATTN2=$m "; VOX_CODEC_ATTN2=$m python tools/codec_timing.py $b 10 2>/dev/null | tail -1; done
done
python - <<'PY'
import os, subprocess, sys, hashlib
# waveform digests of the same chunk sequence under both settings
code = """
import sys, os, hashlib, torch
sys.path.insert(0, os.getcwd())
from vox_serve_amd.synth import synth_qwen3_codec_weights
from vox_serve_amd.tokenizer.qwen3_codec import Qwen3TTSDecoder
dev = torch.device('cuda')
dec = Qwen3TTSDecoder(synth_qwen3_codec_weights(seed=0), device=dev, max_batch=4, max_slots=4, detokenize_interval=10)
g = torch.Generator().manual_seed(3)
cache = dec.init_cache(4)
h = hashlib.sha256()
for i in range(12):
    codes = torch.randint(0, 2048, (4, 16, 10), generator=g)
    w = dec.decode_chunk(codes, cache)
    w = w[0] if isinstance(w, (tuple, list)) else w
    h.update(w.float().cpu().numpy().tobytes())
print(h.hexdigest())
"""
outs = []
for m in ("1", "0"):
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VOX_CODEC_ATTN2=m), capture_output=True, text=True)
    outs.append(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:])
print("digests:", outs, "IDENTICAL" if outs[0] == outs[1] else "DIFFERENT")
PY
timeout 600 python -m pytest tests/test_gpu_codec.py -x -q 2>&1 | tail -3
