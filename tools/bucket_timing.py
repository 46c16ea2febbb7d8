"""Development aid: time of vox_sample in bucket mode (top-p over a large vocabulary) per call: python tools/bucket_timing.py [B] [V]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vox_serve_amd import _native as N
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
V = int(sys.argv[2]) if len(sys.argv) > 2 else 168960
dev = torch.device("cuda")
lg = (torch.randn(B, V, device=dev) * 3).to(torch.bfloat16)
out = torch.empty(B, dtype=torch.int32, device=dev)
cfg = N.SamplingCfg(0, 0, 0.8, 0.0, 0.8, 1.0)
L = N.lib()
for i in range(5):
    N.check(L.vox_sample(N.ctx(), N.stream(), lg.data_ptr(), B, V, cfg, 77, i, out.data_ptr()))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 50
for i in range(n):
    N.check(L.vox_sample(N.ctx(), N.stream(), lg.data_ptr(), B, V, cfg, 77, 10 + i, out.data_ptr()))
e1.record(); torch.cuda.synchronize()
print(f"B={B} V={V} top-p 0.8 T 0.8: {e0.elapsed_time(e1) / n * 1e3:.1f} us per call")
