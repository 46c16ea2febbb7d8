"""Time vox_linear / vox_linear_silu_mul through the C ABI for given (B,N,K) with rotating weight buffers."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vox_serve_amd import _native as N
dev = torch.device("cuda")
N.ctx()
shapes = [(32, 1024, 2048), (32, 1024, 3072), (32, 4096, 1024), (32, 4096, 2048), (32, 2048, 6144), (16, 1024, 2048), (1, 1024, 2048), (8, 1024, 2048), (75, 4096, 2048)]
if os.environ.get("LT_SHAPES"):          # e.g. LT_SHAPES=8x896x4864,8x4096x13696
    shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["LT_SHAPES"].split(",")]
st = torch.cuda.Stream()
for B, Nn, K in shapes:
    nbuf = max(2, int(600e6 // (Nn * K * 2)))
    Ws = [torch.randn(Nn, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(min(nbuf, 40))]
    x = torch.randn(B, K, device=dev, dtype=torch.bfloat16)
    y = torch.empty(B, Nn, device=dev, dtype=torch.bfloat16)
    with torch.cuda.stream(st):
        for i in range(5):
            N.check(N.lib().vox_linear(N.ctx(), N.stream(), N.ptr(Ws[i % len(Ws)]), None, N.ptr(x), None, N.ptr(y), B, Nn, K, 0))
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 200
        e0.record()
        for i in range(it):
            N.check(N.lib().vox_linear(N.ctx(), N.stream(), N.ptr(Ws[i % len(Ws)]), None, N.ptr(x), None, N.ptr(y), B, Nn, K, 0))
        e1.record(); st.synchronize()
    us = e0.elapsed_time(e1) * 1000 / it
    print(f"B={B:3d} N={Nn:5d} K={K:5d}  {us:7.2f} us  {Nn*K*2/us/1e6:5.2f} TB/s")
