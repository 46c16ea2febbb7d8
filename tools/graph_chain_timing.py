"""Development aid: per-kernel cost of dependent chains captured in a hipGraph, through the C ABI.
 (1) one shape repeated, (2) a depth-layer-like mix of shapes (no attention), to separate kernel cost from
 kernel-switch cost."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vox_serve_amd import _native as N
dev = torch.device("cuda")
L, ctx = N.lib(), N.ctx()
st = torch.cuda.Stream()

def chain(shapes, reps, label):
    """shapes: list of (N,K); chain x -> y -> ... with rotating weights."""
    Ws, bufs = [], {}
    for r in range(reps):
        for (Nn, K) in shapes:
            Ws.append(torch.randn(Nn, K, device=dev, dtype=torch.bfloat16) * 0.02)
    x = {k: torch.randn(1, k, device=dev, dtype=torch.bfloat16) for k in {s[1] for s in shapes} | {s[0] for s in shapes}}
    with torch.cuda.stream(st):
        def body():
            i = 0
            for r in range(reps):
                for (Nn, K) in shapes:
                    N.check(L.vox_linear(ctx, N.stream(), N.ptr(Ws[i]), None, N.ptr(x[K]), None, N.ptr(x[Nn]), 1, Nn, K, 0))
                    i += 1
        body(); st.synchronize()
        N.check(L.vox_graph_begin(ctx, N.stream())); body()
        g = ctypes.c_void_p(); N.check(L.vox_graph_end(ctx, N.stream(), ctypes.byref(g)))
        for _ in range(3): N.check(L.vox_graph_launch(g, N.stream()))
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): N.check(L.vox_graph_launch(g, N.stream()))
        e1.record(); st.synchronize()
    n = reps * len(shapes)
    mb = sum(a * b * 2 for a, b in shapes) * reps / 1e6
    us = e0.elapsed_time(e1) * 1000 / 5
    print(f"{label:28s} {n:4d} kernels {mb:7.1f} MB  {us/n:6.2f} us/kernel  {mb/us/1e6*1e6/1e6:5.2f} TB/s")

chain([(1024, 1024)], 300, "1024x1024 repeated")
chain([(4096, 1024)], 75, "4096x1024 repeated")
chain([(1024, 2048)], 150, "1024x2048 repeated")
chain([(4096, 1024), (1024, 2048), (2048, 1024), (3072, 1024), (1024, 3072)], 30, "depth-like mix (5 shapes)")
chain([(4096, 2048), (2048, 2048), (6144, 2048), (2048, 6144)], 28, "talker-like mix (4 shapes)")
