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

def chain(shapes, reps, label, B=1):
    """shapes: list of (N,K); chain x -> y -> ... with rotating weights."""
    Ws, bufs = [], {}
    for r in range(reps):
        for (Nn, K) in shapes:
            Ws.append(torch.randn(Nn, K, device=dev, dtype=torch.bfloat16) * 0.02)
    x = {k: torch.randn(B, k, device=dev, dtype=torch.bfloat16) for k in {s[1] for s in shapes} | {s[0] for s in shapes}}
    with torch.cuda.stream(st):
        def body():
            i = 0
            for r in range(reps):
                for (Nn, K) in shapes:
                    N.check(L.vox_linear(ctx, N.stream(), N.ptr(Ws[i]), None, N.ptr(x[K]), None, N.ptr(x[Nn]), B, Nn, K, 0))
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
    print(f"B={B} {label:28s} {n:4d} kernels {mb:7.1f} MB  {us/n:6.2f} us/kernel  {mb/us/1e6*1e6/1e6:5.2f} TB/s")

if len(sys.argv) > 1 and sys.argv[1] == "rows":
    for B in (1, 2, 4, 8):
        chain([(1024, 2048)], 150, "1024x2048 repeated", B)
        chain([(4096, 1024)], 75, "4096x1024 repeated", B)
        chain([(2048, 6144)], 40, "2048x6144 repeated", B)
    sys.exit(0)


def stack_chain(reps=15):
    """Depth-transformer-shaped stack (5 layers, 1 decode row) through vox_stack_forward, `reps` calls in one graph."""
    from vox_serve_amd.engine import StackCfg, _stack_config, rope_table
    c = StackCfg(1024, 5, 16, 8, 128, 3072)
    sc = _stack_config(c, 16, 2, 16)
    arr = (N.LayerWeights * c.layers)()
    keep = []
    mk = lambda *s: (keep.append(torch.randn(*s, device=dev, dtype=torch.bfloat16) * 0.02) or keep[-1])
    for i in range(c.layers):
        for k, t in dict(wqkv=mk(4096, 1024), wo=mk(1024, 2048), wgate=mk(3072, 1024), wup=mk(3072, 1024), wdown=mk(1024, 3072),
                         ln1=mk(1024) + 1, ln2=mk(1024) + 1, qnorm=mk(128) + 1, knorm=mk(128) + 1).items():
            keep.append(t)
            setattr(arr[i], k, t.data_ptr())
    fn = mk(1024) + 1
    rope = rope_table(64, c, dev)
    h = ctypes.c_void_p()
    N.check(L.vox_stack_create(ctx, ctypes.byref(sc), arr, fn.data_ptr(), rope.data_ptr(), 64, ctypes.byref(h)))
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)
    pos, qreq, kvl, page, slot, indptr, indices = i32([5]), i32([0]), i32([6]), i32([0]), i32([5]), i32([0, 1]), i32([0])
    rows = N.Rows(pos.data_ptr(), qreq.data_ptr(), kvl.data_ptr(), page.data_ptr(), slot.data_ptr(), indptr.data_ptr(),
                  indices.data_ptr(), 1, 6, None, 0, 6, 5, 1)
    kv = torch.randn(5, 1, 2, 16, 8, 128, device=dev, dtype=torch.bfloat16)
    x = torch.randn(2, 1024, device=dev, dtype=torch.bfloat16)
    with torch.cuda.stream(st):
        def body():
            for _ in range(reps):
                N.check(L.vox_stack_forward(h, N.stream(), x.data_ptr(), None, kv.data_ptr(), kv[0].numel(), ctypes.byref(rows)))
        body(); st.synchronize()
        N.check(L.vox_graph_begin(ctx, N.stream())); body()
        g = ctypes.c_void_p(); N.check(L.vox_graph_end(ctx, N.stream(), ctypes.byref(g)))
        for _ in range(3): N.check(L.vox_graph_launch(g, N.stream()))
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): N.check(L.vox_graph_launch(g, N.stream()))
        e1.record(); st.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 5
    print(f"stack chain: {reps} x 5 layers: {us:8.1f} us total, {us/reps/5:6.2f} us/layer  (VOX_ABLATE={os.environ.get('VOX_ABLATE','0')}, VOX_DEV={os.environ.get('VOX_DEV','0')})")


for r in (3, 15, 30, 60):
    stack_chain(r)
