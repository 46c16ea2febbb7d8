"""Development aid: the depth transformer's layer stack alone (vox_stack_forward, 5 layers at depth widths, one row, the depth-loop decode
hints) as a hipGraph of 14 dependent passes over the 14 visible-token counts of a frame — the depth loop without heads / samplers —
timed per pass and per stage.  With the dev-knob library (VOX_LIB=tools/bin/libvoxhip_dev.so) VOX_ABLATE=257 removes the attention
(o_proj stays a plain GEMV): 20 GEMV stages per pass."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vox_serve_amd import _native as N
from vox_serve_amd.engine import StackCfg, _stack_config, rope_table

dev = torch.device("cuda")
L, ctx = N.lib(), N.ctx()
if "--with-engine" in sys.argv:      # a whole Qwen3-TTS engine (3.5 GB of weights, KV pages, workspaces) alive in the process, unused
    from vox_serve_amd.engine import Qwen3Cfg, Qwen3Engine
    from vox_serve_amd.synth import synth_qwen3_weights
    _cfg = Qwen3Cfg()
    _W = synth_qwen3_weights(_cfg, dev, seed=0)
    _eng = Qwen3Engine(_cfg, _W, max_batch=1, page_size=128, max_pages=64, max_seq_len=2304, max_prefill_rows=128)
    sys.argv.remove("--with-engine")
ENGW = "--engine-weights" in sys.argv       # the chain reads the ENGINE's depth-stack weight tensors (needs --with-engine)
if ENGW: sys.argv.remove("--engine-weights")
H, NL, heads, kvh, D, F, G = 1024, 5, 16, 8, 128, 3072, 16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4          # distinct weight sets cycled (footprint = reps x 155 MB)
ec = StackCfg(H, NL, heads, kvh, D, F, 1e-6, 1e6, 1.0, None, False, None, True, False)
g = torch.Generator(device=dev).manual_seed(0)
w = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.02).to(torch.bfloat16)
ones = lambda n: torch.ones(n, device=dev, dtype=torch.bfloat16)
stacks, keep = [], []
rope = rope_table(64, ec, dev)
for r in range(reps):
    arr = (N.LayerWeights * NL)()
    for l in range(NL):
        ts = dict(wqkv=w((heads + 2 * kvh) * D, H), wo=w(H, heads * D), wgate=w(F, H), wup=w(F, H), wdown=w(H, F), ln1=ones(H), ln2=ones(H),
                  qnorm=ones(D), knorm=ones(D))
        for k, v in ts.items():
            keep.append(v)
            setattr(arr[l], k, v.data_ptr())
    fn = ones(H)
    if ENGW: arr = _eng.dl
    keep += [arr, fn]
    sc = _stack_config(ec, G, 2, G)
    h = ctypes.c_void_p()
    N.check(L.vox_stack_create(ctx, ctypes.byref(sc), arr, fn.data_ptr(), rope.data_ptr(), 64, ctypes.byref(h)))
    stacks.append(h)
kv = torch.zeros(NL, 1, 2, G, kvh, D, dtype=torch.bfloat16, device=dev)
x = (torch.randn(1, H, generator=g, device=dev) * 0.5).to(torch.bfloat16)
iota = torch.arange(8, dtype=torch.int32, device=dev)
indptr = torch.tensor([0, 1], dtype=torch.int32, device=dev)
pos = [torch.full((1,), i, dtype=torch.int32, device=dev) for i in range(G)]
kvl = [torch.full((1,), i + 1, dtype=torch.int32, device=dev) for i in range(G)]

def one_pass(i, h):
    rows = N.Rows(pos[i].data_ptr(), iota.data_ptr(), kvl[i].data_ptr(), iota.data_ptr(), pos[i].data_ptr(), indptr.data_ptr(), iota.data_ptr(),
                  1, i + 1, None, 0, i + 1, i, 1)
    N.check(L.vox_stack_forward(h, N.stream(), x.data_ptr(), None, kv.data_ptr(), kv[0].numel(), ctypes.byref(rows)))

st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for i in range(2, G):
        one_pass(i, stacks[0])             # warm-up (kernel attributes)
    st.synchronize()
    N.check(L.vox_graph_begin(ctx, N.stream()))
    n = 0
    for rep in range(reps):
        for i in range(2, G):
            one_pass(i, stacks[rep]); n += 1
    gh = ctypes.c_void_p()
    N.check(L.vox_graph_end(ctx, N.stream(), ctypes.byref(gh)))
    for _ in range(3):
        N.check(L.vox_graph_launch(gh, N.stream()))
    st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        N.check(L.vox_graph_launch(gh, N.stream()))
    e1.record()
    st.synchronize()
us = e0.elapsed_time(e1) * 1e3 / (10 * n)
print(f"ABLATE={os.environ.get('VOX_ABLATE', '0')} weight sets {reps}: {us:.2f} us per 5-layer pass = {us / 20:.2f} us per stage (4 stages per layer)")
