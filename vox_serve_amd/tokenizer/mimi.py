"""Mimi codec decoder on libvoxhip (drop-in surface of MimiDecoder, /root/reference/vox_serve/tokenizer/mimi.py:3024-3089:
`decode(codes[B, K, T]) -> audio[B, 1, T*1920]`).  Stateless per chunk, like the reference's serving path
(model/csm.py:772-787: "TODO: caching for mimi").  Weights are the reference checkpoint's decoder-side state_dict names
(quantizer.*, upsample.*, decoder_transformer.*, decoder.*); packing is layout only:
  * codebooks: embedding_sum / clamp(cluster_usage, 1e-5)           (EuclideanCodebook.embedding, mimi.py:156-160)
  * Conv1d [Cout,Cin,K]      -> taps k=0..K-1 with look-back K-1-k    (causal, zero history: mimi.py:2115-2148)
  * ConvTranspose1d [Cin,Cout,2r] -> two taps writing r*Cout values per input row, right trim implicit (mimi.py:2190-2197)
"""
import ctypes
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from .. import _native as N
from .base import DecoderCache, device_bound
from .qwen3_codec import ConvW


@dataclass
class MimiConfig:
    """_mimi_config of the reference (mimi.py:20-71)"""
    dim: int = 512
    n_filters: int = 64
    ratios: List[int] = field(default_factory=lambda: [8, 6, 5, 4])
    kernel_size: int = 7
    residual_kernel_size: int = 3
    last_kernel_size: int = 3
    compress: int = 2
    num_heads: int = 8
    num_layers: int = 8
    ffn: int = 2048
    max_period: float = 10000.0
    context: int = 250
    vq_dim: int = 256
    bins: int = 2048
    n_q: int = 32
    upsample_stride: int = 2
    sample_rate: int = 24000

    @property
    def hop(self) -> int:
        return int(math.prod(self.ratios)) * self.upsample_stride


def param_shapes(c: MimiConfig) -> Dict[str, tuple]:
    """decoder-side state_dict names -> shapes of the Mimi checkpoint (mimi.py:20-71, 2548-2699, 1841-1895, 719-872)"""
    S = {}
    for name, nq in (("rvq_first", 1), ("rvq_rest", c.n_q - 1)):
        for i in range(nq):
            p = f"quantizer.{name}.vq.layers.{i}._codebook."
            S[p + "embedding_sum"], S[p + "cluster_usage"] = (c.bins, c.vq_dim), (c.bins,)
        S[f"quantizer.{name}.output_proj.weight"] = (c.dim, c.vq_dim, 1)
    S["upsample.convtr.convtr.convtr.weight"] = (c.dim, 1, 2 * c.upsample_stride)
    for l in range(c.num_layers):
        p = f"decoder_transformer.transformer.layers.{l}."
        S[p + "self_attn.in_projs.0.weight"], S[p + "self_attn.out_projs.0.weight"] = (3 * c.dim, c.dim), (c.dim, c.dim)
        for n in ("norm1", "norm2"):
            S[p + n + ".weight"], S[p + n + ".bias"] = (c.dim,), (c.dim,)
        S[p + "linear1.weight"], S[p + "linear2.weight"] = (c.ffn, c.dim), (c.dim, c.ffn)
        S[p + "layer_scale_1.scale"], S[p + "layer_scale_2.scale"] = (c.dim,), (c.dim,)
    ch = 2 ** len(c.ratios) * c.n_filters
    S["decoder.model.0.conv.conv.weight"], S["decoder.model.0.conv.conv.bias"] = (ch, c.dim, c.kernel_size), (ch,)
    idx = 1
    for r in c.ratios:
        S[f"decoder.model.{idx + 1}.convtr.convtr.weight"], S[f"decoder.model.{idx + 1}.convtr.convtr.bias"] = (ch, ch // 2, 2 * r), (ch // 2,)
        hid = ch // 2 // c.compress
        p = f"decoder.model.{idx + 2}.block."
        S[p + "1.conv.conv.weight"], S[p + "1.conv.conv.bias"] = (hid, ch // 2, c.residual_kernel_size), (hid,)
        S[p + "3.conv.conv.weight"], S[p + "3.conv.conv.bias"] = (ch // 2, hid, 1), (ch // 2,)
        idx += 3
        ch //= 2
    S[f"decoder.model.{idx + 1}.conv.conv.weight"], S[f"decoder.model.{idx + 1}.conv.conv.bias"] = (1, ch, c.last_kernel_size), (1,)
    return S


class MimiLayerW(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("ln1_w", "ln1_b", "ln2_w", "ln2_b", "scale1", "scale2")] + \
               [("qkv", ConvW), ("o", ConvW), ("fc1", ConvW), ("fc2", ConvW)]


class MimiBlockW(ctypes.Structure):
    _fields_ = [("tconv", ConvW), ("conv1", ConvW), ("conv2", ConvW)]


class MimiWeights(ctypes.Structure):
    _fields_ = [("emb", ctypes.c_void_p), ("rvq_first_out", ConvW), ("rvq_rest_out", ConvW), ("up_w", ctypes.c_void_p),
                ("layers", MimiLayerW * 16), ("dec0", ConvW), ("blocks", MimiBlockW * 4), ("final_w", ctypes.c_void_p),
                ("final_b", ctypes.c_float)]


class MimiConfigC(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("bins", "vq_dim", "dim", "num_heads", "num_layers", "ffn", "n_q", "n_filters")] + \
               [("ratios", ctypes.c_int32 * 4)] + \
               [(n, ctypes.c_int32) for n in ("kernel_size", "last_kernel_size", "context")] + \
               [("max_period", ctypes.c_float), ("ln_eps", ctypes.c_float)]


def _bind(L):
    if getattr(L, "_mimi_bound", False):
        return
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.vox_mimi_create.restype, L.vox_mimi_create.argtypes = ci, [vp, ctypes.POINTER(MimiConfigC), ctypes.POINTER(MimiWeights), ci, ci,
                                                                ctypes.POINTER(vp)]
    L.vox_mimi_destroy.restype, L.vox_mimi_destroy.argtypes = None, [vp]
    L.vox_mimi_decode.restype, L.vox_mimi_decode.argtypes = ci, [vp, vp, vp, ci, ci, ci, vp]
    L.vox_mimi_stream_enable.restype, L.vox_mimi_stream_enable.argtypes = ci, [vp, ci]
    L.vox_mimi_reset_slot.restype, L.vox_mimi_reset_slot.argtypes = ci, [vp, vp, ci]
    L.vox_mimi_decode_chunk.restype, L.vox_mimi_decode_chunk.argtypes = ci, [vp, vp, vp, ci, vp, ci, ci, vp]
    L._mimi_bound = True


@dataclass
class MimiDecoderCache(DecoderCache):
    """Handle on the in-place streaming state of one request (stateful option): `slot` indexes the engine's state arrays.
    A [1]-shaped int32 tensor so that DecoderCache.cat / __getitem__ keep their meaning."""
    slot: Optional[torch.Tensor] = None


@device_bound
class MimiDecoder:
    def __init__(self, weights: Dict[str, torch.Tensor], config: Optional[MimiConfig] = None, num_codebooks: Optional[int] = None,
                 device="cuda", max_batch=8, max_frames=10):
        self.cfg = c = config or MimiConfig()
        if len(c.ratios) != 4 or c.upsample_stride != 2:
            raise ValueError("MimiDecoder: the native engine expects 4 SEANet stages and the x2 upsampler")
        self.num_codebooks = num_codebooks or c.n_q
        self.device = torch.device(device)
        self.max_batch, self.max_frames = max_batch, max_frames
        self.L = N.lib()
        _bind(self.L)
        self._keep = []
        W, dev = weights, self.device

        def f32(t):
            t = t.detach().to(device=dev, dtype=torch.float32).contiguous()
            self._keep.append(t)
            return t.data_ptr()

        def conv(wp, bias=None, bias_mod=0):      # wp [taps, N, Cin]
            wp = wp.detach().to(device=dev, dtype=torch.bfloat16).contiguous()
            self._keep.append(wp)
            return ConvW(wp.data_ptr(), f32(bias) if bias is not None else None, wp.shape[0], wp.shape[1], wp.shape[2], bias_mod)

        def lin(name):
            return conv(W[name].float().reshape(W[name].shape[0], -1)[None])

        def causal(name):
            return conv(W[name + ".weight"].float().permute(2, 0, 1), W[name + ".bias"])

        def tconv(name, stride):
            w = W[name + ".weight"].float()
            cin, cout, k = w.shape
            taps = [w[:, :, j0:j0 + stride].permute(2, 1, 0).reshape(stride * cout, cin) for j0 in range(0, k, stride)]
            return conv(torch.stack(taps, 0), W[name + ".bias"], bias_mod=cout)

        books = []
        for name, nq in (("rvq_first", 1), ("rvq_rest", c.n_q - 1)):
            for i in range(nq):
                p = f"quantizer.{name}.vq.layers.{i}._codebook."
                books.append(W[p + "embedding_sum"].float() / W[p + "cluster_usage"].float().clamp(min=1e-5)[:, None])
        mw = MimiWeights()
        mw.emb = f32(torch.stack(books, 0))
        mw.rvq_first_out, mw.rvq_rest_out = lin("quantizer.rvq_first.output_proj.weight"), lin("quantizer.rvq_rest.output_proj.weight")
        mw.up_w = f32(W["upsample.convtr.convtr.convtr.weight"].float().reshape(c.dim, 4))
        for l in range(c.num_layers):
            p = f"decoder_transformer.transformer.layers.{l}."
            lw = mw.layers[l]
            lw.ln1_w, lw.ln1_b, lw.ln2_w, lw.ln2_b = (f32(W[p + "norm1.weight"]), f32(W[p + "norm1.bias"]), f32(W[p + "norm2.weight"]),
                                                      f32(W[p + "norm2.bias"]))
            lw.scale1, lw.scale2 = f32(W[p + "layer_scale_1.scale"]), f32(W[p + "layer_scale_2.scale"])
            lw.qkv, lw.o = lin(p + "self_attn.in_projs.0.weight"), lin(p + "self_attn.out_projs.0.weight")
            lw.fc1, lw.fc2 = lin(p + "linear1.weight"), lin(p + "linear2.weight")
        mw.dec0 = causal("decoder.model.0.conv.conv")
        idx = 1
        for b, r in enumerate(c.ratios):
            bw = mw.blocks[b]
            bw.tconv = tconv(f"decoder.model.{idx + 1}.convtr.convtr", r)
            bw.conv1 = causal(f"decoder.model.{idx + 2}.block.1.conv.conv")
            bw.conv2 = causal(f"decoder.model.{idx + 2}.block.3.conv.conv")
            idx += 3
        last = f"decoder.model.{idx + 1}.conv.conv"
        mw.final_w = f32(W[last + ".weight"].float()[0])                 # [C][K]
        mw.final_b = float(W[last + ".bias"].float().item())
        mc = MimiConfigC(c.bins, c.vq_dim, c.dim, c.num_heads, c.num_layers, c.ffn, c.n_q, c.n_filters, (ctypes.c_int32 * 4)(*c.ratios),
                         c.kernel_size, c.last_kernel_size, c.context, c.max_period, 1e-5)
        h = ctypes.c_void_p()
        N.check(self.L.vox_mimi_create(N.ctx(), ctypes.byref(mc), ctypes.byref(mw), max_batch, max_frames, ctypes.byref(h)))
        self.h, self._mw = h, mw
        self._graphs, self.use_graph = {}, True

    sample_rate = property(lambda self: self.cfg.sample_rate)
    hop = property(lambda self: self.cfg.hop)

    def decode(self, codes: torch.Tensor, code_layout: str = "BQT") -> torch.Tensor:
        """codes [B, K, T] (reference layout) or, with code_layout="BTQ", [B, T, >=K] -> audio fp32 [B, 1, T*hop]"""
        if code_layout == "BQT":
            codes = codes.transpose(1, 2)
        codes = codes.to(device=self.device, dtype=torch.int32).contiguous()
        b, t, stride = codes.shape
        if stride < self.cfg.n_q:
            raise ValueError(f"Expected {self.cfg.n_q} codebooks, got {stride}")
        if self.use_graph and b <= self.max_batch:
            return self._decode_graph(codes, b, t, stride)
        out = torch.empty(b, 1, t * self.hop, dtype=torch.float32, device=self.device)
        for b0 in range(0, b, self.max_batch):
            nb = min(self.max_batch, b - b0)
            N.check(self.L.vox_mimi_decode(self.h, N.stream(), codes[b0:b0 + nb].data_ptr(), stride, nb, t, out[b0:b0 + nb].data_ptr()))
        return out

    def _decode_graph(self, codes, b, t, stride):
        """One hipGraph per (rows, frames, code stride) on the decoder's own stream (the stateless chunk is a pure function of
        the codes): one host launch instead of ~60.  First call of a shape eager, second captured, then replayed.  The
        returned tensor is a fresh copy-free view valid until the next call of the same shape."""
        key = (b, t, stride)
        ent = self._graphs.get(key)
        if ent is None:
            ent = self._graphs[key] = {"codes": torch.empty(b, t, stride, dtype=torch.int32, device=self.device),
                                       "out": torch.empty(b, 1, t * self.hop, dtype=torch.float32, device=self.device), "g": None, "calls": 0}
        # on the caller's current stream (N.graph_capture: why the decoder owns none)
        ent["codes"].copy_(codes, non_blocking=True)
        args = lambda: (self.h, N.stream(), ent["codes"].data_ptr(), stride, b, t, ent["out"].data_ptr())
        ent["calls"] += 1
        if ent["calls"] == 1:
            N.check(self.L.vox_mimi_decode(*args()))
        else:
            if ent["g"] is None:
                with N.graph_capture() as cap:
                    N.check(self.L.vox_mimi_decode(*args()))
                ent["g"] = cap.graph
            N.check(self.L.vox_graph_launch(ent["g"], N.stream()))
        return ent["out"]

    # ---- streaming option (not the reference's behaviour: it decodes every chunk from a fresh state, mimi.py:3085-3089) ----
    def enable_streaming(self, max_slots: int):
        """Allocate per-request decoder state for `max_slots` concurrent requests (conv look-back rows + a K/V ring per
        transformer layer): chunks decoded with `decode_chunk` then continue seamlessly from the previous chunk."""
        N.check(self.L.vox_mimi_stream_enable(self.h, int(max_slots)))
        self.max_slots = int(max_slots)
        self._free_slots = list(range(self.max_slots))

    def init_cache(self, batch_size: int = 1, **_) -> MimiDecoderCache:
        return MimiDecoderCache(slot=torch.tensor([self.alloc_slot() for _ in range(batch_size)], dtype=torch.int32, device=self.device))

    def release_cache(self, cache: MimiDecoderCache):
        for s_ in cache.slot.tolist():
            self.free_slot(s_)

    def alloc_slot(self) -> int:
        if not self._free_slots:
            raise RuntimeError("MimiDecoder: no free streaming slot")
        slot = self._free_slots.pop(0)
        N.check(self.L.vox_mimi_reset_slot(self.h, N.stream(), slot))
        return slot

    def free_slot(self, slot: int):
        self._free_slots.append(int(slot))

    def decode_chunk(self, codes: torch.Tensor, slots, code_layout: str = "BQT") -> torch.Tensor:
        """Streaming decode of one chunk per request: codes as in `decode`, slots = one state slot per row."""
        if code_layout == "BQT":
            codes = codes.transpose(1, 2)
        codes = codes.to(device=self.device, dtype=torch.int32).contiguous()
        b, t, stride = codes.shape
        if stride < self.cfg.n_q:
            raise ValueError(f"Expected {self.cfg.n_q} codebooks, got {stride}")
        if len(slots) != b or len(set(int(x) for x in slots)) != b:
            raise ValueError("decode_chunk: one distinct slot per row")
        sl = torch.tensor([int(x) for x in slots], dtype=torch.int32, device=self.device)
        out = torch.empty(b, 1, t * self.hop, dtype=torch.float32, device=self.device)
        for b0 in range(0, b, self.max_batch):
            nb = min(self.max_batch, b - b0)
            N.check(self.L.vox_mimi_decode_chunk(self.h, N.stream(), codes[b0:b0 + nb].data_ptr(), stride, sl[b0:b0 + nb].data_ptr(),
                                                 nb, t, out[b0:b0 + nb].data_ptr()))
        return out

    def close(self):
        for ent in self._graphs.values():
            if ent["g"] is not None:
                self.L.vox_graph_destroy(ent["g"])
        self._graphs.clear()
        if self.h:
            self.L.vox_mimi_destroy(self.h)
            self.h = None
