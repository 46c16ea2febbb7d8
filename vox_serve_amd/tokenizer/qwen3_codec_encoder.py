"""Encoder of the Qwen3-TTS 12 Hz speech tokenizer on the native engine (vox_codecenc_*): reference clip -> codec ids for ICL voice cloning.

Mirrors `Qwen3TTSTokenizerV2Model.encode` of /root/reference/vox_serve/tokenizer/qwen3_codec.py:1743-1773 — the reference builds a
`transformers` MimiModel with the decoder half removed (`Qwen3TTSTokenizerV2Encoder`, :1669-1679) and keeps the first 16 of its 32
code rows — as called by `Qwen3TTSModel._encode_audio_to_codes` (model/qwen3_tts.py:1330-1371).  Weights keep the MimiModel state_dict
names (`encoder.` checkpoint prefix stripped): encoder.layers.*, encoder_transformer.layers.*, downsample.conv,
quantizer.{semantic,acoustic}_residual_vector_quantizer.*.
"""
import ctypes
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from .. import _native as N
from .base import device_bound
from .mimi import MimiLayerW
from .qwen3_codec import ConvW


@dataclass
class Qwen3TTSTokenizerV2EncoderConfig:
    """qwen3_codec.py:115-158 (the fields the encode path reads) + encoder_valid_num_quantizers of the tokenizer config."""
    num_filters: int = 64
    upsampling_ratios: List[int] = field(default_factory=lambda: [8, 6, 5, 4])     # the encoder strides are these reversed
    kernel_size: int = 7
    residual_kernel_size: int = 3
    last_kernel_size: int = 3
    compress: int = 2
    hidden_size: int = 512
    num_attention_heads: int = 8
    head_dim: int = 64
    num_hidden_layers: int = 8
    intermediate_size: int = 2048
    rope_theta: float = 10000.0
    sliding_window: int = 250
    norm_eps: float = 1e-5
    codebook_size: int = 2048
    codebook_dim: int = 256
    num_quantizers: int = 32
    num_semantic_quantizers: int = 1
    encoder_valid_num_quantizers: int = 16
    sampling_rate: int = 24000

    @property
    def encode_downsample_rate(self) -> int:
        r = 2
        for v in self.upsampling_ratios:
            r *= v
        return r


def param_shapes(c: Qwen3TTSTokenizerV2EncoderConfig) -> Dict[str, tuple]:
    """state_dict names and shapes of the encode path of transformers' MimiModel (encoder, encoder_transformer, downsample, quantizer)."""
    S: Dict[str, tuple] = {}

    def conv(p, cout, cin, k):
        S[p + ".conv.weight"], S[p + ".conv.bias"] = (cout, cin, k), (cout,)

    conv("encoder.layers.0", c.num_filters, 1, c.kernel_size)
    idx, ch = 1, c.num_filters
    for r in reversed(c.upsampling_ratios):
        conv(f"encoder.layers.{idx}.block.1", ch // c.compress, ch, c.residual_kernel_size)
        conv(f"encoder.layers.{idx}.block.3", ch, ch // c.compress, 1)
        conv(f"encoder.layers.{idx + 2}", 2 * ch, ch, 2 * r)
        idx, ch = idx + 3, 2 * ch
    conv(f"encoder.layers.{idx + 1}", c.hidden_size, ch, c.last_kernel_size)
    H, A = c.hidden_size, c.num_attention_heads * c.head_dim
    for l in range(c.num_hidden_layers):
        p = f"encoder_transformer.layers.{l}."
        for n in "qkv":
            S[p + f"self_attn.{n}_proj.weight"] = (A, H)
        S[p + "self_attn.o_proj.weight"] = (H, A)
        S[p + "mlp.fc1.weight"], S[p + "mlp.fc2.weight"] = (c.intermediate_size, H), (H, c.intermediate_size)
        for n in ("input_layernorm", "post_attention_layernorm"):
            S[p + n + ".weight"], S[p + n + ".bias"] = (H,), (H,)
        S[p + "self_attn_layer_scale.scale"], S[p + "mlp_layer_scale.scale"] = (H,), (H,)
    S["downsample.conv.weight"] = (H, H, 4)
    for name, n in (("semantic", c.num_semantic_quantizers), ("acoustic", c.num_quantizers - c.num_semantic_quantizers)):
        q = f"quantizer.{name}_residual_vector_quantizer."
        S[q + "input_proj.weight"] = (c.codebook_dim, H, 1)
        for i in range(n):
            S[q + f"layers.{i}.codebook.embed_sum"] = (c.codebook_size, c.codebook_dim)
            S[q + f"layers.{i}.codebook.cluster_usage"] = (c.codebook_size,)
    return S


def strided_taps(w: torch.Tensor, r: int) -> torch.Tensor:
    """Conv1d weight [Cout, Cin, 2r] of a causal stride-r conv -> two GEMM taps [2][Cout][r * Cin] over the input viewed as rows of r
    frames ([L / r][r * Cin], frame-major inside a row): output o = tap0 . row(o - 1) + tap1 . row(o), row(-1) = the causal zero padding."""
    cout, cin, k = w.shape
    if k != 2 * r:
        raise ValueError(f"stride-{r} conv with kernel {k}: the two-tap form needs kernel = 2 * stride")
    return torch.stack([w[:, :, j0:j0 + r].permute(0, 2, 1).reshape(cout, r * cin) for j0 in (0, r)], 0)


class StageW(ctypes.Structure):
    _fields_ = [("conv1", ConvW), ("conv2", ConvW), ("down", ConvW)]


class CodecEncWeights(ctypes.Structure):
    _fields_ = [("in_w", ctypes.c_void_p), ("in_b", ctypes.c_void_p), ("stage", StageW * 4), ("last", ConvW), ("layers", MimiLayerW * 16),
                ("inv_freq", ctypes.c_void_p), ("downsample", ConvW), ("sem_proj", ConvW), ("ac_proj", ConvW), ("sem_emb", ctypes.c_void_p),
                ("ac_emb", ctypes.c_void_p)]


class CodecEncConfigC(ctypes.Structure):
    _fields_ = [("num_filters", ctypes.c_int32), ("ratios", ctypes.c_int32 * 4)] + \
               [(n, ctypes.c_int32) for n in ("kernel_size", "residual_kernel_size", "last_kernel_size", "compress", "hidden", "num_heads",
                                              "head_dim", "num_layers", "ffn", "window", "codebook_size", "codebook_dim", "n_semantic",
                                              "n_acoustic")] + [("ln_eps", ctypes.c_float)]


def _bind(L):
    if getattr(L, "_cenc_bound", False):
        return
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.vox_codecenc_create.restype, L.vox_codecenc_create.argtypes = ci, [vp, ctypes.POINTER(CodecEncConfigC), ctypes.POINTER(CodecEncWeights),
                                                                        ci, ctypes.POINTER(vp)]
    L.vox_codecenc_destroy.restype, L.vox_codecenc_destroy.argtypes = None, [vp]
    L.vox_codecenc_encode.restype, L.vox_codecenc_encode.argtypes = ci, [vp, vp, vp, ci, vp, ctypes.POINTER(ctypes.c_int32), vp]
    L._cenc_bound = True


@device_bound
class Qwen3TTSTokenizerV2Encoder:
    def __init__(self, weights: Dict[str, torch.Tensor], config: Optional[Qwen3TTSTokenizerV2EncoderConfig] = None, device="cuda",
                 max_seconds: float = 30.0):
        self.cfg = c = config or Qwen3TTSTokenizerV2EncoderConfig()
        if len(c.upsampling_ratios) != 4:
            raise ValueError("Qwen3TTSTokenizerV2Encoder: four encoder stages")
        self.device = torch.device(device)
        self.L = N.lib()
        _bind(self.L)
        self._keep = []
        W, dev = weights, self.device
        ratios = list(reversed(c.upsampling_ratios))

        def f32(t):
            t = torch.as_tensor(t).detach().to(device=dev, dtype=torch.float32).contiguous()
            self._keep.append(t)
            return t.data_ptr()

        def conv(wp, bias=None):                  # wp [taps, N, Cin] -> one bf16 plane (the reference serves the tokenizer in bf16)
            pl = wp.float().to(torch.bfloat16).to(dev).contiguous()
            self._keep.append(pl)
            return ConvW(pl.data_ptr(), f32(bias) if bias is not None else None, pl.shape[0], pl.shape[1], pl.shape[2], 0)

        def causal(name):                         # Conv1d [Cout, Cin, k]: tap j reads x[t - (k-1) + j]
            return conv(W[name + ".conv.weight"].permute(2, 0, 1), W[name + ".conv.bias"])

        def strided(w, r, bias):                  # [Cout, Cin, 2r] -> two taps over rows of r frames: [2][Cout][r * Cin]
            return conv(strided_taps(w, r), bias)

        def lin(*names):
            return conv(torch.cat([W[n].float().reshape(W[n].shape[0], -1) for n in names], 0)[None])

        cw = CodecEncWeights()
        cw.in_w, cw.in_b = f32(W["encoder.layers.0.conv.weight"].reshape(c.num_filters, -1)), f32(W["encoder.layers.0.conv.bias"])
        idx = 1
        for s, r in enumerate(ratios):
            cw.stage[s].conv1, cw.stage[s].conv2 = causal(f"encoder.layers.{idx}.block.1"), causal(f"encoder.layers.{idx}.block.3")
            cw.stage[s].down = strided(W[f"encoder.layers.{idx + 2}.conv.weight"].float(), r, W[f"encoder.layers.{idx + 2}.conv.bias"])
            idx += 3
        cw.last = causal(f"encoder.layers.{idx + 1}")
        for l in range(c.num_hidden_layers):
            p = f"encoder_transformer.layers.{l}."
            lw = cw.layers[l]
            lw.ln1_w, lw.ln1_b = f32(W[p + "input_layernorm.weight"]), f32(W[p + "input_layernorm.bias"])
            lw.ln2_w, lw.ln2_b = f32(W[p + "post_attention_layernorm.weight"]), f32(W[p + "post_attention_layernorm.bias"])
            lw.scale1, lw.scale2 = f32(W[p + "self_attn_layer_scale.scale"]), f32(W[p + "mlp_layer_scale.scale"])
            lw.qkv = lin(p + "self_attn.q_proj.weight", p + "self_attn.k_proj.weight", p + "self_attn.v_proj.weight")
            lw.o, lw.fc1, lw.fc2 = lin(p + "self_attn.o_proj.weight"), lin(p + "mlp.fc1.weight"), lin(p + "mlp.fc2.weight")
        D = c.head_dim
        cw.inv_freq = f32(1.0 / (c.rope_theta ** (torch.arange(0, D, 2, dtype=torch.float32) / D)))     # MimiRotaryEmbedding
        cw.downsample = strided(W["downsample.conv.weight"].float(), 2, None)
        n_sem, n_ac = c.num_semantic_quantizers, c.encoder_valid_num_quantizers - c.num_semantic_quantizers

        def books(name, n):
            q = f"quantizer.{name}_residual_vector_quantizer.layers."
            return f32(torch.stack([W[q + f"{i}.codebook.embed_sum"].float() /
                                    W[q + f"{i}.codebook.cluster_usage"].float().clamp(min=1e-5)[:, None] for i in range(n)]))

        cw.sem_proj = lin("quantizer.semantic_residual_vector_quantizer.input_proj.weight")
        cw.ac_proj = lin("quantizer.acoustic_residual_vector_quantizer.input_proj.weight")
        cw.sem_emb, cw.ac_emb = books("semantic", n_sem), books("acoustic", max(n_ac, 1))
        cc = CodecEncConfigC(c.num_filters, (ctypes.c_int32 * 4)(*ratios), c.kernel_size, c.residual_kernel_size, c.last_kernel_size, c.compress,
                             c.hidden_size, c.num_attention_heads, c.head_dim, c.num_hidden_layers, c.intermediate_size, c.sliding_window,
                             c.codebook_size, c.codebook_dim, n_sem, n_ac, c.norm_eps)
        self.max_samples = int(max_seconds * c.sampling_rate)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            N.check(self.L.vox_codecenc_create(N.ctx(), ctypes.byref(cc), ctypes.byref(cw), self.max_samples, ctypes.byref(h)))
        self.h, self._cw = h, cw

    def encode(self, audio: torch.Tensor, return_latents: bool = False):
        """audio [N] float (24 kHz) -> codes [ceil(N / 1920), 16] int64 on the encoder's device (+ the frames that were quantised)."""
        c = self.cfg
        a = torch.as_tensor(audio).reshape(-1).to(self.device, torch.float32).contiguous()
        if a.numel() > self.max_samples:
            raise ValueError(f"reference clip of {a.numel()} samples exceeds the encoder capacity ({self.max_samples})")
        T = -(-a.numel() // c.encode_downsample_rate)
        codes = torch.empty(T + 1, c.encoder_valid_num_quantizers, dtype=torch.int32, device=self.device)
        lat = torch.empty(T + 1, c.hidden_size, dtype=torch.float32, device=self.device) if return_latents else None
        nf = ctypes.c_int32(0)
        with torch.cuda.device(self.device):
            N.check(self.L.vox_codecenc_encode(self.h, N.stream(), a.data_ptr(), a.numel(), codes.data_ptr(), ctypes.byref(nf),
                                               lat.data_ptr() if return_latents else None))
            torch.cuda.current_stream().synchronize()
        if nf.value != T:
            raise RuntimeError(f"codec encoder produced {nf.value} frames for {a.numel()} samples (expected {T})")
        out = codes[:T].long()
        return (out, lat[:T]) if return_latents else out

    __call__ = encode

    def close(self):
        if getattr(self, "h", None):
            self.L.vox_codecenc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
