"""ORACLE (test infrastructure, never imported by the product path): CPU restatement (torch-CPU fp32) of the Qwen3-TTS 12 Hz
speech tokenizer's ENCODER — waveform -> 16 codebook ids per 80 ms frame — which `Qwen3TTSModel._encode_audio_to_codes`
(/root/reference/vox_serve/model/qwen3_tts.py:1330-1371) runs over the reference clip of an ICL voice-clone request.

The reference wires it as `Qwen3TTSTokenizerV2Model.encode` (/root/reference/vox_serve/tokenizer/qwen3_codec.py:1743-1773):
`Qwen3TTSTokenizerV2Encoder(MimiModel)` (:1669-1679) -> `.encode(...)` -> the first `encoder_valid_num_quantizers` (16) code
rows, trimmed to ceil(n_samples / 1920) frames.  MimiModel is third-party code: `transformers` (pyproject.toml pins
transformers; the installed wheel here is the one the goldens are generated with — its version is recorded in the fixture).
Its published algorithm, restated here (transformers/models/mimi/modeling_mimi.py):
  MimiConv1d            causal: left pad (k-1)d+1-stride, right pad up to a whole number of strides (zeros; "replicate" for
                        the downsample conv)
  MimiEncoder           conv k7, 4 x [ResnetBlock(ELU, conv k3, ELU, conv k1, + skip), ELU, conv k=2r stride r], ELU, conv k3
  MimiTransformerModel  8 pre-LayerNorm layers, rotate-half RoPE, causal attention over a sliding window, LayerScale, GELU MLP
  downsample            conv k4 stride 2, no bias, replicate padding
  MimiSplitResidualVectorQuantizer.encode   semantic RVQ (1 layer) + acoustic RVQ (15 of its 31 layers), each with its own 1x1
                        input projection; per layer: nearest centroid (embed_sum / clamp(cluster_usage, 1e-5)), residual -= centroid

Nearest-centroid search is a discrete decision on floating-point distances: two implementations that differ in the last bits
can pick different codes at a near tie.  The parity tests therefore assert equality of the codes on the fixtures (no tie within
float noise there) and, in general, that a differing code's distance is within 1e-5 relative of the minimum.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class CodecEncCfg:
    num_filters: int = 64
    ratios: List[int] = field(default_factory=lambda: [4, 5, 6, 8])      # reversed upsampling_ratios (encoder order)
    kernel_size: int = 7
    residual_kernel_size: int = 3
    last_kernel_size: int = 3
    compress: int = 2
    hidden_size: int = 512
    num_heads: int = 8
    head_dim: int = 64
    num_layers: int = 8
    intermediate_size: int = 2048
    rope_theta: float = 10000.0
    sliding_window: int = 250
    norm_eps: float = 1e-5
    codebook_size: int = 2048
    codebook_dim: int = 256
    num_quantizers: int = 32
    num_semantic_quantizers: int = 1
    valid_quantizers: int = 16

    @property
    def hop(self) -> int:
        return int(np.prod(self.ratios)) * 2


def tiny_codec_enc_cfg() -> CodecEncCfg:
    return CodecEncCfg(num_filters=64, ratios=[2, 3, 2, 2], hidden_size=64, num_heads=2, head_dim=32, num_layers=2, intermediate_size=128,
                       sliding_window=5, codebook_size=64, codebook_dim=32, num_quantizers=6, valid_quantizers=4)


def conv_names(c: CodecEncCfg):
    """[(state_dict prefix, cin, cout, k, stride, dilation)] of the SEANet encoder's convs, in order."""
    out = [("encoder.layers.0", 1, c.num_filters, c.kernel_size, 1, 1)]
    idx, ch = 1, c.num_filters
    for r in c.ratios:
        out.append((f"encoder.layers.{idx}.block.1", ch, ch // c.compress, c.residual_kernel_size, 1, 1))
        out.append((f"encoder.layers.{idx}.block.3", ch // c.compress, ch, 1, 1, 1))
        out.append((f"encoder.layers.{idx + 2}", ch, 2 * ch, 2 * r, r, 1))
        idx, ch = idx + 3, 2 * ch
    out.append((f"encoder.layers.{idx + 1}", ch, c.hidden_size, c.last_kernel_size, 1, 1))
    return out


def param_shapes(c: CodecEncCfg) -> Dict[str, tuple]:
    S: Dict[str, tuple] = {}
    for p, cin, cout, k, _, _ in conv_names(c):
        S[p + ".conv.weight"], S[p + ".conv.bias"] = (cout, cin, k), (cout,)
    H, A = c.hidden_size, c.num_heads * c.head_dim
    for l in range(c.num_layers):
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


def random_codec_enc_weights(c: CodecEncCfg, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    W = {}
    for k, shp in param_shapes(c).items():
        if k.endswith("cluster_usage"):
            w = torch.rand(shp, generator=g) * 4 + 0.5
        elif k.endswith("embed_sum"):
            w = torch.randn(shp, generator=g) * 1.5
        elif k.endswith("layernorm.weight"):
            w = 1 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            w = 0.05 * torch.randn(shp, generator=g)
        elif k.endswith(".scale"):
            w = 0.3 + 0.05 * torch.randn(shp, generator=g)
        else:
            fan = shp[1] * (shp[2] if len(shp) == 3 else 1)
            w = (torch.randn(shp, generator=g) * (1.4 / math.sqrt(fan))).to(torch.bfloat16)      # GEMM weights: bf16-valued, as served
        W[k] = w.float()
    return W


class CodecEncRef:
    def __init__(self, cfg: CodecEncCfg, W: Dict[str, torch.Tensor]):
        self.c, self.W = cfg, {k: v.float() for k, v in W.items()}

    def conv(self, x, p, k, stride=1, dil=1, mode="constant", bias=True):
        """MimiConv1d.forward (causal): x [1, C, L]."""
        w = self.W[p + ".conv.weight"]
        b = self.W[p + ".conv.bias"] if bias else None
        keff = (k - 1) * dil + 1
        pad_total = keff - stride
        L = x.shape[-1]
        n_frames = math.ceil((L - keff + pad_total) / stride + 1) - 1
        extra = n_frames * stride + keff - pad_total - L
        x = F.pad(x, (pad_total, extra), mode=mode)
        return F.conv1d(x, w, b, stride=stride, dilation=dil)

    def seanet(self, wav):
        """wav [N] -> [T25, hidden]"""
        c = self.c
        names = conv_names(c)
        x = self.conv(wav.view(1, 1, -1), names[0][0], c.kernel_size)
        i = 1
        for r in c.ratios:
            h = self.conv(F.elu(x), names[i][0], c.residual_kernel_size)
            x = x + self.conv(F.elu(h), names[i + 1][0], 1)
            x = self.conv(F.elu(x), names[i + 2][0], 2 * r, stride=r)
            i += 3
        x = self.conv(F.elu(x), names[i][0], c.last_kernel_size)
        return x[0].transpose(0, 1)

    def transformer(self, x):
        """x [T, H] -> [T, H]"""
        c, W = self.c, self.W
        T, Hd = x.shape
        nh, D = c.num_heads, c.head_dim
        inv = 1.0 / (c.rope_theta ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
        fr = torch.arange(T, dtype=torch.float32)[:, None] * inv[None]
        emb = torch.cat([fr, fr], -1)
        cos, sin = emb.cos()[None], emb.sin()[None]

        def rot(v):
            return torch.cat([-v[..., D // 2:], v[..., : D // 2]], -1)

        pos = torch.arange(T)
        delta = pos[:, None] - pos[None, :]
        mask = (delta >= 0) & (delta < c.sliding_window)
        for l in range(c.num_layers):
            p = f"encoder_transformer.layers.{l}."
            h = F.layer_norm(x, (Hd,), W[p + "input_layernorm.weight"], W[p + "input_layernorm.bias"], c.norm_eps)
            q = F.linear(h, W[p + "self_attn.q_proj.weight"]).view(T, nh, D).transpose(0, 1)
            k = F.linear(h, W[p + "self_attn.k_proj.weight"]).view(T, nh, D).transpose(0, 1)
            v = F.linear(h, W[p + "self_attn.v_proj.weight"]).view(T, nh, D).transpose(0, 1)
            q, k = q * cos + rot(q) * sin, k * cos + rot(k) * sin
            a = F.scaled_dot_product_attention(q[None], k[None], v[None], mask)[0]
            a = a.transpose(0, 1).reshape(T, nh * D)
            x = x + W[p + "self_attn_layer_scale.scale"] * F.linear(a, W[p + "self_attn.o_proj.weight"])
            h = F.layer_norm(x, (Hd,), W[p + "post_attention_layernorm.weight"], W[p + "post_attention_layernorm.bias"], c.norm_eps)
            x = x + W[p + "mlp_layer_scale.scale"] * F.linear(F.gelu(F.linear(h, W[p + "mlp.fc1.weight"])), W[p + "mlp.fc2.weight"])
        return x

    def latents(self, wav):
        """wav [N] -> pre-quantisation frames [T, hidden] at 12.5 Hz."""
        x = self.transformer(self.seanet(wav))
        return self.conv(x.transpose(0, 1)[None], "downsample", 4, stride=2, mode="replicate", bias=False)[0].transpose(0, 1)

    def codebook(self, name, i):
        p = f"quantizer.{name}_residual_vector_quantizer.layers.{i}.codebook."
        return self.W[p + "embed_sum"] / self.W[p + "cluster_usage"].clamp(min=1e-5)[:, None]

    def quantize(self, z, return_margins=False):
        """z [T, hidden] -> codes [T, valid_quantizers] (+ per code: (second-best - best) / best squared distance)."""
        c = self.c
        codes, margins = [], []
        for name, n in (("semantic", c.num_semantic_quantizers), ("acoustic", c.valid_quantizers - c.num_semantic_quantizers)):
            r = F.linear(z, self.W[f"quantizer.{name}_residual_vector_quantizer.input_proj.weight"][:, :, 0])
            for i in range(n):
                e = self.codebook(name, i)
                d = ((r[:, None, :] - e[None]) ** 2).sum(-1)
                idx = d.argmin(-1)
                two = d.topk(2, dim=-1, largest=False).values
                margins.append((two[:, 1] - two[:, 0]) / two[:, 0].clamp(min=1e-30))
                codes.append(idx)
                r = r - e[idx]
        out = torch.stack(codes, 1)
        return (out, torch.stack(margins, 1)) if return_margins else out

    def encode(self, wav: torch.Tensor) -> torch.Tensor:
        """wav [N] float32 (24 kHz) -> codes [ceil(N / hop), valid_quantizers] int64."""
        n = wav.numel()
        return self.quantize(self.latents(wav.float()))[: -(-n // self.c.hop)]
