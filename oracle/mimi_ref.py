"""CPU oracle for the Mimi codec decoder (token -> waveform) used by CSM, torch-CPU fp32.

TEST INFRASTRUCTURE ONLY (see oracle/voxref.c header).  Restates, module by module, the decode path of
/root/reference/vox_serve/tokenizer/mimi.py as the serving path runs it — STATELESS, every chunk from zero history:
  MimiModel.decode / decode_latent / _to_encoder_framerate            mimi.py:2993-3022
  SplitResidualVectorQuantizer.decode (first codebook + the rest, each through its own output_proj)   :719-872, 570-717
  EuclideanCodebook.embedding = embedding_sum / clamp(cluster_usage)                                 :92-298
  ConvTrUpsample1d (channel-wise transposed conv, kernel 4, stride 2, causal trim)                    :2272-2323
  ProjectedTransformer / StreamingTransformerLayer (LayerNorm, RoPE on interleaved pairs, causal SDPA, LayerScale,
      GELU MLP)                                                                                      :874-931, 1338-1895
  SEANetDecoder (conv k7, 4 x [ELU, transposed conv 2r/r, residual unit], ELU, conv k3), zero left padding,
      transposed convs trimmed on the right                                                          :2042-2216, 2548-2699
Weights use the reference's state_dict names.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class MimiCfg:
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

    @property
    def hop(self) -> int:
        return int(np.prod(self.ratios)) * self.upsample_stride


def tiny_mimi_cfg() -> MimiCfg:
    return MimiCfg(dim=64, n_filters=64, ratios=[4, 3, 2, 2], num_heads=2, num_layers=2, ffn=128, vq_dim=32, bins=64, n_q=6)


def param_shapes(c: MimiCfg) -> Dict[str, tuple]:
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
    mult = 2 ** len(c.ratios)
    ch = mult * c.n_filters
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


def random_mimi_weights(c: MimiCfg, seed=0) -> Dict[str, torch.Tensor]:
    """fp32 tensors holding bf16-representable values (the released checkpoint is bf16), scaled so that the waveform
    stays O(0.1): kaiming-like std per fan-in, layer scales 0.1, codebook entries N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    W = {}
    for k, s in param_shapes(c).items():
        if k.endswith("cluster_usage"):
            t = torch.rand(s, generator=g) * 3 + 0.5
        elif k.endswith("embedding_sum"):
            t = torch.randn(s, generator=g) * 2.0
        elif k.endswith(".scale"):
            t = torch.full(s, 0.1)
        elif "norm" in k and k.endswith("weight"):
            t = 1 + 0.1 * torch.randn(s, generator=g)
        elif k.endswith("bias"):
            t = 0.02 * torch.randn(s, generator=g)
        elif "convtr" in k and k.startswith("decoder"):
            t = torch.randn(s, generator=g) * math.sqrt(2.0 / (s[0] * 2))          # two taps reach each output sample
        elif k.startswith("upsample"):
            t = 0.7 + 0.2 * torch.randn(s, generator=g)
        else:
            fan_in = int(np.prod(s[1:]))
            t = torch.randn(s, generator=g) * math.sqrt(1.5 / fan_in)
        W[k] = t.to(torch.bfloat16).to(torch.float32)
    # bring the waveform to speech-like amplitude (rms ~0.2): rescale the last conv on a fixed probe input
    last = max((k for k in W if k.startswith("decoder.model.") and k.endswith(".conv.conv.weight")), key=lambda k: int(k.split(".")[2]))
    probe = torch.randint(0, c.bins, (1, c.n_q, 6), generator=g)
    rms = MimiRef(c, W).decode(probe).pow(2).mean().sqrt().item()
    for k in (last, last.replace("weight", "bias")):
        W[k] = (W[k] * (0.2 / max(rms, 1e-6))).to(torch.bfloat16).to(torch.float32)
    return W


class MimiRef:
    def __init__(self, cfg: MimiCfg, W: Dict[str, torch.Tensor]):
        self.c, self.W = cfg, {k: v.float() for k, v in W.items()}

    def codebook(self, name, i):
        p = f"quantizer.{name}.vq.layers.{i}._codebook."
        return self.W[p + "embedding_sum"] / self.W[p + "cluster_usage"].clamp(min=1e-5)[:, None]      # mimi.py:156-160

    @staticmethod
    def conv(x, w, b, dilation=1):
        """causal conv1d, zero history (StreamingConv1d.forward with a fresh state, pad_mode constant)"""
        k = w.shape[-1]
        return F.conv1d(F.pad(x, ((k - 1) * dilation, 0)), w, b, dilation=dilation)

    @staticmethod
    def convtr(x, w, b, stride, groups=1):
        k = w.shape[-1]
        y = F.conv_transpose1d(x, w, b, stride=stride, groups=groups)
        return y[..., : y.shape[-1] - (k - stride)]                                  # unpad1d(y, (0, K - S))

    def rope(self, q, k):
        """apply_rope (mimi.py:874-931) at offset 0: pairs (2i, 2i+1), freq_i = exp(-ln(max_period) * 2i / D)"""
        B, H, T, D = q.shape
        ds = torch.arange(D // 2, dtype=torch.float32)
        freqs = torch.exp(ds * (-math.log(self.c.max_period) * 2 / D))
        ang = torch.arange(T, dtype=torch.float32).view(1, 1, T, 1) * freqs
        cr, ci = torch.cos(ang), torch.sin(ang)

        def rot(x):
            x = x.view(B, H, T, D // 2, 2)
            xr, xi = x[..., 0], x[..., 1]
            return torch.stack([xr * cr - xi * ci, xr * ci + xi * cr], -1).view(B, H, T, D)
        return rot(q), rot(k)

    def transformer(self, x):
        c, W = self.c, self.W
        B, T, C = x.shape
        H, D = c.num_heads, c.dim // c.num_heads
        for l in range(c.num_layers):
            p = f"decoder_transformer.transformer.layers.{l}."
            h = F.layer_norm(x, (C,), W[p + "norm1.weight"], W[p + "norm1.bias"], 1e-5)
            qkv = F.linear(h, W[p + "self_attn.in_projs.0.weight"]).view(B, T, 3, H, D).permute(2, 0, 3, 1, 4)
            q, k = self.rope(qkv[0], qkv[1])
            pos = torch.arange(T)
            delta = pos.view(-1, 1) - pos.view(1, -1)
            mask = (delta >= 0) & (delta < c.context)
            a = F.scaled_dot_product_attention(q, k, qkv[2], mask)
            a = a.permute(0, 2, 1, 3).reshape(B, T, C)
            x = x + W[p + "layer_scale_1.scale"] * F.linear(a, W[p + "self_attn.out_projs.0.weight"])
            h = F.layer_norm(x, (C,), W[p + "norm2.weight"], W[p + "norm2.bias"], 1e-5)
            x = x + W[p + "layer_scale_2.scale"] * F.linear(F.gelu(F.linear(h, W[p + "linear1.weight"])), W[p + "linear2.weight"])
        return x

    def decode(self, codes: torch.Tensor) -> torch.Tensor:
        """codes [B, n_q, T] int -> wav [B, 1, T*hop] fp32"""
        c, W = self.c, self.W
        codes = codes.long().clamp(0, c.bins - 1)
        q0 = F.embedding(codes[:, 0], self.codebook("rvq_first", 0))                  # [B,T,vq]
        qr = torch.zeros_like(q0)
        for i in range(c.n_q - 1):
            qr = qr + F.embedding(codes[:, i + 1], self.codebook("rvq_rest", i))
        emb = F.conv1d(q0.transpose(1, 2), W["quantizer.rvq_first.output_proj.weight"]) + \
            F.conv1d(qr.transpose(1, 2), W["quantizer.rvq_rest.output_proj.weight"])  # [B,dim,T]
        emb = self.convtr(emb, W["upsample.convtr.convtr.convtr.weight"], None, c.upsample_stride, groups=c.dim)
        emb = self.transformer(emb.transpose(1, 2)).transpose(1, 2)
        x = self.conv(emb, W["decoder.model.0.conv.conv.weight"], W["decoder.model.0.conv.conv.bias"])
        idx = 1
        for r in c.ratios:
            x = self.convtr(F.elu(x), W[f"decoder.model.{idx + 1}.convtr.convtr.weight"], W[f"decoder.model.{idx + 1}.convtr.convtr.bias"], r)
            p = f"decoder.model.{idx + 2}.block."
            v = self.conv(F.elu(x), W[p + "1.conv.conv.weight"], W[p + "1.conv.conv.bias"])
            x = x + self.conv(F.elu(v), W[p + "3.conv.conv.weight"], W[p + "3.conv.conv.bias"])
            idx += 3
        return self.conv(F.elu(x), W[f"decoder.model.{idx + 1}.conv.conv.weight"], W[f"decoder.model.{idx + 1}.conv.conv.bias"])
