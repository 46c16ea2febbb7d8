"""CPU oracle of the Qwen3-TTS 12 Hz codec decoder's STREAMING path (token -> waveform).

TEST INFRASTRUCTURE ONLY (see oracle/voxref.c header).  Restates, op by op,
  Qwen3TTSTokenizerV2Decoder.forward_chunk     /root/reference/vox_serve/tokenizer/qwen3_codec.py:1541-1666
  SplitResidualVectorQuantizer.decode          qwen3_codec.py:1298-1304, 1159-1162, 1204-1210
  CausalConvNet.forward_chunk                  qwen3_codec.py:274-340   (left state = last `padding` inputs)
  CausalTransConvNet.forward_chunk / forward   qwen3_codec.py:353-397   (1-sample input state, right trim)
  ConvNeXtBlock.forward_chunk                  qwen3_codec.py:434-468
  DecoderAttention.forward_chunk               qwen3_codec.py:573-685   (72-slot window, zero slots NOT masked: quirk Q4)
  DecoderTransformerModel.forward_chunk        qwen3_codec.py:914-977
  SnakeBeta / ResidualUnit / DecoderBlock      qwen3_codec.py:1004-1141
Weights use the reference's state_dict names.

Numeric mode ("mixed", the one the HIP path implements): activations fp32 end to end; every matmul / conv
operand (activation AND weight) is rounded to bf16 at the contraction input, products accumulate in fp32;
the attention KV window is stored in bf16 like the reference's cache.  The reference itself runs the codec
with every tensor in bf16 (model/qwen3_tts.py:1061-1064); tests state the tolerance against both.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List

import torch
import torch.nn.functional as F


@dataclass
class CodecCfg:
    codebook_size: int = 2048
    codebook_dim: int = 512          # quantizer in/out dimension; codebook vectors have codebook_dim // 2
    latent_dim: int = 1024
    decoder_dim: int = 1536
    hidden_size: int = 512
    intermediate_size: int = 1024
    head_dim: int = 64
    num_heads: int = 16
    num_layers: int = 8
    num_quantizers: int = 16
    rms_eps: float = 1e-5
    rope_theta: float = 10000.0
    sliding_window: int = 72
    upsample_rates: List[int] = field(default_factory=lambda: [8, 5, 4, 3])
    upsampling_ratios: List[int] = field(default_factory=lambda: [2, 2])

    @property
    def total_upsample(self):
        return int(math.prod(self.upsample_rates + self.upsampling_ratios))


def tiny_codec_cfg():
    return CodecCfg(codebook_size=64, codebook_dim=64, latent_dim=64, decoder_dim=128, hidden_size=32,
                    intermediate_size=64, head_dim=16, num_heads=4, num_layers=2, num_quantizers=4,
                    sliding_window=12, upsample_rates=[4, 2, 2, 2], upsampling_ratios=[2, 2])


def param_shapes(c: CodecCfg) -> Dict[str, tuple]:
    """Reference state_dict names -> shapes (checked against the reference module in make_goldens.py)."""
    S = {}
    H, L, I = c.hidden_size, c.latent_dim, c.intermediate_size
    qd = c.num_heads * c.head_dim
    for i in range(c.num_layers):
        p = f"pre_transformer.layers.{i}."
        S[p + "self_attn.q_proj.weight"] = (qd, H)
        S[p + "self_attn.k_proj.weight"] = (qd, H)
        S[p + "self_attn.v_proj.weight"] = (qd, H)
        S[p + "self_attn.o_proj.weight"] = (H, qd)
        S[p + "mlp.gate_proj.weight"] = (I, H)
        S[p + "mlp.up_proj.weight"] = (I, H)
        S[p + "mlp.down_proj.weight"] = (H, I)
        S[p + "input_layernorm.weight"] = (H,)
        S[p + "post_attention_layernorm.weight"] = (H,)
        S[p + "self_attn_layer_scale.scale"] = (H,)
        S[p + "mlp_layer_scale.scale"] = (H,)
    S["pre_transformer.norm.weight"] = (H,)
    S["pre_transformer.input_proj.weight"] = (H, L)
    S["pre_transformer.input_proj.bias"] = (H,)
    S["pre_transformer.output_proj.weight"] = (L, H)
    S["pre_transformer.output_proj.bias"] = (L,)
    vq = c.codebook_dim // 2
    for name, n in (("rvq_first", 1), ("rvq_rest", c.num_quantizers - 1)):
        S[f"quantizer.{name}.input_proj.weight"] = (vq, c.codebook_dim, 1)
        S[f"quantizer.{name}.output_proj.weight"] = (c.codebook_dim, vq, 1)
        for j in range(n):
            S[f"quantizer.{name}.vq.layers.{j}._codebook.cluster_usage"] = (c.codebook_size,)
            S[f"quantizer.{name}.vq.layers.{j}._codebook.embedding_sum"] = (c.codebook_size, vq)
    S["pre_conv.conv.weight"] = (L, c.codebook_dim, 3)
    S["pre_conv.conv.bias"] = (L,)
    for u, f_ in enumerate(c.upsampling_ratios):
        p = f"upsample.{u}."
        S[p + "0.conv.weight"] = (L, L, f_)
        S[p + "0.conv.bias"] = (L,)
        S[p + "1.gamma"] = (L,)
        S[p + "1.dwconv.conv.weight"] = (L, 1, 7)
        S[p + "1.dwconv.conv.bias"] = (L,)
        S[p + "1.norm.weight"] = (L,)
        S[p + "1.norm.bias"] = (L,)
        S[p + "1.pwconv1.weight"] = (4 * L, L)
        S[p + "1.pwconv1.bias"] = (4 * L,)
        S[p + "1.pwconv2.weight"] = (L, 4 * L)
        S[p + "1.pwconv2.bias"] = (L,)
    S["decoder.0.conv.weight"] = (c.decoder_dim, L, 7)
    S["decoder.0.conv.bias"] = (c.decoder_dim,)
    for b, r in enumerate(c.upsample_rates):
        cin, cout = c.decoder_dim // 2 ** b, c.decoder_dim // 2 ** (b + 1)
        p = f"decoder.{b + 1}.block."
        S[p + "0.alpha"] = (cin,)
        S[p + "0.beta"] = (cin,)
        S[p + "1.conv.weight"] = (cin, cout, 2 * r)
        S[p + "1.conv.bias"] = (cout,)
        for u in range(3):
            q = f"{p}{u + 2}."
            for a in ("act1", "act2"):
                S[q + a + ".alpha"] = (cout,)
                S[q + a + ".beta"] = (cout,)
            S[q + "conv1.conv.weight"] = (cout, cout, 7)
            S[q + "conv1.conv.bias"] = (cout,)
            S[q + "conv2.conv.weight"] = (cout, cout, 1)
            S[q + "conv2.conv.bias"] = (cout,)
    nb = len(c.upsample_rates)
    cl = c.decoder_dim // 2 ** nb
    S[f"decoder.{nb + 1}.alpha"] = (cl,)
    S[f"decoder.{nb + 1}.beta"] = (cl,)
    S[f"decoder.{nb + 2}.conv.weight"] = (1, cl, 7)
    S[f"decoder.{nb + 2}.conv.bias"] = (1,)
    return S


def random_codec_weights(c: CodecCfg, seed=0) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights (fp32 tensors holding bf16-representable values).  Scales chosen so the
    waveform is O(0.1) and nothing saturates: fan-in scaled normals, codebooks N(0,1), norm weights ~1,
    SnakeBeta alpha/beta small, layer scales 0.05, ConvNeXt gamma 0.1 (real checkpoints: learnt)."""
    g = torch.Generator().manual_seed(seed)
    W = {}
    for k, s in param_shapes(c).items():
        if k.endswith("cluster_usage"):
            t = 1.0 + torch.rand(s, generator=g)
        elif k.endswith("embedding_sum"):
            t = torch.randn(s, generator=g)
        elif k.endswith(("layernorm.weight", "norm.weight")):
            t = 1.0 + 0.1 * torch.randn(s, generator=g)
        elif k.endswith((".alpha", ".beta")):
            t = 0.3 * torch.randn(s, generator=g)
        elif k.endswith("layer_scale.scale"):
            t = torch.full(s, 0.05)
        elif k.endswith(".gamma"):
            t = torch.full(s, 0.1)
        elif k.endswith(".bias"):
            t = 0.02 * torch.randn(s, generator=g)
        else:
            fan_in = s[1] * (s[2] if len(s) == 3 else 1)
            if "block.1.conv.weight" in k or ("upsample" in k and k.endswith("0.conv.weight")):
                fan_in = s[0] * 2 if "block.1" in k else s[0]     # ConvTranspose: [Cin, Cout, k], 2 taps overlap
            if k.endswith("dwconv.conv.weight"):
                fan_in = 7
            t = torch.randn(s, generator=g) / math.sqrt(fan_in)
            if k.endswith("conv2.conv.weight"):
                t = t * 0.25            # residual branches small: the 12 stacked units must not blow up
            if s[0] == 1:
                t = t * 0.1             # final 1-channel conv: waveform O(0.1), clamp(-1,1) rarely active
        W[k] = t.to(torch.bfloat16).float()
    return W


def bfr(x):
    """round to bf16, keep fp32 container"""
    return x.to(torch.bfloat16).float()


class Qwen3CodecRef:
    def __init__(self, cfg: CodecCfg, W: Dict[str, torch.Tensor]):
        self.c, self.W = cfg, {k: v.float() for k, v in W.items()}
        c = cfg
        self.emb = []
        for name, n in (("rvq_first", 1), ("rvq_rest", c.num_quantizers - 1)):
            for j in range(n):
                p = f"quantizer.{name}.vq.layers.{j}._codebook."
                self.emb.append(self.W[p + "embedding_sum"] / self.W[p + "cluster_usage"].clamp(min=1e-5)[:, None])
        self.inv_freq = 1.0 / (c.rope_theta ** (torch.arange(0, c.head_dim, 2, dtype=torch.float32) / c.head_dim))

    # ---- state ------------------------------------------------------------------------------------
    def init_state(self, B):
        c = self.c
        st = {"pos": 0,
              "kv": torch.zeros(c.num_layers, B, c.sliding_window, 2, c.num_heads * c.head_dim),   # bf16-valued
              "pre_conv": torch.zeros(B, c.codebook_dim, 2),
              "dw": [torch.zeros(B, c.latent_dim, 6) for _ in c.upsampling_ratios],
              "dec0": torch.zeros(B, c.latent_dim, 6)}
        st["tc"], st["ru"] = [], []
        for b, r in enumerate(c.upsample_rates):
            cin, cout = c.decoder_dim // 2 ** b, c.decoder_dim // 2 ** (b + 1)
            st["tc"].append(torch.zeros(B, cin, 1))
            st["ru"].append([torch.zeros(B, cout, 6 * d) for d in (1, 3, 9)])
        st["final"] = torch.zeros(B, c.decoder_dim // 2 ** len(c.upsample_rates), 6)
        return st

    # ---- primitives (contraction operands rounded to bf16) ------------------------------------------
    @staticmethod
    def lin(x, w, b=None):
        return F.linear(bfr(x), bfr(w), b)

    @staticmethod
    def causal_conv(x, state, w, b, dil=1):
        """x [B,C,T]; state [B,C,P] holds the previous P inputs; returns y [B,Cout,T], new state."""
        P = state.shape[2]
        ext = torch.cat([state, x], 2)
        y = F.conv1d(bfr(ext), bfr(w), b, dilation=dil)
        return y, ext[:, :, ext.shape[2] - P:].clone()

    @staticmethod
    def trans_conv_stream(x, state, w, b, stride):
        """ConvTranspose1d k=2*stride with 1-sample input state (qwen3_codec.py:359-397)."""
        L = x.shape[2]
        ext = torch.cat([state, x], 2)
        raw = F.conv_transpose1d(bfr(ext), bfr(w), b, stride=stride)
        return raw[:, :, stride: stride + L * stride].contiguous(), x[:, :, -1:].clone()

    def snake(self, x, pfx):
        a = torch.exp(self.W[pfx + ".alpha"])[None, :, None]
        bt = torch.exp(self.W[pfx + ".beta"])[None, :, None]
        return x + (1.0 / (bt + 1e-9)) * torch.sin(x * a) ** 2

    def rms(self, x, w):
        v = x.pow(2).mean(-1, keepdim=True)
        return w * (x * torch.rsqrt(v + self.c.rms_eps))

    # ---- transformer ---------------------------------------------------------------------------------
    def transformer(self, x, st):
        """x [B,T,latent] -> [B,T,latent]"""
        c, W = self.c, self.W
        B, T, _ = x.shape
        Hh, D, Wn = c.num_heads, c.head_dim, c.sliding_window
        h = self.lin(x, W["pre_transformer.input_proj.weight"], W["pre_transformer.input_proj.bias"])
        pos = st["pos"] + torch.arange(T, dtype=torch.float32)
        fr = pos[:, None] * self.inv_freq[None, :]
        emb = torch.cat([fr, fr], -1)
        cos, sin = emb.cos()[None, :, None, :], emb.sin()[None, :, None, :]

        def rot(t):   # [B,T,H,D]
            t1, t2 = t[..., : D // 2], t[..., D // 2:]
            return t * cos + torch.cat([-t2, t1], -1) * sin
        for i in range(c.num_layers):
            p = f"pre_transformer.layers.{i}."
            xn = self.rms(h, W[p + "input_layernorm.weight"])
            q = rot(self.lin(xn, W[p + "self_attn.q_proj.weight"]).view(B, T, Hh, D))
            k = rot(self.lin(xn, W[p + "self_attn.k_proj.weight"]).view(B, T, Hh, D))
            v = self.lin(xn, W[p + "self_attn.v_proj.weight"]).view(B, T, Hh, D)
            kv = st["kv"][i]                                             # [B,Wn,2,H*D]; shift-left + append
            kv[:, : Wn - T] = kv[:, T:].clone()
            kv[:, Wn - T:, 0] = bfr(k.reshape(B, T, Hh * D))
            kv[:, Wn - T:, 1] = bfr(v.reshape(B, T, Hh * D))
            Kf, Vf = kv[:, :, 0].view(B, Wn, Hh, D), kv[:, :, 1].view(B, Wn, Hh, D)
            s = torch.einsum("bthd,bjhd->bhtj", q, Kf) / math.sqrt(D)
            j = torch.arange(Wn)[None, :]
            mask = j <= (Wn - T + torch.arange(T))[:, None]              # zero-filled old slots stay visible (Q4)
            s = s.masked_fill(~mask[None, None], float("-inf"))
            a = torch.einsum("bhtj,bjhd->bthd", torch.softmax(s, -1), Vf).reshape(B, T, Hh * D)
            h = h + W[p + "self_attn_layer_scale.scale"] * self.lin(a, W[p + "self_attn.o_proj.weight"])
            xn = self.rms(h, W[p + "post_attention_layernorm.weight"])
            g = self.lin(xn, W[p + "mlp.gate_proj.weight"])
            u = self.lin(xn, W[p + "mlp.up_proj.weight"])
            h = h + W[p + "mlp_layer_scale.scale"] * self.lin(F.silu(g) * u, W[p + "mlp.down_proj.weight"])
        h = self.rms(h, W["pre_transformer.norm.weight"])
        st["pos"] += T
        return self.lin(h, W["pre_transformer.output_proj.weight"], W["pre_transformer.output_proj.bias"])

    # ---- one streaming chunk -------------------------------------------------------------------------
    def forward_chunk(self, codes, st):
        """codes int64 [B, num_quantizers, T] -> wav fp32 [B, 1, T*total_upsample]; st updated in place."""
        c, W = self.c, self.W
        B, Q, T = codes.shape
        q0 = F.embedding(codes[:, 0], self.emb[0])                               # [B,T,vq]
        qr = torch.zeros_like(q0)
        for k in range(1, Q):
            qr = qr + F.embedding(codes[:, k], self.emb[k])
        h = self.lin(q0, W["quantizer.rvq_first.output_proj.weight"][:, :, 0]) + \
            self.lin(qr, W["quantizer.rvq_rest.output_proj.weight"][:, :, 0])    # [B,T,codebook_dim]
        h = h.transpose(1, 2)
        h, st["pre_conv"] = self.causal_conv(h, st["pre_conv"], W["pre_conv.conv.weight"], W["pre_conv.conv.bias"])
        h = self.transformer(h.transpose(1, 2), st).transpose(1, 2)             # [B,latent,T]
        for u, f_ in enumerate(c.upsampling_ratios):
            p = f"upsample.{u}."
            h = F.conv_transpose1d(bfr(h), bfr(W[p + "0.conv.weight"]), W[p + "0.conv.bias"], stride=f_)
            res = h
            w_dw = W[p + "1.dwconv.conv.weight"]
            ext = torch.cat([st["dw"][u], h], 2)
            y = F.conv1d(ext, w_dw, W[p + "1.dwconv.conv.bias"], groups=h.shape[1])   # depthwise: fp32 (no MFMA)
            st["dw"][u] = ext[:, :, ext.shape[2] - 6:].clone()
            y = F.layer_norm(y.transpose(1, 2), (h.shape[1],), W[p + "1.norm.weight"], W[p + "1.norm.bias"], 1e-6)
            y = F.gelu(self.lin(y, W[p + "1.pwconv1.weight"], W[p + "1.pwconv1.bias"]))
            y = self.lin(y, W[p + "1.pwconv2.weight"], W[p + "1.pwconv2.bias"])
            h = res + (W[p + "1.gamma"] * y).transpose(1, 2)
        h, st["dec0"] = self.causal_conv(h, st["dec0"], W["decoder.0.conv.weight"], W["decoder.0.conv.bias"])
        for b, r in enumerate(c.upsample_rates):
            p = f"decoder.{b + 1}.block."
            h = self.snake(h, p + "0")
            h, st["tc"][b] = self.trans_conv_stream(h, st["tc"][b], W[p + "1.conv.weight"], W[p + "1.conv.bias"], r)
            for u, d in enumerate((1, 3, 9)):
                q = f"{p}{u + 2}."
                res = h
                y = self.snake(h, q + "act1")
                y, st["ru"][b][u] = self.causal_conv(y, st["ru"][b][u], W[q + "conv1.conv.weight"],
                                                     W[q + "conv1.conv.bias"], d)
                y = self.snake(y, q + "act2")
                y = F.conv1d(bfr(y), bfr(W[q + "conv2.conv.weight"]), W[q + "conv2.conv.bias"])
                h = y + res
        nb = len(c.upsample_rates)
        h = self.snake(h, f"decoder.{nb + 1}")
        ext = torch.cat([st["final"], h], 2)
        wav = F.conv1d(ext, W[f"decoder.{nb + 2}.conv.weight"], W[f"decoder.{nb + 2}.conv.bias"])   # 1 channel: fp32 dot
        st["final"] = ext[:, :, ext.shape[2] - 6:].clone()
        return wav.clamp(-1, 1)
