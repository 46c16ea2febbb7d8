"""CPU oracle for the CosyVoice2 token -> mel flow (conformer encoder + conditional flow matching) and the decode_chunk glue around it
and the HiFT vocoder.  TEST INFRASTRUCTURE ONLY.

Restates, in plain torch-CPU fp32 on explicit tensors (reference state_dict names):
  CausalMaskedDiffWithXvec.forward_chunk          /root/reference/vox_serve/tokenizer/cosyvoice_flow.py:2909-2980
  UpsampleConformerEncoder.forward_chunk          cosyvoice_flow.py:1185-1358  (LinearNoSubsampling :489-531, EspnetRelPositionalEncoding
                                                  :399-487, PreLookaheadLayer :561-605, ConformerEncoderLayer :899-1020 without macaron / conv
                                                  module, RelPositionMultiHeadedAttention :742-860 incl. its rel_shift on a cached key axis,
                                                  PositionwiseFeedForward :862-897, Upsample1D :533-559)
  CausalConditionalCFM.forward_chunk / solve_euler_with_cache   cosyvoice_flow.py:2646-2793  (cosine schedule, classifier-free guidance 0.7)
  CausalConditionalDecoder.forward_chunk          cosyvoice_flow.py:2440-2586  (SinusoidalPosEmb :1756, TimestepEmbedding :1815, CausalResnetBlock1D
                                                  :1989 with cached CausalConv1d :1915, BasicTransformerBlock.forward_chunk :1681-1754 with
                                                  Attention.forward_chunk :203-271; the down / up CausalConv1d and the final block are NOT cached there)
  CosyVoice2Decoder.init_cache / decode_chunk     /root/reference/vox_serve/tokenizer/cosyvoice2.py:862-1046 (sliding-window truncation of the
                                                  caches: 16-entry prefix + suffix, 128 entries; shared-prompt mode and per-request mode)
The reference draws the CFM's start noise with torch.randn (one [1, 80, T] draw per call, shared by the batch): the contract here is an
explicit tensor; `cfm_noise` below is the seeded stream (Philox, as oracle/snac_ref.py::philox_noise), element c * T + t of stream `stream`.
The reference serves this module in bf16; like the codecs, the oracle (and the library) compute in fp32.
Pinned: tests/test_oracle_goldens.py::test_flow_* against g12 (reference modules, tiny + CosyVoice2 size).
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .snac_ref import philox_noise


@dataclass
class FlowCfg:
    """CosyVoice2 (tokenizer/cosyvoice2.py:812-835)"""
    vocab: int = 6561
    dim: int = 512                   # encoder width (input_size = output_size = 512; the reference hard-codes 512 in two sub-modules)
    mel: int = 80
    spk_dim: int = 192
    enc_layers: int = 6
    up_layers: int = 4
    enc_heads: int = 8
    enc_ffn: int = 2048
    pre_lookahead: int = 3
    est_ch: int = 256
    est_heads: int = 8
    est_head_dim: int = 64
    est_blocks: int = 4              # transformer blocks per resnet
    est_mid: int = 12
    n_steps: int = 10
    cfg_rate: float = 0.7
    max_cache: int = 128             # CosyVoice2Decoder.MAX_CACHE_LEN
    prefix: int = 16                 # CosyVoice2Decoder.PREFIX_LEN

    @property
    def est_in(self) -> int:
        return 4 * self.mel

    @property
    def n_resnets(self) -> int:
        return 2 + self.est_mid

    @property
    def n_att(self) -> int:
        return self.n_resnets * self.est_blocks


def tiny_flow_cfg() -> FlowCfg:
    return FlowCfg(vocab=97, enc_layers=1, enc_heads=4, enc_ffn=256, est_ch=64, est_heads=2, est_head_dim=32, est_blocks=1, est_mid=1)      # (n_steps stays 10: forward_chunk hard-codes it)


def param_shapes(c: FlowCfg) -> Dict[str, tuple]:
    s = {}

    def lin(n, o, i, bias=True):
        s[n + ".weight"] = (o, i)
        if bias:
            s[n + ".bias"] = (o,)

    def ln(n, d):
        s[n + ".weight"], s[n + ".bias"] = (d,), (d,)

    def conv(n, o, i, k):
        s[n + ".weight"], s[n + ".bias"] = (o, i, k), (o,)

    D = c.dim
    s["input_embedding.weight"] = (c.vocab, D)
    lin("spk_embed_affine_layer", c.mel, c.spk_dim)
    lin("encoder_proj", c.mel, D)
    for emb in ("encoder.embed", "encoder.up_embed"):
        lin(emb + ".out.0", D, D)
        ln(emb + ".out.1", D)
    ln("encoder.after_norm", D)
    conv("encoder.pre_lookahead_layer.conv1", D, D, c.pre_lookahead + 1)
    conv("encoder.pre_lookahead_layer.conv2", D, D, 3)
    conv("encoder.up_layer.conv", D, D, 5)
    for grp, nl in (("encoder.encoders", c.enc_layers), ("encoder.up_encoders", c.up_layers)):
        for i in range(nl):
            p = f"{grp}.{i}."
            for q in ("linear_q", "linear_k", "linear_v", "linear_out"):
                lin(p + "self_attn." + q, D, D)
            lin(p + "self_attn.linear_pos", D, D, bias=False)
            s[p + "self_attn.pos_bias_u"] = s[p + "self_attn.pos_bias_v"] = (c.enc_heads, D // c.enc_heads)
            lin(p + "feed_forward.w_1", c.enc_ffn, D)
            lin(p + "feed_forward.w_2", D, c.enc_ffn)
            ln(p + "norm_ff", D)
            ln(p + "norm_mha", D)
    e = "decoder.estimator."
    C, TE, inner = c.est_ch, 4 * c.est_ch, c.est_heads * c.est_head_dim
    lin(e + "time_mlp.linear_1", TE, c.est_in)
    lin(e + "time_mlp.linear_2", TE, TE)

    def resnet(p, cin):
        lin(p + "mlp.1", C, TE)
        conv(p + "block1.block.0", C, cin, 3)
        ln(p + "block1.block.2", C)
        conv(p + "block2.block.0", C, C, 3)
        ln(p + "block2.block.2", C)
        conv(p + "res_conv", C, cin, 1)

    def tblock(p):
        ln(p + "norm1", C)
        for q in ("to_q", "to_k", "to_v"):
            lin(p + "attn1." + q, inner, C, bias=False)
        lin(p + "attn1.to_out.0", C, inner)
        ln(p + "norm3", C)
        lin(p + "ff.net.0.proj", 4 * C, C)
        lin(p + "ff.net.2", C, 4 * C)

    def group(p, cin):
        resnet(p + "0.", cin)
        for j in range(c.est_blocks):
            tblock(f"{p}1.{j}.")

    group(e + "down_blocks.0.", c.est_in)
    conv(e + "down_blocks.0.2", C, C, 3)
    for i in range(c.est_mid):
        group(f"{e}mid_blocks.{i}.", C)
    group(e + "up_blocks.0.", 2 * C)
    conv(e + "up_blocks.0.2", C, C, 3)
    conv(e + "final_block.block.0", C, C, 3)
    ln(e + "final_block.block.2", C)
    conv(e + "final_proj", c.mel, C, 1)
    return s


def random_flow_weights(c: FlowCfg, seed=0) -> Dict[str, torch.Tensor]:
    """fp32 CPU tensors with bf16-representable values, fan-in scaled so that activations stay O(1) through the stacks."""
    g = torch.Generator().manual_seed(seed)
    W = {}
    for k, shp in param_shapes(c).items():
        if k.endswith("pos_bias_u") or k.endswith("pos_bias_v"):
            t = 0.3 * torch.randn(shp, generator=g)
        elif k == "input_embedding.weight":
            t = torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            t = 0.05 * torch.randn(shp, generator=g)
        elif len(shp) == 1:                                     # LayerNorm weight
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        else:
            fan = int(np.prod(shp[1:]))
            t = torch.randn(shp, generator=g) / math.sqrt(fan)
            if ".to_out." in k or k.endswith("ff.net.2.weight") or k.endswith("linear_out.weight") or k.endswith("w_2.weight"):
                t = t * 0.5                                     # residual branches
        W[k] = t.to(torch.bfloat16).float()
    return W


def cfm_noise(seed: int, stream: int, mel: int, T: int) -> torch.Tensor:
    """[1, mel, T] standard normals: element (c, t) = element c * T + t of Philox stream `stream`."""
    return torch.from_numpy(philox_noise(seed, stream, mel * T).reshape(1, mel, T))


def truncate_cache(x: torch.Tensor, dim: int, max_len: int, prefix: int) -> torch.Tensor:
    """cosyvoice2.py:899-919: keep the first `prefix` and the last `max_len - prefix` entries along `dim`."""
    n = x.shape[dim]
    if n <= max_len:
        return x
    idx = torch.cat([torch.arange(prefix), torch.arange(n - (max_len - prefix), n)])
    return x.index_select(dim, idx)


def rel_pos_table(d_model: int, size: int) -> torch.Tensor:
    """EspnetRelPositionalEncoding.position_encoding(offset=0, size): [2 size - 1, d_model], row j = encoding of position size - 1 - j."""
    pos = torch.arange(size - 1, -size, -1, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
    pe = torch.zeros(2 * size - 1, d_model)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def rel_shift(x: torch.Tensor) -> torch.Tensor:
    """RelPositionMultiHeadedAttention.rel_shift, literally (time1 may differ from (x.size(3) + 1) / 2 when keys are cached)."""
    zero_pad = torch.zeros((x.size(0), x.size(1), x.size(2), 1), dtype=x.dtype)
    xp = torch.cat([zero_pad, x], dim=-1)
    xp = xp.view(x.size(0), x.size(1), x.size(3) + 1, x.size(2))
    return xp[:, :, 1:].view_as(x)[:, :, :, : x.size(-1) // 2 + 1]


class FlowRef:
    def __init__(self, cfg: FlowCfg, W: Dict[str, torch.Tensor]):
        self.c, self.W = cfg, {k: v.float() for k, v in W.items()}

    # ---- small helpers --------------------------------------------------------------------------------------------
    def _lin(self, x, n, bias=True):
        return F.linear(x, self.W[n + ".weight"], self.W[n + ".bias"] if bias else None)

    def _ln(self, x, n, eps):
        return F.layer_norm(x, (x.shape[-1],), self.W[n + ".weight"], self.W[n + ".bias"], eps)

    # ---- encoder ----------------------------------------------------------------------------------------------------
    def _embed(self, x, n):
        """LinearNoSubsampling + the positional encoding's x * sqrt(d)"""
        return self._ln(self._lin(x, n + ".out.0"), n + ".out.1", 1e-5) * math.sqrt(self.c.dim)

    def _conformer(self, x, p, pos_emb, cache, mask=None):
        """x [B, T, D]; cache [Bc, H, Tc, 2 dk] or None -> (x, new_cache [B, H, Tc + T, 2 dk])"""
        c = self.c
        H, dk = c.enc_heads, c.dim // c.enc_heads
        B, T, _ = x.shape
        r = x
        n = self._ln(x, p + "norm_mha", 1e-12)
        q = self._lin(n, p + "self_attn.linear_q").view(B, T, H, dk)
        k = self._lin(n, p + "self_attn.linear_k").view(B, T, H, dk).transpose(1, 2)
        v = self._lin(n, p + "self_attn.linear_v").view(B, T, H, dk).transpose(1, 2)
        if cache is not None and cache.shape[0] > 0:
            kc, vc = torch.split(cache.expand(B, -1, -1, -1), dk, dim=-1)
            k, v = torch.cat([kc, k], dim=2), torch.cat([vc, v], dim=2)
        new_cache = torch.cat((k, v), dim=-1)
        pe = self._lin(pos_emb, p + "self_attn.linear_pos", bias=False).view(1, -1, H, dk).transpose(1, 2)
        qu = (q + self.W[p + "self_attn.pos_bias_u"]).transpose(1, 2)
        qv = (q + self.W[p + "self_attn.pos_bias_v"]).transpose(1, 2)
        ac = torch.matmul(qu, k.transpose(-2, -1))
        bd = torch.matmul(qv, pe.transpose(-2, -1))
        if ac.shape != bd.shape:
            bd = rel_shift(bd)
        sc = (ac + bd) / math.sqrt(dk)
        if mask is not None:                 # MultiHeadedAttention.forward_attention: -inf before the softmax, 0 after
            sc = sc.masked_fill(~mask, -float("inf"))
        att = torch.softmax(sc, dim=-1)
        if mask is not None:
            att = att.masked_fill(~mask, 0.0)
        o = torch.matmul(att, v).transpose(1, 2).contiguous().view(B, T, H * dk)
        x = r + self._lin(o, p + "self_attn.linear_out")
        n = self._ln(x, p + "norm_ff", 1e-12)
        x = x + self._lin(F.silu(self._lin(n, p + "feed_forward.w_1")), p + "feed_forward.w_2")
        return x, new_cache

    def encoder_chunk(self, tok_emb, enc_cache, up_cache):
        """tok_emb [B, T, D] (embedded tokens); enc_cache [Bc, L, H, Tc, 2dk] / up_cache likewise or None.
        -> (h [B, 2T, D], new_enc_cache [B, L, H, Tc + T, 2dk], new_up_cache)   (context empty, as the streaming path calls it)"""
        c = self.c
        x = self._embed(tok_emb, "encoder.embed")
        # PreLookaheadLayer with an empty context: right-pad by pre_lookahead zeros, conv k4, leaky_relu, left-pad 2, conv k3, residual
        y = F.pad(x.transpose(1, 2), (0, c.pre_lookahead))
        y = F.leaky_relu(F.conv1d(y, self.W["encoder.pre_lookahead_layer.conv1.weight"], self.W["encoder.pre_lookahead_layer.conv1.bias"]))
        y = F.conv1d(F.pad(y, (2, 0)), self.W["encoder.pre_lookahead_layer.conv2.weight"], self.W["encoder.pre_lookahead_layer.conv2.bias"])
        x = y.transpose(1, 2) + x
        Tc = enc_cache.shape[3] if enc_cache is not None else 0
        pos = rel_pos_table(c.dim, Tc + x.shape[1]).unsqueeze(0)
        new_enc = []
        for i in range(c.enc_layers):
            x, nc = self._conformer(x, f"encoder.encoders.{i}.", pos, enc_cache[:, i] if enc_cache is not None else None)
            new_enc.append(nc)
        # Upsample1D: nearest x2, left-pad 4 zeros, conv k5
        y = F.interpolate(x.transpose(1, 2), scale_factor=2.0, mode="nearest")
        y = F.conv1d(F.pad(y, (4, 0)), self.W["encoder.up_layer.conv.weight"], self.W["encoder.up_layer.conv.bias"])
        x = self._embed(y.transpose(1, 2), "encoder.up_embed")
        Tc = up_cache.shape[3] if up_cache is not None else 0
        pos = rel_pos_table(c.dim, Tc + x.shape[1]).unsqueeze(0)
        new_up = []
        for i in range(c.up_layers):
            x, nc = self._conformer(x, f"encoder.up_encoders.{i}.", pos, up_cache[:, i] if up_cache is not None else None)
            new_up.append(nc)
        x = self._ln(x, "encoder.after_norm", 1e-5)
        return x, torch.stack(new_enc, 1), torch.stack(new_up, 1)

    # ---- estimator ----------------------------------------------------------------------------------------------------
    def time_embedding(self, t: torch.Tensor) -> torch.Tensor:
        """t [N] -> [N, 4 C]   (SinusoidalPosEmb(est_in, scale 1000) -> Linear, SiLU, Linear)"""
        half = self.c.est_in // 2
        e = torch.exp(torch.arange(half).float() * -(math.log(10000) / (half - 1)))
        e = 1000 * t.unsqueeze(1) * e.unsqueeze(0)
        e = torch.cat((e.sin(), e.cos()), dim=-1)
        return self._lin(F.silu(self._lin(e, "decoder.estimator.time_mlp.linear_1")), "decoder.estimator.time_mlp.linear_2")

    def _causal_block(self, x, p, cache):
        """CausalBlock1D.forward_chunk: cached conv k3 -> LayerNorm over channels -> Mish.  x [N, C, T], cache [N, C, 2] or None."""
        if cache is None:
            cache = x.new_zeros((x.shape[0], x.shape[1], 2))
        xc = torch.cat([cache, x], dim=2)
        y = F.conv1d(xc, self.W[p + "block.0.weight"], self.W[p + "block.0.bias"])
        y = self._ln(y.transpose(1, 2), p + "block.2", 1e-5).transpose(1, 2)
        return F.mish(y), xc[..., -2:]

    def _resnet(self, x, p, temb, cache):
        C = self.c.est_ch
        c1, c2 = (None, None) if cache is None else cache.split([x.shape[1], C], dim=1)
        h, n1 = self._causal_block(x, p + "block1.", c1)
        h = h + self._lin(F.mish(temb), p + "mlp.1").unsqueeze(-1)
        h, n2 = self._causal_block(h, p + "block2.", c2)
        out = h + F.conv1d(x, self.W[p + "res_conv.weight"], self.W[p + "res_conv.bias"])
        return out, torch.cat([n1, n2], dim=1)

    def _tblock(self, x, p, kv):
        """x [N, T, C]; kv [N, H, Tc, 2 hd] or None -> (x, new kv)"""
        H, hd = self.c.est_heads, self.c.est_head_dim
        N, T, _ = x.shape
        n = self._ln(x, p + "norm1", 1e-5)
        q = self._lin(n, p + "attn1.to_q", False).view(N, T, H, hd).transpose(1, 2)
        k = self._lin(n, p + "attn1.to_k", False).view(N, T, H, hd).transpose(1, 2)
        v = self._lin(n, p + "attn1.to_v", False).view(N, T, H, hd).transpose(1, 2)
        if kv is not None:
            k, v = torch.cat([kv[..., :hd], k], dim=2), torch.cat([kv[..., hd:], v], dim=2)
        new_kv = torch.cat([k, v], dim=-1)
        att = torch.softmax(torch.matmul(q, k.transpose(-2, -1)) * (hd ** -0.5), dim=-1)
        o = torch.matmul(att, v).transpose(1, 2).reshape(N, T, H * hd)
        x = self._lin(o, p + "attn1.to_out.0") + x
        n = self._ln(x, p + "norm3", 1e-5)
        x = self._lin(F.gelu(self._lin(n, p + "ff.net.0.proj")), p + "ff.net.2") + x
        return x, new_kv

    def _causal_conv(self, x, n):
        return F.conv1d(F.pad(x, (2, 0)), self.W[n + ".weight"], self.W[n + ".bias"])

    def estimator_chunk(self, x, mu, t, spks, cond, cnn_cache, att_cache):
        """x, mu, cond [N, mel, T]; t [N]; spks [N, mel]; cnn_cache: list (per resnet) of [N, Cin + C, 2] or None;
        att_cache [N, n_att, H, Tc, 2 hd] or None -> (dphi [N, mel, T], new_cnn list, new_att [N, n_att, H, Tc + T, 2 hd])"""
        c, e = self.c, "decoder.estimator."
        temb = self.time_embedding(t)
        x = torch.cat([x, mu, spks.unsqueeze(-1).expand(-1, -1, x.shape[-1]), cond], dim=1)
        new_cnn, new_att = [], []
        li = [0, 0]

        def group(x, p):
            x, cc = self._resnet(x, p + "0.", temb, cnn_cache[li[0]] if cnn_cache is not None else None)
            new_cnn.append(cc)
            li[0] += 1
            x = x.transpose(1, 2)
            for j in range(c.est_blocks):
                x, kv = self._tblock(x, f"{p}1.{j}.", att_cache[:, li[1]] if att_cache is not None else None)
                new_att.append(kv)
                li[1] += 1
            return x.transpose(1, 2)

        x = group(x, e + "down_blocks.0.")
        skip = x
        x = self._causal_conv(x, e + "down_blocks.0.2")
        for i in range(c.est_mid):
            x = group(x, f"{e}mid_blocks.{i}.")
        x = group(torch.cat([x[:, :, : skip.shape[-1]], skip], dim=1), e + "up_blocks.0.")
        x = self._causal_conv(x, e + "up_blocks.0.2")
        y = self._causal_conv(x, e + "final_block.block.0")
        y = F.mish(self._ln(y.transpose(1, 2), e + "final_block.block.2", 1e-5).transpose(1, 2))
        out = F.conv1d(y, self.W[e + "final_proj.weight"], self.W[e + "final_proj.bias"])
        return out, new_cnn, torch.stack(new_att, dim=1)

    def t_span(self) -> torch.Tensor:
        ts = torch.linspace(0, 1, self.c.n_steps + 1)
        return 1 - torch.cos(ts * 0.5 * torch.pi)

    def cfm_chunk(self, mu, spks, cond, z, cnn_cache, att_cache):
        """solve_euler_with_cache.  mu, cond [B, mel, T]; spks [B, mel]; z [1, mel, T] start noise (shared by the batch);
        cnn_cache: list (per step) of lists (per resnet) of [Bc, 2, Cin + C, 2] or None; att_cache [Bc, 2, n_steps, n_att, H, Tc, 2hd] or None.
        -> (mel [B, mel, T], new_cnn_cache, new_att_cache [B, 2, n_steps, n_att, H, Tc + T, 2hd])"""
        B = mu.shape[0]
        ts = self.t_span()
        x = z.expand(B, -1, -1)
        t, dt = ts[0], ts[1] - ts[0]
        zeros = lambda a: torch.zeros_like(a)
        mu_in, spk_in, cond_in = torch.cat([mu, zeros(mu)]), torch.cat([spks, zeros(spks)]), torch.cat([cond, zeros(cond)])
        new_cnn, new_att = [], []
        for step in range(1, len(ts)):
            x_in = torch.cat([x, x])
            t_in = torch.full((2 * B,), float(t))
            cc = ac = None
            if att_cache is not None:
                ac = att_cache.expand(B, *att_cache.shape[1:])[:, :, step - 1].transpose(0, 1).reshape(2 * B, *att_cache.shape[3:])
            if cnn_cache is not None:
                cc = [k.expand(B, *k.shape[1:]).transpose(0, 1).reshape(2 * B, *k.shape[2:]) for k in cnn_cache[step - 1]]
            d, ncc, nac = self.estimator_chunk(x_in, mu_in, t_in, spk_in, cond_in, cc, ac)
            new_cnn.append([k.reshape(2, B, *k.shape[1:]).transpose(0, 1) for k in ncc])
            new_att.append(nac.reshape(2, B, *nac.shape[1:]).transpose(0, 1))
            d = (1.0 + self.c.cfg_rate) * d[:B] - self.c.cfg_rate * d[B:]
            x = x + dt * d
            t = t + dt
            if step < len(ts) - 1:
                dt = ts[step + 1] - t
        return x, new_cnn, torch.stack(new_att, dim=0).transpose(0, 1).transpose(1, 2)

    # ---- CausalMaskedDiffWithXvec.forward_chunk -----------------------------------------------------------------------
    def flow_chunk(self, token, prompt_feat, embedding, z, cache):
        """token [B, T] int; prompt_feat [1, Tp, mel] (Tp = 0 while decoding); embedding [Be, spk_dim]; z [1, mel, 2T];
        cache: dict(enc, up, cnn, att) of the shapes above, or None (init).  -> (mel [B, mel, 2T], new cache dict)"""
        emb = self._lin(F.normalize(embedding, dim=1), "spk_embed_affine_layer")
        tok = F.embedding(torch.clamp(token, min=0), self.W["input_embedding.weight"])
        h, ne, nu = self.encoder_chunk(tok, cache["enc"] if cache else None, cache["up"] if cache else None)
        h = self._lin(h, "encoder_proj")
        cond = torch.zeros_like(h)
        cond[:, : prompt_feat.shape[1]] = prompt_feat
        B = h.shape[0]
        mel, ncnn, natt = self.cfm_chunk(h.transpose(1, 2).contiguous(), emb.expand(B, -1), cond.transpose(1, 2), z,
                                         cache["cnn"] if cache else None, cache["att"] if cache else None)
        return mel, {"enc": ne, "up": nu, "cnn": ncnn, "att": natt}

    # ---- CosyVoice2Decoder.init_cache (the flow part) -------------------------------------------------------------------
    def init_cache(self, prompt_token, prompt_feat, embedding, z):
        """prompt_token [1, Np]; prompt_feat [1, 2 Np, mel]; z [1, mel, 2 (Np + 3)] -> (prompt mels, truncated cache dict)"""
        c = self.c
        tok = torch.cat([prompt_token, prompt_token[:, :3]], dim=1)
        mel, cache = self.flow_chunk(tok, prompt_feat, embedding, z, None)
        cache["enc"] = truncate_cache(cache["enc"], 3, c.max_cache // 2, c.prefix // 2)
        cache["up"] = truncate_cache(cache["up"], 3, c.max_cache, c.prefix)
        cache["att"] = truncate_cache(cache["att"], 5, c.max_cache, c.prefix)
        return mel, cache


# ---- CosyVoice2Decoder.decode_chunk, shared-prompt mode (the plugin's default: cosyvoice2.py:944-1046, model/cosyvoice2.py:1093-1103) ----
def decode_chunk_shared(fr: "FlowRef", hr, token, embedding, cache, z, rand_ini, noise, mel_cache_len: int = 6):
    """token [B, T] -> (audio [B, 2 T scale - mel_cache_len scale], mel [B, mel, 2T]): flow against the static prompt caches, HiFT, fade-in
    against the (all-zero) initial speech cache with the rising half of a float64 Hamming window, trailing mel_cache_len frames trimmed."""
    mel, _ = fr.flow_chunk(token, torch.zeros(1, 0, fr.c.mel), embedding, z, cache)
    wav, _ = hr.forward_chunk(mel, rand_ini, noise)
    n = mel_cache_len * hr.cfg.upsample_scale
    win = torch.from_numpy(np.hamming(2 * n))
    out = wav.clone()
    out[..., :n] = (wav[..., :n] * win[:n] + torch.zeros(1, n, dtype=torch.bfloat16) * win[n:]).to(wav.dtype)
    return out[:, :-n], mel


# ---- CosyVoice2Decoder.decode_chunk, per-request evolving caches (use_detokenizer_cache=True: cosyvoice2.py:1010-1083) ----
def decode_chunk_evolving(fr: "FlowRef", hr, token, embedding, cache, speech_cache, z, rand_ini, noise, mel_cache_len: int = 6):
    """One chunk of a request that owns its caches (the reference's non-shared mode; not the plugin default).  Pinned to the reference by
    fixture g17; the HIP path (vox_flow_decode_chunk_slots) is held to both.  token [B, T]; cache: dict(enc, up, cnn, att) with batch B (after
    init_cache: the prompt's caches); speech_cache [B, mel_cache_len * scale]: the tail of the previous chunk's faded audio (zeros at
    the start).  Returns (audio [B, (2T - mel_cache_len) scale], mel, new cache, new speech_cache): the flow runs against the request's
    own caches, which then grow by this chunk's rows and are cut back to the sliding window (first `prefix` rows + the most recent
    ones); HiFT as in the shared mode; the fade-in blends against the previous chunk's tail instead of zeros."""
    c = fr.c
    mel, new = fr.flow_chunk(token, torch.zeros(1, 0, c.mel), embedding, z, cache)
    new["enc"] = truncate_cache(new["enc"], 3, c.max_cache // 2, c.prefix // 2)
    new["up"] = truncate_cache(new["up"], 3, c.max_cache, c.prefix)
    new["att"] = truncate_cache(new["att"], 5, c.max_cache, c.prefix)
    wav, _ = hr.forward_chunk(mel, rand_ini, noise)
    n = mel_cache_len * hr.cfg.upsample_scale
    win = torch.from_numpy(np.hamming(2 * n))
    out = wav.clone()
    out[..., :n] = (wav[..., :n] * win[:n] + speech_cache * win[n:]).to(wav.dtype)       # fade_in_out (cosyvoice2.py:1060)
    return out[:, :-n], mel, new, out[:, -n:]
