"""CPU oracle for the GLM-4-Voice detokenizer (speech tokens -> mel -> waveform).  TEST INFRASTRUCTURE ONLY.

Restates, in plain torch-CPU fp32 on explicit tensors (reference state_dict names):
  GLMFlowModel.inference                 /root/reference/vox_serve/tokenizer/glm.py:2065-2112
  BlockConformerEncoder (BaseEncoder.forward :803-853) with BlockRelPositionMultiHeadedAttention :434-599: relative-position attention
                                         under a causal-or-same-block mask (block 10), no cache
  InterpolateRegulator                   glm.py:1114-1148   nearest resampling to the mel length, 4 x (conv k3, GroupNorm(1), Mish), conv k1
  ConditionalCFM.forward / solve_euler   glm.py:1922-1990   cosine schedule, 10 Euler steps, classifier-free guidance 0.7 as two estimator calls
  ConditionalDecoder.forward             glm.py:1812-1895   non-causal U-Net: 2 down (stride-2 conv between), 12 mid, 2 up (transposed conv
                                         between) ResnetBlock1D (conv k3, GroupNorm(8), Mish) x 4 BasicTransformerBlock each, no masks
  GLMHiFTModel.forward                   glm.py:2556-2594   the HiFT vocoder with two x8 stages and SineGen v1 (:2254-2331): per-sample phase
                                         accumulation at the sample rate + a random initial phase per harmonic
  GLMAudioDecoder.forward                glm.py:2640-2651   embedding = zeros(192)
Noise contract (the reference draws torch.randn_like / Uniform.sample): the CFM start noise [B, 80, T] — request b = Philox stream
`first + b` as oracle/flow_ref.py::cfm_noise; the vocoder: oracle/hift_ref.py::make_noise (uniform stream -> initial phases mapped to
[-pi, pi), normal stream -> additive noise), which for this variant DO matter.
Pinned: tests/test_oracle_goldens.py::test_glm_decoder_* against g13 (reference modules, tiny + GLM-4-Voice size).
"""
import math
from dataclasses import dataclass
from typing import Dict, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from .flow_ref import FlowRef, cfm_noise, rel_pos_table
from .hift_ref import HiftCfg, HiftRef


@dataclass
class GlmFlowCfg:
    vocab: int = 16384
    dim: int = 512
    mel: int = 80
    spk_dim: int = 192
    enc_layers: int = 6
    enc_heads: int = 8
    enc_ffn: int = 2048
    block_size: int = 10
    est_ch: int = 256
    est_heads: int = 8
    est_head_dim: int = 64
    est_blocks: int = 4
    est_mid: int = 12
    n_steps: int = 10
    cfg_rate: float = 0.7
    frame_rate: float = 12.5
    sampling_rate: int = 22050
    hop: int = 256
    reg_layers: int = 4

    @property
    def est_in(self) -> int:
        return 4 * self.mel

    def mel_len(self, n_tokens: int) -> int:
        """(token_len / input_frame_rate * 22050 / 256).int()   glm.py:2084, in float32 like the reference's tensor arithmetic"""
        return int((torch.tensor([n_tokens], dtype=torch.int32) / self.frame_rate * self.sampling_rate / self.hop).int().item())


def tiny_glm_flow_cfg() -> GlmFlowCfg:
    return GlmFlowCfg(vocab=113, enc_layers=1, enc_heads=4, enc_ffn=256, est_ch=64, est_heads=2, est_head_dim=32, est_blocks=1, est_mid=1)


def glm_hift_cfg(**kw) -> HiftCfg:
    """GLMHiFTModel's defaults (glm.py:2391-2410)"""
    d = dict(sampling_rate=22050, upsample_rates=(8, 8), upsample_kernel_sizes=(16, 16), source_resblock_kernel_sizes=(7, 11))
    d.update(kw)
    return HiftCfg(**d)


def param_shapes(c: GlmFlowCfg) -> Dict[str, tuple]:
    s = {}

    def lin(n, o, i, bias=True):
        s[n + ".weight"] = (o, i)
        if bias:
            s[n + ".bias"] = (o,)

    def nrm(n, d):
        s[n + ".weight"], s[n + ".bias"] = (d,), (d,)

    def conv(n, o, i, k):
        s[n + ".weight"], s[n + ".bias"] = (o, i, k), (o,)

    D = c.dim
    s["input_embedding.weight"] = (c.vocab, D)
    lin("spk_embed_affine_layer", c.mel, c.spk_dim)
    lin("encoder_proj", c.mel, D)
    lin("encoder.embed.out.0", D, D)
    nrm("encoder.embed.out.1", D)
    nrm("encoder.after_norm", D)
    for i in range(c.enc_layers):
        p = f"encoder.encoders.{i}."
        for q in ("linear_q", "linear_k", "linear_v", "linear_out"):
            lin(p + "self_attn." + q, D, D)
        lin(p + "self_attn.linear_pos", D, D, bias=False)
        s[p + "self_attn.pos_bias_u"] = s[p + "self_attn.pos_bias_v"] = (c.enc_heads, D // c.enc_heads)
        lin(p + "feed_forward.w_1", c.enc_ffn, D)
        lin(p + "feed_forward.w_2", D, c.enc_ffn)
        nrm(p + "norm_ff", D)
        nrm(p + "norm_mha", D)
    for i in range(c.reg_layers):
        conv(f"length_regulator.model.{3 * i}", c.mel, c.mel, 3)
        nrm(f"length_regulator.model.{3 * i + 1}", c.mel)
    conv(f"length_regulator.model.{3 * c.reg_layers}", c.mel, c.mel, 1)
    e = "decoder.estimator."
    C, TE, inner = c.est_ch, 4 * c.est_ch, c.est_heads * c.est_head_dim
    lin(e + "time_mlp.linear_1", TE, c.est_in)
    lin(e + "time_mlp.linear_2", TE, TE)

    def group(p, cin):
        lin(p + "0.mlp.1", C, TE)
        conv(p + "0.block1.block.0", C, cin, 3)
        nrm(p + "0.block1.block.1", C)
        conv(p + "0.block2.block.0", C, C, 3)
        nrm(p + "0.block2.block.1", C)
        conv(p + "0.res_conv", C, cin, 1)
        for j in range(c.est_blocks):
            q = f"{p}1.{j}."
            nrm(q + "norm1", C)
            for n in ("to_q", "to_k", "to_v"):
                lin(q + "attn1." + n, inner, C, bias=False)
            lin(q + "attn1.to_out.0", C, inner)
            nrm(q + "norm3", C)
            lin(q + "ff.net.0.proj", 4 * C, C)
            lin(q + "ff.net.2", C, 4 * C)

    group(e + "down_blocks.0.", c.est_in)
    conv(e + "down_blocks.0.2.conv", C, C, 3)                  # Downsample1D: stride 2
    group(e + "down_blocks.1.", C)
    conv(e + "down_blocks.1.2", C, C, 3)
    for i in range(c.est_mid):
        group(f"{e}mid_blocks.{i}.", C)
    group(e + "up_blocks.0.", 2 * C)
    s[e + "up_blocks.0.2.conv.weight"], s[e + "up_blocks.0.2.conv.bias"] = (C, C, 4), (C,)     # ConvTranspose1d(4, 2, 1): [Cin, Cout, k]
    group(e + "up_blocks.1.", 2 * C)
    conv(e + "up_blocks.1.2", C, C, 3)
    conv(e + "final_block.block.0", C, C, 3)
    nrm(e + "final_block.block.1", C)
    conv(e + "final_proj", c.mel, C, 1)
    return s


def random_glm_flow_weights(c: GlmFlowCfg, seed=0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    W = {}
    for k, shp in param_shapes(c).items():
        if k.endswith("pos_bias_u") or k.endswith("pos_bias_v"):
            t = 0.3 * torch.randn(shp, generator=g)
        elif k == "input_embedding.weight":
            t = torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            t = 0.05 * torch.randn(shp, generator=g)
        elif len(shp) == 1:
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        else:
            fan = int(np.prod(shp[1:]))
            t = torch.randn(shp, generator=g) / math.sqrt(fan)
            if ".to_out." in k or k.endswith("ff.net.2.weight") or k.endswith("linear_out.weight") or k.endswith("w_2.weight"):
                t = t * 0.5
        W[k] = t.to(torch.bfloat16).float()
    return W


def block_mask(T: int, block: int) -> torch.Tensor:
    """BlockRelPositionMultiHeadedAttention._create_grid_mask(fill_triangle=True): key j visible to query i iff j <= i or same block."""
    i = torch.arange(T)
    return (i[None, :] <= i[:, None]) | ((i[None, :] // block) == (i[:, None] // block))


class GlmFlowRef(FlowRef):
    def __init__(self, cfg: GlmFlowCfg, W):
        self.c, self.W = cfg, {k: v.float() for k, v in W.items()}

    def encoder(self, tok_emb):
        c = self.c
        x = self._embed(tok_emb, "encoder.embed")
        T = x.shape[1]
        pos = rel_pos_table(c.dim, T).unsqueeze(0)
        m = block_mask(T, c.block_size)[None, None]
        for i in range(c.enc_layers):
            x, _ = self._conformer(x, f"encoder.encoders.{i}.", pos, None, mask=m)
        return self._ln(x, "encoder.after_norm", 1e-5)

    def regulator(self, h, Tm):
        """h [B, T, mel] -> [B, Tm, mel]"""
        x = F.interpolate(h.transpose(1, 2).contiguous(), size=Tm, mode="nearest")
        for i in range(self.c.reg_layers):
            n = f"length_regulator.model.{3 * i}"
            x = F.conv1d(x, self.W[n + ".weight"], self.W[n + ".bias"], padding=1)
            n = f"length_regulator.model.{3 * i + 1}"
            x = F.mish(F.group_norm(x, 1, self.W[n + ".weight"], self.W[n + ".bias"]))
        n = f"length_regulator.model.{3 * self.c.reg_layers}"
        return F.conv1d(x, self.W[n + ".weight"], self.W[n + ".bias"]).transpose(1, 2)

    def _block(self, x, p):
        y = F.conv1d(x, self.W[p + "block.0.weight"], self.W[p + "block.0.bias"], padding=1)
        return F.mish(F.group_norm(y, 8, self.W[p + "block.1.weight"], self.W[p + "block.1.bias"]))

    def _res(self, x, p, temb):
        h = self._block(x, p + "block1.")
        h = h + self._lin(F.mish(temb), p + "mlp.1").unsqueeze(-1)
        h = self._block(h, p + "block2.")
        return h + F.conv1d(x, self.W[p + "res_conv.weight"], self.W[p + "res_conv.bias"])

    def estimator(self, x, mu, t, spks, cond):
        c, e = self.c, "decoder.estimator."
        temb = self.time_embedding(t)
        x = torch.cat([x, mu, spks.unsqueeze(-1).expand(-1, -1, x.shape[-1]), cond], dim=1)

        def group(x, p):
            x = self._res(x, p + "0.", temb).transpose(1, 2)
            for j in range(c.est_blocks):
                x, _ = self._tblock(x, f"{p}1.{j}.", None)
            return x.transpose(1, 2)

        W = self.W
        hs = []
        x = group(x, e + "down_blocks.0.")
        hs.append(x)
        x = F.conv1d(x, W[e + "down_blocks.0.2.conv.weight"], W[e + "down_blocks.0.2.conv.bias"], stride=2, padding=1)
        x = group(x, e + "down_blocks.1.")
        hs.append(x)
        x = F.conv1d(x, W[e + "down_blocks.1.2.weight"], W[e + "down_blocks.1.2.bias"], padding=1)
        for i in range(c.est_mid):
            x = group(x, f"{e}mid_blocks.{i}.")
        sk = hs.pop()
        x = group(torch.cat([x[:, :, : sk.shape[-1]], sk], dim=1), e + "up_blocks.0.")
        x = F.conv_transpose1d(x, W[e + "up_blocks.0.2.conv.weight"], W[e + "up_blocks.0.2.conv.bias"], stride=2, padding=1)
        sk = hs.pop()
        x = group(torch.cat([x[:, :, : sk.shape[-1]], sk], dim=1), e + "up_blocks.1.")
        x = F.conv1d(x, W[e + "up_blocks.1.2.weight"], W[e + "up_blocks.1.2.bias"], padding=1)
        x = self._block(x, e + "final_block.")
        return F.conv1d(x, W[e + "final_proj.weight"], W[e + "final_proj.bias"])

    def cfm(self, mu, spks, cond, z):
        ts = self.t_span()
        x = z
        t, dt = ts[0], ts[1] - ts[0]
        for step in range(1, len(ts)):
            tt = t.unsqueeze(0) if t.ndim < 1 else t
            d = self.estimator(x, mu, tt, spks, cond)
            d0 = self.estimator(x, torch.zeros_like(mu), tt, torch.zeros_like(spks), torch.zeros_like(cond))
            d = (1.0 + self.c.cfg_rate) * d - self.c.cfg_rate * d0
            x = x + dt * d
            t = t + dt
            if step < len(ts) - 1:
                dt = ts[step + 1] - t
        return x

    def time_embedding(self, t):
        """the reference passes a 0-dim t (one embedding row, broadcast over the batch)"""
        return super().time_embedding(t.reshape(-1))

    def inference(self, token, z):
        """token [B, T] -> mel [B, mel, Tm]; z [B, mel, Tm] start noise; the speaker embedding is all-zero (glm.py:2647)"""
        c = self.c
        B, T = token.shape
        emb = self._lin(F.normalize(torch.zeros(B, c.spk_dim), dim=1), "spk_embed_affine_layer")
        h = self.encoder(F.embedding(torch.clamp(token, min=0), self.W["input_embedding.weight"]))
        h = self._lin(h, "encoder_proj")
        Tm = c.mel_len(T)
        h = self.regulator(h, Tm)
        return self.cfm(h.transpose(1, 2).contiguous(), emb, torch.zeros(B, c.mel, Tm), z)


def glm_cfm_noise(seed: int, first_stream: int, B: int, mel: int, T: int) -> torch.Tensor:
    return torch.cat([cfm_noise(seed, first_stream + b, mel, T) for b in range(B)], 0)


class GlmHiftRef(HiftRef):
    """GLMHiFTModel: HiftRef's decode with SineGen v1 as the harmonic source."""

    def source(self, f0, rand_ini, noise):
        """f0 [B, T]; rand_ini [B, H+1] uniforms in [0, 1) (column 0 = 0) -> initial phases -pi + 2 pi u; noise [B, L, H+1]"""
        c = self.cfg
        up, H1 = c.upsample_scale, c.nb_harmonics + 1
        f0s = f0[:, None, :].repeat_interleave(up, dim=2)                                  # [B, 1, L]
        Fm = torch.zeros(f0.shape[0], H1, f0s.shape[-1])
        for i in range(H1):
            Fm[:, i:i + 1, :] = f0s * (i + 1) / c.sampling_rate
        theta = 2 * np.pi * (torch.cumsum(Fm, dim=-1) % 1)
        phase = (-np.pi + 2 * np.pi * rand_ini).unsqueeze(-1).float()
        phase[:, 0, :] = 0
        sines = c.nsf_alpha * torch.sin(theta + phase)
        uv = (f0s > c.voiced_threshold).float()
        amp = uv * c.nsf_sigma + (1 - uv) * c.nsf_alpha / 3
        sw = sines * uv + amp * noise.transpose(1, 2)
        merged = torch.tanh(F.linear(sw.transpose(1, 2), self.W["m_source.l_linear.weight"], self.W["m_source.l_linear.bias"]))
        return merged.transpose(1, 2)
