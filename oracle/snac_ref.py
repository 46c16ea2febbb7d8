"""CPU oracle for the SNAC decoder (token -> waveform of the Orpheus family).  TEST INFRASTRUCTURE ONLY.

Restates, in plain torch-CPU fp32 on explicit (weight-norm-folded) tensors:
  ResidualVectorQuantize.from_codes      /root/reference/vox_serve/tokenizer/snac.py:350-357
  Decoder / DecoderBlock / ResidualUnit  snac.py:119-176, 215-241   (depthwise or dense convs: `depthwise`)
  LocalMHA / SinusoidalEmbeddings        snac.py:20-90              (`attn_window_size`: LayerNorm -> qkv -> rotary q, k per window
                                                                     position -> full softmax inside each window -> out + residual)
  NoiseBlock                             snac.py:201-212            x + noise[b,1,t] * conv1x1(x)
  Snake1d                                snac.py:253-267            x + sin(alpha x)^2 / (alpha + 1e-9)
  SNAC.decode                            snac.py:438-441
  OrpheusModel.postprocess               model/orpheus.py:483-507   7 tokens/frame -> 3 code levels, window of 4 frames,
                                                                     samples [2048:4096] of the 8192 decoded
The reference draws NoiseBlock's noise with torch.randn (irreproducible); the contract here is an explicit noise tensor per
decoder stage — `philox_noise` below is the seeded stream libvoxhip generates on the device (Philox4x32-10 + Box-Muller),
and the parity fixtures inject the same tensors into the reference module (tests/golden/make_goldens.py::g10_snac).
Pinned: tests/test_oracle_goldens.py::test_snac_* against g10 (reference SNAC module, tiny + snac_24khz size).
"""
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class SnacCfg:
    """hubertsiuzdak/snac_24khz (config.json of the checkpoint the reference loads, model/orpheus.py:246-249)."""
    latent_dim: int = 768          # encoder_dim 48 * 2^4
    decoder_dim: int = 1024
    rates: Sequence[int] = (8, 8, 4, 2)
    codebook_size: int = 4096
    codebook_dim: int = 8
    vq_strides: Sequence[int] = (4, 2, 1)
    noise: bool = True
    sampling_rate: int = 24000
    depthwise: bool = True                     # snac_24khz; False: the k7 convs are dense (groups = 1), one conv at the decoder input
    attn_window_size: Optional[int] = None     # LocalMHA after the decoder's input conv(s) (snac_32khz / 44khz: 32)

    @property
    def first_block(self):
        """Index of the first DecoderBlock in decoder.model (snac.py:130-146)."""
        return (2 if self.depthwise else 1) + (1 if self.attn_window_size is not None else 0)

    @property
    def hop(self):
        return int(np.prod(self.rates))


def tiny_snac_cfg() -> SnacCfg:
    # (channels 256 / 128 / 64 / 32: the implicit-GEMM kernels take input channels in multiples of 32)
    return SnacCfg(latent_dim=64, decoder_dim=512, rates=(4, 2, 2, 2), codebook_size=64, codebook_dim=8, vq_strides=(4, 2, 1))


def param_shapes(cfg: SnacCfg) -> Dict[str, tuple]:
    """Reference state_dict names (weight_norm parametrizations: original0 = g, original1 = v)."""
    s = {}

    def wn(name, shape, gdim0):
        s[name + ".parametrizations.weight.original0"] = (gdim0, 1, 1)
        s[name + ".parametrizations.weight.original1"] = shape

    for i in range(len(cfg.vq_strides)):
        q = f"quantizer.quantizers.{i}."
        s[q + "codebook.weight"] = (cfg.codebook_size, cfg.codebook_dim)
        wn(q + "out_proj", (cfg.latent_dim, cfg.codebook_dim, 1), cfg.latent_dim)
        s[q + "out_proj.bias"] = (cfg.latent_dim,)
    d = "decoder.model."
    if cfg.depthwise:
        wn(d + "0", (cfg.latent_dim, 1, 7), cfg.latent_dim)
        s[d + "0.bias"] = (cfg.latent_dim,)
        wn(d + "1", (cfg.decoder_dim, cfg.latent_dim, 1), cfg.decoder_dim)
        s[d + "1.bias"] = (cfg.decoder_dim,)
    else:
        wn(d + "0", (cfg.decoder_dim, cfg.latent_dim, 7), cfg.decoder_dim)
        s[d + "0.bias"] = (cfg.decoder_dim,)
    if cfg.attn_window_size is not None:
        a = f"{d}{cfg.first_block - 1}."
        s[a + "norm.weight"] = (cfg.decoder_dim,)
        s[a + "norm.bias"] = (cfg.decoder_dim,)
        s[a + "to_qkv.weight"] = (3 * cfg.decoder_dim, cfg.decoder_dim)
        s[a + "to_out.weight"] = (cfg.decoder_dim, cfg.decoder_dim)
    ch = cfg.decoder_dim
    for bi, r in enumerate(cfg.rates):
        b = f"{d}{cfg.first_block + bi}.block."
        cin, cout = ch, ch // 2
        s[b + "0.alpha"] = (1, cin, 1)
        wn(b + "1", (cin, cout, 2 * r), cin)                  # ConvTranspose1d: weight_norm over dim 0 = input channels
        s[b + "1.bias"] = (cout,)
        j = 2
        if cfg.noise:
            wn(b + "2.linear", (cout, cout, 1), cout)
            j = 3
        for u in range(3):
            ru = f"{b}{j + u}.block."
            s[ru + "0.alpha"] = (1, cout, 1)
            wn(ru + "1", (cout, 1 if cfg.depthwise else cout, 7), cout)
            s[ru + "1.bias"] = (cout,)
            s[ru + "2.alpha"] = (1, cout, 1)
            wn(ru + "3", (cout, cout, 1), cout)
            s[ru + "3.bias"] = (cout,)
        ch = cout
    n = cfg.first_block + len(cfg.rates)
    s[f"{d}{n}.alpha"] = (1, ch, 1)
    wn(f"{d}{n + 1}", (1, ch, 7), 1)
    s[f"{d}{n + 1}.bias"] = (1,)
    return s


def random_snac_weights(cfg: SnacCfg, seed=0, final_gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """fp32 CPU tensors with bf16-representable values, scaled so that the waveform stays O(0.1)."""
    g = torch.Generator().manual_seed(seed)
    W = {}
    for k, shp in param_shapes(cfg).items():
        if k.endswith("alpha"):
            t = 0.5 + torch.rand(shp, generator=g)
        elif k.endswith("original0"):
            t = 0.7 + 0.6 * torch.rand(shp, generator=g)
        elif k.endswith("codebook.weight"):
            t = torch.randn(shp, generator=g)
        elif k.endswith("norm.weight"):
            t = 0.8 + 0.4 * torch.rand(shp, generator=g)
        elif k.endswith("to_qkv.weight") or k.endswith("to_out.weight"):
            t = torch.randn(shp, generator=g) / math.sqrt(shp[1]) * (1.0 if k.endswith("to_qkv.weight") else 0.5)
        elif k.endswith("bias"):
            t = 0.02 * torch.randn(shp, generator=g)
        else:
            t = torch.randn(shp, generator=g)
        W[k] = t.to(torch.bfloat16).float()
    # per-layer gains (through g) that keep the 12 stacked residual units and 4 upsamplers from blowing up
    for k in W:
        if k.endswith("original0"):
            if ".linear." in k:
                W[k] = (W[k] * 0.3).to(torch.bfloat16).float()              # noise branch
            elif k.split(".parametrizations")[0].endswith(".3"):
                W[k] = (W[k] * 0.35).to(torch.bfloat16).float()             # residual-unit output conv
            elif W[k].shape[0] == 1:
                W[k] = (W[k] * 0.12 * final_gain).to(torch.bfloat16).float()   # final conv: keep tanh out of saturation
    return W


def fold_weight_norm(W: Dict[str, torch.Tensor], name: str) -> torch.Tensor:
    """w = g * v / ||v||, the norm over every dim but 0 (torch weight_norm default dim=0, also for ConvTranspose1d)."""
    g, v = W[name + ".parametrizations.weight.original0"].float(), W[name + ".parametrizations.weight.original1"].float()
    n = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return g * v / n


# ---- seeded noise stream (what libvoxhip's k_snac_noise generates) -------------------------------------------------
def _philox4x32_10(ctr: np.ndarray, key: np.ndarray) -> np.ndarray:
    """ctr [n,4] uint32, key [2] uint32 -> [n,4] uint32 (Philox4x32-10, same constants as oracle/voxref.c::philox_u32)."""
    M0, M1, W0, W1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
    c = ctr.astype(np.uint32).copy()
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    for _ in range(10):
        p0 = c[:, 0].astype(np.uint64) * M0
        p1 = c[:, 2].astype(np.uint64) * M1
        hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
        hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
        c = np.stack([hi1 ^ c[:, 1] ^ k0, lo1, hi0 ^ c[:, 3] ^ k1, lo0], axis=1)
        k0, k1 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF), np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c


def philox_noise(seed: int, stream: int, n: int) -> np.ndarray:
    """n standard normals: element i = Box-Muller of words 0,1 of Philox(counter = (i, stream, 0, 0), key = seed lo/hi):
    u1 = (w0 >> 8 + 1) / 2^24 in (0,1], u2 = (w1 >> 8) / 2^24, z = sqrt(-2 ln u1) cos(2 pi u2)   (fp32 on the device)."""
    ctr = np.zeros((n, 4), np.uint32)
    ctr[:, 0] = np.arange(n, dtype=np.uint32)
    ctr[:, 1] = np.uint32(stream)
    r = _philox4x32_10(ctr, np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], np.uint32))
    u1 = ((r[:, 0] >> np.uint32(8)).astype(np.float64) + 1.0) / 16777216.0
    u2 = (r[:, 1] >> np.uint32(8)).astype(np.float64) / 16777216.0
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)).astype(np.float32)


def stage_lengths(cfg: SnacCfg, T: int) -> List[int]:
    out, t = [], T
    for r in cfg.rates:
        t *= r
        out.append(t)
    return out


def make_noise(cfg: SnacCfg, B: int, T: int, seed: int, first_stream: int = 0) -> List[torch.Tensor]:
    """One [B,1,T_i] tensor per decoder stage; request b, stage i uses stream first_stream + b * n_stages + i."""
    ns = len(cfg.rates)
    return [torch.from_numpy(np.stack([philox_noise(seed, first_stream + b * ns + i, L) for b in range(B)])[:, None, :])
            for i, L in enumerate(stage_lengths(cfg, T))]


# ---- the decoder --------------------------------------------------------------------------------------------------
def snake(x, alpha):
    return x + (alpha + 1e-9).reciprocal() * torch.sin(alpha * x).pow(2)


class SnacRef:
    def __init__(self, cfg: SnacCfg, W: Dict[str, torch.Tensor]):
        self.cfg, self.W = cfg, {k: v.float() for k, v in W.items()}
        self.w = {k.split(".parametrizations")[0]: fold_weight_norm(self.W, k.split(".parametrizations")[0])
                  for k in W if k.endswith("original0")}

    def from_codes(self, codes: List[torch.Tensor]) -> torch.Tensor:
        """codes[i] [B, T / stride_i] -> z_q [B, latent, T]   (snac.py:350-357)"""
        z = 0.0
        for i, st in enumerate(self.cfg.vq_strides):
            q = f"quantizer.quantizers.{i}."
            e = F.embedding(codes[i].long(), self.W[q + "codebook.weight"]).transpose(1, 2)
            zi = F.conv1d(e, self.w[q + "out_proj"], self.W[q + "out_proj.bias"])
            z = z + zi.repeat_interleave(st, dim=-1)
        return z

    def local_mha(self, x, p):
        """LocalMHA.forward (snac.py:33-47), dim_head 64, rotary without xpos: x [B, C, T], T a multiple of the window."""
        W, ws, dh = self.W, self.cfg.attn_window_size, 64
        B, C, T = x.shape
        if T % ws:
            raise ValueError(f"LocalMHA: {T} frames are not a multiple of the window {ws}")
        h = C // dh
        y = F.layer_norm(x.transpose(1, 2), (C,), W[p + "norm.weight"], W[p + "norm.bias"], 1e-5)
        q, k, v = (y @ W[p + "to_qkv.weight"].t()).chunk(3, dim=-1)
        q, k, v = (t.reshape(B, T // ws, ws, h, dh).permute(0, 3, 1, 2, 4) for t in (q, k, v))       # b h w n d
        inv_freq = 1.0 / (10000 ** (torch.arange(0, dh, 2).float() / dh))
        fr = torch.einsum("i,j->ij", torch.arange(ws).float(), inv_freq)
        fr = torch.cat((fr, fr), dim=-1)
        rot = lambda t: torch.cat((-t[..., dh // 2:], t[..., : dh // 2]), dim=-1)
        q = q * fr.cos() + rot(q) * fr.sin()
        k = k * fr.cos() + rot(k) * fr.sin()
        att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(dh), dim=-1) @ v
        out = att.permute(0, 2, 3, 1, 4).reshape(B, T, C) @ W[p + "to_out.weight"].t()
        return out.transpose(1, 2) + x

    def _res_unit(self, x, p, dilation):
        W, w = self.W, self.w
        C = x.shape[1] if self.cfg.depthwise else 1
        y = snake(x, W[p + "0.alpha"])
        y = F.conv1d(y, w[p + "1"], W[p + "1.bias"], dilation=dilation, padding=3 * dilation, groups=C)
        y = snake(y, W[p + "2.alpha"])
        y = F.conv1d(y, w[p + "3"], W[p + "3.bias"])
        return x + y

    def decode_latents(self, z: torch.Tensor, noise: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
        cfg, W, w = self.cfg, self.W, self.w
        d = "decoder.model."
        if cfg.depthwise:
            x = F.conv1d(z, w[d + "0"], W[d + "0.bias"], padding=3, groups=cfg.latent_dim)
            x = F.conv1d(x, w[d + "1"], W[d + "1.bias"])
        else:
            x = F.conv1d(z, w[d + "0"], W[d + "0.bias"], padding=3)
        if cfg.attn_window_size is not None:
            x = self.local_mha(x, f"{d}{cfg.first_block - 1}.")
        for bi, r in enumerate(cfg.rates):
            b = f"{d}{cfg.first_block + bi}.block."
            x = snake(x, W[b + "0.alpha"])
            x = F.conv_transpose1d(x, w[b + "1"], W[b + "1.bias"], stride=r, padding=math.ceil(r / 2), output_padding=r % 2)
            j = 2
            if cfg.noise:
                h = F.conv1d(x, w[b + "2.linear"])
                n = noise[bi] if noise is not None else torch.zeros(x.shape[0], 1, x.shape[2])
                x = x + n * h
                j = 3
            for u, dil in enumerate((1, 3, 9)):
                x = self._res_unit(x, f"{b}{j + u}.block.", dil)
        n = cfg.first_block + len(cfg.rates)
        x = snake(x, W[f"{d}{n}.alpha"])
        x = F.conv1d(x, w[f"{d}{n + 1}"], W[f"{d}{n + 1}.bias"], padding=3)
        return torch.tanh(x)

    def decode(self, codes: List[torch.Tensor], noise: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
        return self.decode_latents(self.from_codes(codes), noise)


# ---- Orpheus token layout (model/orpheus.py:479-507) ---------------------------------------------------------------
def orpheus_codes(token_ids: torch.Tensor, codebook_size=4096):
    """token_ids [B, 28] LM ids of 4 frames x 7 -> the three SNAC code levels [B,4], [B,8], [B,16]."""
    mf = (token_ids.view(-1, 4, 7).long() - 128256 - 10) % codebook_size
    return [mf[:, :, 0], mf[:, :, [1, 4]].reshape(-1, 8), mf[:, :, [2, 3, 5, 6]].reshape(-1, 16)]


def orpheus_postprocess(ref: SnacRef, token_ids: torch.Tensor, noise=None) -> torch.Tensor:
    return ref.decode(orpheus_codes(token_ids, ref.cfg.codebook_size), noise)[:, :, 2048:4096]
