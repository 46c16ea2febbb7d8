"""SNAC decoder on libvoxhip (drop-in surface of the reference's `SNAC.decode`, /root/reference/vox_serve/tokenizer/snac.py:438-441,
as OrpheusModel.postprocess uses it, model/orpheus.py:483-507): `decode(codes: List[Tensor[B, T/stride_i]]) -> audio [B, 1, T*hop]`.

Weights are the reference checkpoint's state_dict names (quantizer.quantizers.i.{codebook,out_proj}, decoder.model.*, weight-norm
parametrizations original0 = g / original1 = v, or already-folded `.weight`).  Packing is layout + exact algebra only:
  * weight norm folded: w = g * v / ||v||                                            (torch weight_norm, dim 0)
  * per VQ level a table out_proj(codebook) + bias [codebook_size][latent]          (from_codes, snac.py:350-357)
  * 1x1 / transposed convs -> implicit-GEMM taps; an fp32 weight is carried as TWO bf16 planes (hi + residual = 16
    significand bits) stacked as extra taps, so the matrix-core products are exact to 2^-17 relative
  * ConvTranspose1d [Cin, Cout, 2r] (stride r, padding r/2) -> two taps writing r*Cout values per input row
  * Snake1d alpha -> (alpha, 1/(alpha + 1e-9))
NoiseBlock's noise: a seeded Philox stream generated on the device (see include/voxhip.h), or a tensor handed in (tests).
Variants: depthwise convs without attention (hubertsiuzdak/snac_24khz, what the reference's Orpheus plugin loads) and the module's
dense-conv / LocalMHA forms (the 32 / 44 kHz checkpoints' structure: `depthwise=False`, `attn_window_size` <= 32 frames, the frame
count a multiple of the window as in the reference: snac.py:38-40).
"""
import ctypes
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch

from .. import _native as N
from .base import device_bound
from .qwen3_codec import ConvW, SnakeW


@dataclass
class SNACConfig:
    """config.json of hubertsiuzdak/snac_24khz (the checkpoint OrpheusModel loads, model/orpheus.py:246-249)"""
    sampling_rate: int = 24000
    latent_dim: int = 768                      # encoder_dim 48 * 2^len(encoder_rates)
    decoder_dim: int = 1024
    decoder_rates: List[int] = field(default_factory=lambda: [8, 8, 4, 2])
    codebook_size: int = 4096
    codebook_dim: int = 8
    vq_strides: List[int] = field(default_factory=lambda: [4, 2, 1])
    noise: bool = True
    depthwise: bool = True
    attn_window_size: Optional[int] = None

    @property
    def hop(self) -> int:
        return int(math.prod(self.decoder_rates))


class SnacResW(ctypes.Structure):
    _fields_ = [("act1", SnakeW), ("act2", SnakeW), ("dw_w", ctypes.c_void_p), ("dw_b", ctypes.c_void_p), ("pw", ConvW), ("dense", ConvW)]


class SnacBlockW(ctypes.Structure):
    _fields_ = [("snake0", SnakeW), ("tconv", ConvW), ("noise", ConvW), ("res", SnacResW * 3)]


class SnacWeights(ctypes.Structure):
    _fields_ = [("tab", ctypes.c_void_p * 4), ("dw0_w", ctypes.c_void_p), ("dw0_b", ctypes.c_void_p), ("pw0", ConvW),
                ("blocks", SnacBlockW * 4), ("final_snake", SnakeW), ("final_w", ctypes.c_void_p), ("final_b", ctypes.c_float),
                ("conv0", ConvW), ("attn_ln_w", ctypes.c_void_p), ("attn_ln_b", ctypes.c_void_p), ("attn_qkv", ConvW), ("attn_out", ConvW)]


class SnacConfigC(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("latent_dim", "decoder_dim", "codebook_size", "n_levels")] + \
               [("vq_strides", ctypes.c_int32 * 4), ("rates", ctypes.c_int32 * 4), ("noise", ctypes.c_int32), ("attn_window", ctypes.c_int32)]


def _bind(L):
    if getattr(L, "_snac_bound", False):
        return
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.vox_snac_create.restype, L.vox_snac_create.argtypes = ci, [vp, ctypes.POINTER(SnacConfigC), ctypes.POINTER(SnacWeights), ci, ci,
                                                                ctypes.POINTER(vp)]
    L.vox_snac_destroy.restype, L.vox_snac_destroy.argtypes = None, [vp]
    L.vox_snac_decode.restype = ci
    L.vox_snac_decode.argtypes = [vp, vp, vp, ci, ci, vp, ctypes.c_uint64, vp, vp, ci, ci]
    L._snac_bound = True


def _folded(W: Dict[str, torch.Tensor], name: str) -> torch.Tensor:
    """conv weight `name` with weight norm folded (accepts the parametrized and the plain form)"""
    g = W.get(name + ".parametrizations.weight.original0")
    if g is None:
        if name + ".weight_g" in W:
            g, v = W[name + ".weight_g"], W[name + ".weight_v"]
        else:
            return W[name + ".weight"].float().cpu()
    else:
        v = W[name + ".parametrizations.weight.original1"]
    g, v = g.float().cpu(), v.float().cpu()
    return g * v / v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))


@device_bound
class SNACDecoder:
    def __init__(self, weights: Dict[str, torch.Tensor], config: Optional[SNACConfig] = None, device="cuda", max_batch=8, max_T=16,
                 seed: int = 0):
        self.cfg = c = config or SNACConfig()
        if c.attn_window_size is not None and (not 1 <= c.attn_window_size <= 32 or c.decoder_dim % 64):
            raise ValueError("SNACDecoder: LocalMHA windows of 1..32 frames, decoder_dim a multiple of the 64-wide heads")
        if len(c.decoder_rates) > 4 or len(c.vq_strides) > 4 or any(r % 2 for r in c.decoder_rates):
            raise ValueError("SNACDecoder: at most 4 even decoder rates / 4 VQ levels")
        self.device = torch.device(device)
        self.max_batch, self.max_T, self.seed = max_batch, max_T, seed
        self.L = N.lib()
        _bind(self.L)
        self._keep = []
        W, dev = weights, self.device

        def f32(t):
            t = t.detach().to(device=dev, dtype=torch.float32).contiguous()
            self._keep.append(t)
            return t.data_ptr()

        def conv(wp, bias=None, bias_mod=0):      # wp [taps, N, Cin] fp32 -> hi / residual bf16 planes stacked as 2 * taps
            wp = wp.float()
            hi = wp.to(torch.bfloat16)
            lo = (wp - hi.float()).to(torch.bfloat16)
            planes = torch.cat([hi, lo], 0).to(dev).contiguous()
            self._keep.append(planes)
            return ConvW(planes.data_ptr(), f32(bias) if bias is not None else None, planes.shape[0], planes.shape[1], planes.shape[2], bias_mod)

        def snake(name):
            a = W[name].float().cpu().reshape(-1)
            return SnakeW(f32(a), f32(1.0 / (a + 1e-9)))

        sw = SnacWeights()
        for i in range(len(c.vq_strides)):
            q = f"quantizer.quantizers.{i}."
            proj = _folded(W, q + "out_proj")[:, :, 0]                                     # [latent, codebook_dim]
            tab = W[q + "codebook.weight"].float().cpu() @ proj.t() + W[q + "out_proj.bias"].float().cpu()[None]
            sw.tab[i] = f32(tab)
        d = "decoder.model."
        if c.depthwise:
            sw.dw0_w, sw.dw0_b = f32(_folded(W, d + "0")[:, 0, :]), f32(W[d + "0.bias"])
            sw.pw0 = conv(_folded(W, d + "1")[:, :, 0][None], W[d + "1.bias"])
        else:                                                                            # one dense k7 conv: taps [7][Cout][Cin]
            sw.conv0 = conv(_folded(W, d + "0").permute(2, 0, 1), W[d + "0.bias"])
        first = (2 if c.depthwise else 1) + (1 if c.attn_window_size is not None else 0)   # index of the first DecoderBlock (snac.py:130-146)
        if c.attn_window_size is not None:
            a = f"{d}{first - 1}."
            sw.attn_ln_w, sw.attn_ln_b = f32(W[a + "norm.weight"]), f32(W[a + "norm.bias"])
            sw.attn_qkv = conv(W[a + "to_qkv.weight"].float().cpu()[None])
            sw.attn_out = conv(W[a + "to_out.weight"].float().cpu()[None])
        ch = c.decoder_dim
        for bi, r in enumerate(c.decoder_rates):
            b = f"{d}{first + bi}.block."
            bw = sw.blocks[bi]
            bw.snake0 = snake(b + "0.alpha")
            wt = _folded(W, b + "1")                                                     # [Cin, Cout, 2r]
            cin, cout, k = wt.shape
            taps = [wt[:, :, j0:j0 + r].permute(2, 1, 0).reshape(r * cout, cin) for j0 in range(0, k, r)]
            bw.tconv = conv(torch.stack(taps, 0), W[b + "1.bias"], bias_mod=cout)
            j = 2
            if c.noise:
                bw.noise = conv(_folded(W, b + "2.linear")[:, :, 0][None])
                j = 3
            for u in range(3):
                ru = f"{b}{j + u}.block."
                rw = bw.res[u]
                rw.act1, rw.act2 = snake(ru + "0.alpha"), snake(ru + "2.alpha")
                if c.depthwise:
                    rw.dw_w, rw.dw_b = f32(_folded(W, ru + "1")[:, 0, :]), f32(W[ru + "1.bias"])
                else:
                    rw.dense = conv(_folded(W, ru + "1").permute(2, 0, 1), W[ru + "1.bias"])
                rw.pw = conv(_folded(W, ru + "3")[:, :, 0][None], W[ru + "3.bias"])
            ch = cout
        n = first + len(c.decoder_rates)
        sw.final_snake = snake(f"{d}{n}.alpha")
        sw.final_w = f32(_folded(W, f"{d}{n + 1}")[0])                                  # [C][7]
        sw.final_b = float(W[f"{d}{n + 1}.bias"].float().item())
        rates = list(c.decoder_rates) + [0] * (4 - len(c.decoder_rates))
        strides = list(c.vq_strides) + [0] * (4 - len(c.vq_strides))
        sc = SnacConfigC(c.latent_dim, c.decoder_dim, c.codebook_size, len(c.vq_strides), (ctypes.c_int32 * 4)(*strides),
                         (ctypes.c_int32 * 4)(*rates), int(c.noise), int(c.attn_window_size or 0))
        h = ctypes.c_void_p()
        N.check(self.L.vox_snac_create(N.ctx(), ctypes.byref(sc), ctypes.byref(sw), max_batch, max_T, ctypes.byref(h)))
        self.h, self._sw = h, sw
        self._window = 0        # advances the noise streams from call to call

    sample_rate = property(lambda self: self.cfg.sampling_rate)
    hop = property(lambda self: self.cfg.hop)

    def decode(self, codes: Sequence[torch.Tensor], noise: Optional[Sequence[torch.Tensor]] = None, out_off: int = 0,
               out_len: Optional[int] = None, stream_base: Optional[torch.Tensor] = None) -> torch.Tensor:
        """codes[i] [B, T / vq_strides[i]] -> audio fp32 [B, 1, out_len] (samples [out_off, out_off + out_len) of T * hop).
        noise: optional per-stage [B, 1, T_i] tensors (else the seeded device stream)."""
        B, T = codes[-1].shape[0], codes[0].shape[1] * self.cfg.vq_strides[0]
        flat = torch.cat([c.to(self.device, torch.int32).reshape(B, -1) for c in codes], dim=1).contiguous()
        out_len = T * self.hop - out_off if out_len is None else out_len
        out = torch.empty(B, 1, out_len, dtype=torch.float32, device=self.device)
        nz = None
        if noise is not None:
            nz = torch.cat([x.to(self.device, torch.float32).reshape(-1) for x in noise]) if B <= self.max_batch else None
        ns = len(self.cfg.decoder_rates)
        for b0 in range(0, B, self.max_batch):
            nb = min(self.max_batch, B - b0)
            if noise is not None and B > self.max_batch:
                nz = torch.cat([x[b0:b0 + nb].to(self.device, torch.float32).reshape(-1) for x in noise])
            sb = stream_base
            if sb is None and noise is None:
                sb = (torch.arange(b0, b0 + nb, device=self.device, dtype=torch.int64) * ns + self._window * ns * 65536).to(torch.int32)
            elif sb is not None:
                sb = sb[b0:b0 + nb].to(self.device, torch.int32).contiguous()
            N.check(self.L.vox_snac_decode(self.h, N.stream(), flat[b0:b0 + nb].data_ptr(), nb, T, nz.data_ptr() if nz is not None else None,
                                           ctypes.c_uint64(self.seed), sb.data_ptr() if sb is not None else None, out[b0:b0 + nb].data_ptr(),
                                           out_off, out_len))
        if noise is None and stream_base is None:
            self._window += 1
        return out

    def close(self):
        if self.h:
            self.L.vox_snac_destroy(self.h)
            self.h = None
