"""HiFT vocoder on libvoxhip (drop-in surface of the reference's `HiFTGenerator.forward_chunk`,
/root/reference/vox_serve/tokenizer/hifigan.py:641-665, as CosyVoice2Decoder.decode_chunk uses it, tokenizer/cosyvoice2.py:1043-1046):
`forward_chunk(speech_feat [B, 80, T]) -> (audio [B, T * 480], source [B, 1, T * 480])`, plus `fade_in_out` (cosyvoice2.py:46-54).

Weights are the reference checkpoint's state_dict names (hift.pt with the `generator.` prefix removed: conv_pre, ups.i, source_downs.i,
source_resblocks.i, resblocks.k, conv_post, m_source.l_linear, f0_predictor.condnet.2i / classifier; weight-norm parametrizations
original0 = g / original1 = v, the older weight_g / weight_v, or already-folded `.weight`).  Packing is layout + exact algebra only:
  * weight norm folded: w = g * v / ||v||   (norm over every dim but 0, also for ConvTranspose1d)
  * Conv1d [Cout, Cin, k] -> implicit-GEMM taps [k][Cout][Cin]; an fp32 weight is carried as TWO bf16 planes (hi + residual = 16
    significand bits; THREE = 24 bits for the f0 predictor, whose output the harmonic source multiplies by 2 pi * 480 * frames) stacked
    as extra taps; Cin padded to a multiple of 32 with zero columns (the 80 mel channels)
  * ConvTranspose1d [Cin, Cout, k] (stride u, padding (k-u)/2) -> taps d = dmin..dmax writing u * Cout values per input row
  * Snake alpha -> (alpha, 1 / (alpha + 1e-9))
The harmonic source's additive noise: a seeded Philox stream generated on the device (include/voxhip.h), or a tensor handed in (tests).
"""
import ctypes
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from .. import _native as N
from .base import device_bound
from .qwen3_codec import ConvW, SnakeW


@dataclass
class HiFTConfig:
    """CosyVoice2's HiFTGenerator (tokenizer/cosyvoice2.py:840-848 over the defaults of hifigan.py:435-455)"""
    in_channels: int = 80
    base_channels: int = 512
    nb_harmonics: int = 8
    sampling_rate: int = 24000
    nsf_alpha: float = 0.1
    nsf_sigma: float = 0.003
    nsf_voiced_threshold: float = 10.0
    upsample_rates: List[int] = field(default_factory=lambda: [8, 5, 3])
    upsample_kernel_sizes: List[int] = field(default_factory=lambda: [16, 11, 7])
    istft_n_fft: int = 16
    istft_hop_len: int = 4
    resblock_kernel_sizes: List[int] = field(default_factory=lambda: [3, 7, 11])
    resblock_dilation_sizes: List[int] = field(default_factory=lambda: [1, 3, 5])          # the same list for every resblock
    source_resblock_kernel_sizes: List[int] = field(default_factory=lambda: [7, 7, 11])
    lrelu_slope: float = 0.1
    audio_limit: float = 0.99
    f0_channels: int = 512
    sine_gen_v1: bool = False          # GLM-4-Voice's GLMHiFTModel (tokenizer/glm.py:2385-2594): SineGen v1, two x8 stages, 22.05 kHz

    @property
    def upsample_scale(self) -> int:
        return int(math.prod(self.upsample_rates)) * self.istft_hop_len


class HiftResblockW(ctypes.Structure):
    _fields_ = [("c1", ConvW * 3), ("c2", ConvW * 3), ("a1", SnakeW * 3), ("a2", SnakeW * 3)]


class HiftWeights(ctypes.Structure):
    _fields_ = [("f0_conv", ConvW * 5), ("f0_cls_w", ctypes.c_void_p), ("f0_cls_b", ctypes.c_float), ("src_lin_w", ctypes.c_void_p),
                ("src_lin_b", ctypes.c_float), ("conv_pre", ConvW), ("ups", ConvW * 4), ("sd_w", ctypes.c_void_p * 4),
                ("sd_b", ctypes.c_void_p * 4), ("src_rb", HiftResblockW * 4), ("rb", HiftResblockW * 12), ("conv_post", ConvW)]


class HiftConfigC(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("in_channels", "in_channels_padded", "base_channels", "nb_harmonics", "sampling_rate", "n_stages")] + \
               [("upsample_rates", ctypes.c_int32 * 4), ("upsample_kernels", ctypes.c_int32 * 4), ("n_fft", ctypes.c_int32),
                ("hop_len", ctypes.c_int32), ("n_kernels", ctypes.c_int32), ("resblock_kernels", ctypes.c_int32 * 4),
                ("dilations", ctypes.c_int32 * 3), ("source_resblock_kernels", ctypes.c_int32 * 4), ("f0_channels", ctypes.c_int32)] + \
               [(n, ctypes.c_float) for n in ("nsf_alpha", "nsf_sigma", "voiced_threshold", "lrelu_slope", "audio_limit")] + \
               [("sine_gen_v1", ctypes.c_int32)]


def _bind(L):
    if getattr(L, "_hift_bound", False):
        return
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.vox_hift_create.restype, L.vox_hift_create.argtypes = ci, [vp, ctypes.POINTER(HiftConfigC), ctypes.POINTER(HiftWeights), ci, ci,
                                                                ctypes.POINTER(vp)]
    L.vox_hift_destroy.restype, L.vox_hift_destroy.argtypes = None, [vp]
    L.vox_hift_decode.restype = ci
    L.vox_hift_decode.argtypes = [vp, vp, vp, ci, ci, vp, ctypes.c_uint64, vp, vp, vp, vp]
    L._hift_bound = True


def _folded(W: Dict[str, torch.Tensor], name: str) -> torch.Tensor:
    """conv weight `name` with weight norm folded (accepts the parametrized, the weight_g / weight_v and the plain form)"""
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


def tconv_taps(w: torch.Tensor, u: int) -> torch.Tensor:
    """ConvTranspose1d weight [Cin, Cout, k] (stride u, padding (k-u)//2) -> [n_d, u * Cout, Cin]: tap d (= dmin..dmax) reads input
    row t - d and writes output rows t * u + phi: W_d[phi * Cout + co][ci] = w[ci][co][d * u + phi + p]."""
    cin, cout, k = w.shape
    p = (k - u) // 2
    dmin, dmax = -((p + u - 1) // u), (k - 1 - p) // u
    taps = []
    for d in range(dmin, dmax + 1):
        t = torch.zeros(u, cout, cin)
        for phi in range(u):
            j = d * u + phi + p
            if 0 <= j < k:
                t[phi] = w[:, :, j].t()
        taps.append(t.reshape(u * cout, cin))
    return torch.stack(taps, 0)


@device_bound
class HiFTGenerator:
    def __init__(self, weights: Dict[str, torch.Tensor], config: Optional[HiFTConfig] = None, device="cuda", max_batch=8, max_T=64,
                 seed: int = 0):
        self.cfg = c = config or HiFTConfig()
        nst, nk = len(c.upsample_rates), len(c.resblock_kernel_sizes)
        if nst > 4 or nk > 3 or len(c.resblock_dilation_sizes) != 3 or c.base_channels % (32 << nst):
            raise ValueError("HiFTGenerator: at most 4 stages / 3 resblock kernels, 3 dilations, stage channels in multiples of 32")
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

        def conv(wp, bias=None, bias_mod=0, planes=2):     # wp [taps, N, Cin] fp32 -> bf16 planes stacked as planes * taps
            wp = wp.float()
            pad = (-wp.shape[2]) % 32
            if pad:
                wp = torch.nn.functional.pad(wp, (0, pad))
            parts, rest = [], wp
            for _ in range(planes):
                hi = rest.to(torch.bfloat16)
                parts.append(hi)
                rest = rest - hi.float()
            pl = torch.cat(parts, 0).to(dev).contiguous()
            self._keep.append(pl)
            return ConvW(pl.data_ptr(), f32(bias) if bias is not None else None, pl.shape[0], pl.shape[1], pl.shape[2], bias_mod)

        def conv1d(name, planes=2):                         # Conv1d [Cout, Cin, k] -> taps [k][Cout][Cin]
            return conv(_folded(W, name).permute(2, 0, 1), W[name + ".bias"], planes=planes)

        def snake(name):
            a = W[name].float().cpu().reshape(-1)
            return SnakeW(f32(a), f32(1.0 / (a + 1e-9)))

        def resblock(dst, p):
            for j in range(3):
                dst.c1[j], dst.c2[j] = conv1d(f"{p}.convs1.{j}"), conv1d(f"{p}.convs2.{j}")
                dst.a1[j], dst.a2[j] = snake(f"{p}.activations1.{j}.alpha"), snake(f"{p}.activations2.{j}.alpha")

        hw = HiftWeights()
        for i in range(5):
            hw.f0_conv[i] = conv1d(f"f0_predictor.condnet.{2 * i}", planes=3)
        hw.f0_cls_w = f32(W["f0_predictor.classifier.weight"].reshape(-1))
        hw.f0_cls_b = float(W["f0_predictor.classifier.bias"].float().item())
        hw.src_lin_w = f32(W["m_source.l_linear.weight"].reshape(-1))
        hw.src_lin_b = float(W["m_source.l_linear.bias"].float().item())
        hw.conv_pre = conv1d("conv_pre")
        for i, u in enumerate(c.upsample_rates):
            wt = _folded(W, f"ups.{i}")
            hw.ups[i] = conv(tconv_taps(wt, u), W[f"ups.{i}.bias"], bias_mod=wt.shape[1])
            hw.sd_w[i], hw.sd_b[i] = f32(W[f"source_downs.{i}.weight"]), f32(W[f"source_downs.{i}.bias"])
            resblock(hw.src_rb[i], f"source_resblocks.{i}")
            for j in range(nk):
                resblock(hw.rb[i * nk + j], f"resblocks.{i * nk + j}")
        hw.conv_post = conv1d("conv_post")
        i4 = lambda xs: (ctypes.c_int32 * 4)(*(list(xs) + [0] * (4 - len(xs))))
        hc = HiftConfigC(c.in_channels, c.in_channels + (-c.in_channels) % 32, c.base_channels, c.nb_harmonics, c.sampling_rate, nst,
                         i4(c.upsample_rates), i4(c.upsample_kernel_sizes), c.istft_n_fft, c.istft_hop_len, nk, i4(c.resblock_kernel_sizes),
                         (ctypes.c_int32 * 3)(*c.resblock_dilation_sizes), i4(c.source_resblock_kernel_sizes), c.f0_channels,
                         c.nsf_alpha, c.nsf_sigma, c.nsf_voiced_threshold, c.lrelu_slope, c.audio_limit, int(c.sine_gen_v1))
        h = ctypes.c_void_p()
        N.check(self.L.vox_hift_create(N.ctx(), ctypes.byref(hc), ctypes.byref(hw), max_batch, max_T, ctypes.byref(h)))
        self.h, self._hw = h, hw
        self._chunk = 0          # advances the noise streams from call to call

    sample_rate = property(lambda self: self.cfg.sampling_rate)
    upsample_scale = property(lambda self: self.cfg.upsample_scale)

    def forward_chunk(self, speech_feat: torch.Tensor, cache_source: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None,
                      stream_base: Optional[torch.Tensor] = None, rand_ini: Optional[torch.Tensor] = None):
        """speech_feat fp32 [B, in_channels, T] -> (audio fp32 [B, T * scale], source fp32 [B, 1, T * scale]).
        noise: optional [B, T * scale, H + 1] (else the seeded device stream).  cache_source must be None / empty, as the reference's
        streaming path calls it (decode_chunk passes the mels only)."""
        if cache_source is not None and cache_source.numel() > 0:
            raise NotImplementedError("HiFTGenerator.forward_chunk: a source cache is not used by the streaming path")
        mel = speech_feat.to(self.device, torch.float32).contiguous()
        B, _, T = mel.shape
        Ls = T * self.upsample_scale
        wav = torch.empty(B, Ls, dtype=torch.float32, device=self.device)
        src = torch.empty(B, 1, Ls, dtype=torch.float32, device=self.device)
        for b0 in range(0, B, self.max_batch):
            nb = min(self.max_batch, B - b0)
            nz = noise[b0:b0 + nb].to(self.device, torch.float32).contiguous() if noise is not None else None
            ri = rand_ini[b0:b0 + nb].to(self.device, torch.float32).contiguous() if rand_ini is not None else None
            sb = stream_base
            if sb is None and noise is None:
                sb = ((torch.arange(b0, b0 + nb, device=self.device, dtype=torch.int64) + self._chunk * 65536) * 2).to(torch.int32)
            elif sb is not None:
                sb = sb[b0:b0 + nb].to(self.device, torch.int32).contiguous()
            N.check(self.L.vox_hift_decode(self.h, N.stream(), mel[b0:b0 + nb].data_ptr(), nb, T, nz.data_ptr() if nz is not None else None,
                                           ctypes.c_uint64(self.seed), sb.data_ptr() if sb is not None else None, wav[b0:b0 + nb].data_ptr(),
                                           src[b0:b0 + nb].data_ptr(), ri.data_ptr() if ri is not None else None))
        if noise is None and stream_base is None:
            self._chunk += 1
        return wav, src

    def close(self):
        if self.h:
            self.L.vox_hift_destroy(self.h)
            self.h = None


def fade_in_out(fade_in_mel: torch.Tensor, fade_out_mel: torch.Tensor, window: torch.Tensor) -> torch.Tensor:
    """cosyvoice2.py:46-54 (plumbing between chunks: the head of the new chunk cross-fades with the tail of the previous one)"""
    n = int(window.shape[0] / 2)
    out = fade_in_mel.clone()
    out[..., :n] = out[..., :n] * window[:n] + fade_out_mel[..., -n:] * window[n:]
    return out
