"""CPU oracle for the HiFT vocoder (mel -> waveform) of the CosyVoice2 / GLM-4-Voice detokenizers.  TEST INFRASTRUCTURE ONLY.

Restates, in plain torch-CPU fp32 on explicit (weight-norm-folded) tensors:
  ConvRNNF0Predictor                  /root/reference/vox_serve/tokenizer/hifigan.py:394-426   5 x (conv k3 + ELU), linear, abs
  SineGen2 / SourceModuleHnNSF2       hifigan.py:251-391   harmonics of the upsampled f0: phase accumulated at the mel-frame rate,
                                                          linearly interpolated back to the sample rate; voiced / unvoiced noise;
                                                          tanh(linear) merge of the 9 harmonics
  HiFTGenerator._stft / decode / _istft_graph_safe / forward_chunk    hifigan.py:542-665
  Snake, ResBlock                     hifigan.py:45-146
  fade_in_out                         /root/reference/vox_serve/tokenizer/cosyvoice2.py:46-54
The reference draws the harmonics' initial phases with torch.rand and the additive noise with torch.randn_like (irreproducible);
the contract here is explicit tensors `rand_ini [B, H+1]` and `noise [B, L, H+1]` — `make_noise` below is the seeded stream the
library generates on the device (Philox4x32-10, counter = (element, stream, 0, 0), key = seed; uniform = word0 >> 8 / 2^24,
normal = Box-Muller as in oracle/snac_ref.py::philox_noise), and the parity fixtures inject the same tensors into the reference
module (tests/golden/make_goldens.py::g11_hift).
Pinned: tests/test_oracle_goldens.py::test_hift_* against g11 (reference HiFTGenerator, tiny + CosyVoice2 size).
"""
import math
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .snac_ref import _philox4x32_10, fold_weight_norm, philox_noise


@dataclass
class HiftCfg:
    """CosyVoice2's HiFTGenerator (tokenizer/cosyvoice2.py:840-848 over the defaults of hifigan.py:435-455)."""
    in_channels: int = 80
    base_channels: int = 512
    nb_harmonics: int = 8
    sampling_rate: int = 24000
    nsf_alpha: float = 0.1
    nsf_sigma: float = 0.003
    voiced_threshold: float = 10.0
    upsample_rates: Sequence[int] = (8, 5, 3)
    upsample_kernel_sizes: Sequence[int] = (16, 11, 7)
    n_fft: int = 16
    hop_len: int = 4
    resblock_kernel_sizes: Sequence[int] = (3, 7, 11)
    resblock_dilations: Sequence[int] = (1, 3, 5)
    source_resblock_kernel_sizes: Sequence[int] = (7, 7, 11)
    lrelu_slope: float = 0.1
    audio_limit: float = 0.99
    f0_channels: int = 512

    @property
    def upsample_scale(self) -> int:            # samples per mel frame
        return int(np.prod(self.upsample_rates)) * self.hop_len

    @property
    def n_bins(self) -> int:
        return self.n_fft // 2 + 1


def tiny_hift_cfg() -> HiftCfg:
    # (channels 128 / 64 / 32 after the three stages: the implicit-GEMM kernels take input channels in multiples of 32)
    return HiftCfg(in_channels=32, base_channels=256, upsample_rates=(4, 3, 2), upsample_kernel_sizes=(8, 7, 4), f0_channels=64)


def _source_down_geometry(cfg: HiftCfg):
    """(kernel, stride, padding) of source_downs[i]   (hifigan.py:497-510)"""
    down = [1] + list(cfg.upsample_rates[::-1][:-1])
    cum = np.cumprod(down)[::-1]
    return [(1, 1, 0) if u == 1 else (int(u) * 2, int(u), int(u) // 2) for u in cum]


def param_shapes(cfg: HiftCfg) -> Dict[str, tuple]:
    """Reference state_dict names (weight_norm parametrizations: original0 = g, original1 = v)."""
    s = {}

    def wn(name, shape, g0=None):
        s[name + ".parametrizations.weight.original0"] = (shape[0] if g0 is None else g0, 1, 1)
        s[name + ".parametrizations.weight.original1"] = shape
        s[name + ".bias"] = (shape[0],)

    H1 = cfg.nb_harmonics + 1
    s["m_source.l_linear.weight"], s["m_source.l_linear.bias"] = (1, H1), (1,)
    wn("conv_pre", (cfg.base_channels, cfg.in_channels, 7))
    nst = len(cfg.upsample_rates)
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        cin, cout = cfg.base_channels // 2 ** i, cfg.base_channels // 2 ** (i + 1)
        s[f"ups.{i}.parametrizations.weight.original0"] = (cin, 1, 1)          # ConvTranspose1d: weight_norm over dim 0 = input channels
        s[f"ups.{i}.parametrizations.weight.original1"] = (cin, cout, k)
        s[f"ups.{i}.bias"] = (cout,)

    def resblock(p, ch, k):
        for j in range(len(cfg.resblock_dilations)):
            wn(f"{p}.convs1.{j}", (ch, ch, k))
            wn(f"{p}.convs2.{j}", (ch, ch, k))
            s[f"{p}.activations1.{j}.alpha"] = (ch,)
            s[f"{p}.activations2.{j}.alpha"] = (ch,)

    for i, (k, st, pd) in enumerate(_source_down_geometry(cfg)):
        ch = cfg.base_channels // 2 ** (i + 1)
        s[f"source_downs.{i}.weight"], s[f"source_downs.{i}.bias"] = (ch, cfg.n_fft + 2, k), (ch,)
        resblock(f"source_resblocks.{i}", ch, cfg.source_resblock_kernel_sizes[i])
    for i in range(nst):
        ch = cfg.base_channels // 2 ** (i + 1)
        for j, k in enumerate(cfg.resblock_kernel_sizes):
            resblock(f"resblocks.{i * len(cfg.resblock_kernel_sizes) + j}", ch, k)
    wn("conv_post", (cfg.n_fft + 2, cfg.base_channels // 2 ** nst, 7))
    cin = cfg.in_channels
    for li in range(5):
        wn(f"f0_predictor.condnet.{2 * li}", (cfg.f0_channels, cin, 3))
        cin = cfg.f0_channels
    s["f0_predictor.classifier.weight"], s["f0_predictor.classifier.bias"] = (1, cfg.f0_channels), (1,)
    return s


def random_hift_weights(cfg: HiftCfg, seed=0) -> Dict[str, torch.Tensor]:
    """fp32 CPU tensors with bf16-representable values; gains keep the activations O(1) through the 72 stacked convs, the log-magnitude
    head inside exp's comfortable range and the predicted f0 around 100-400 Hz (voiced and unvoiced frames both occur)."""
    g = torch.Generator().manual_seed(seed)
    W = {}
    shapes = param_shapes(cfg)
    for k, shp in shapes.items():
        if k.endswith("alpha"):
            t = 0.5 + torch.rand(shp, generator=g)
        elif k.endswith("original0"):
            t = 0.7 + 0.6 * torch.rand(shp, generator=g)
        elif k.endswith("bias"):
            t = 0.02 * torch.randn(shp, generator=g)
        else:
            t = torch.randn(shp, generator=g)
        W[k] = t
    for k in list(W):
        if k.endswith("original0"):
            name = k.split(".parametrizations")[0]
            if ".convs2." in name:
                W[k] = W[k] * 0.25                                   # residual branch output
            elif name == "conv_post":
                W[k] = W[k] * 0.35
            elif name.startswith("f0_predictor"):
                W[k] = W[k] * 1.2
        if k.startswith("source_downs") and k.endswith("weight"):
            W[k] = W[k] / math.sqrt(W[k].shape[1] * W[k].shape[2])
    W["f0_predictor.classifier.weight"] = W["f0_predictor.classifier.weight"] * (120.0 / math.sqrt(cfg.f0_channels))
    W["f0_predictor.classifier.bias"] = torch.full((1,), 60.0)
    W["m_source.l_linear.weight"] = W["m_source.l_linear.weight"] * 0.6
    W["conv_post.bias"][: cfg.n_bins] += 1.5                          # log-magnitude offset: waveform RMS ~ 0.1
    return {k: v.to(torch.bfloat16).float() for k, v in W.items()}


# ---- seeded noise contract ------------------------------------------------------------------------------------------
def philox_uniform(seed: int, stream: int, n: int) -> np.ndarray:
    """n uniforms in [0, 1): element i = (word 0 of Philox(counter = (i, stream, 0, 0), key = seed) >> 8) / 2^24."""
    ctr = np.zeros((n, 4), np.uint32)
    ctr[:, 0] = np.arange(n, dtype=np.uint32)
    ctr[:, 1] = np.uint32(stream)
    r = _philox4x32_10(ctr, np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], np.uint32))
    return ((r[:, 0] >> np.uint32(8)).astype(np.float32) / np.float32(16777216.0)).astype(np.float32)


def make_noise(cfg: HiftCfg, B: int, T: int, seed: int, first_stream: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """(rand_ini [B, H+1] with column 0 = 0, noise [B, T * upsample_scale, H+1]); request b uses streams first_stream + 2b (uniform)
    and first_stream + 2b + 1 (normal); the normal of (sample t, harmonic h) is element t * (H+1) + h of its stream."""
    H1, L = cfg.nb_harmonics + 1, T * cfg.upsample_scale
    ini = np.stack([philox_uniform(seed, first_stream + 2 * b, H1) for b in range(B)])
    ini[:, 0] = 0.0
    nz = np.stack([philox_noise(seed, first_stream + 2 * b + 1, L * H1).reshape(L, H1) for b in range(B)])
    return torch.from_numpy(ini), torch.from_numpy(nz)


# ---- the vocoder ----------------------------------------------------------------------------------------------------
def snake(x, alpha):
    a = alpha.view(1, -1, 1)
    return x + (1.0 / (a + 1e-9)) * torch.sin(x * a).pow(2)


class HiftRef:
    def __init__(self, cfg: HiftCfg, W: Dict[str, torch.Tensor]):
        self.cfg, self.W = cfg, {k: v.float() for k, v in W.items()}
        self.w = {k.split(".parametrizations")[0]: fold_weight_norm(self.W, k.split(".parametrizations")[0])
                  for k in W if k.endswith("original0")}
        n = cfg.n_fft
        # scipy.signal.get_window("hann", n, fftbins=True): the periodic Hann window
        self.window = (0.5 - 0.5 * torch.cos(2.0 * math.pi * torch.arange(n, dtype=torch.float64) / n)).float()

    # hifigan.py:423-426
    def f0_predict(self, mel: torch.Tensor) -> torch.Tensor:
        x = mel
        for li in range(5):
            p = f"f0_predictor.condnet.{2 * li}"
            x = F.elu(F.conv1d(x, self.w[p], self.W[p + ".bias"], padding=1))
        y = F.linear(x.transpose(1, 2), self.W["f0_predictor.classifier.weight"], self.W["f0_predictor.classifier.bias"])
        return y.squeeze(-1).abs()

    # hifigan.py:291-343, 379-391 (the non-pulse branch) on f0 [B, T] at the mel-frame rate
    def source(self, f0: torch.Tensor, rand_ini: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
        c = self.cfg
        up, H1 = c.upsample_scale, c.nb_harmonics + 1
        f0s = f0[:, :, None].repeat_interleave(up, dim=1)                                   # nn.Upsample (nearest) [B, L, 1]
        fn = f0s * torch.arange(1, H1 + 1, dtype=f0.dtype)                                  # [B, L, H+1]
        rad = (fn / c.sampling_rate) % 1
        rad[:, 0, :] = rad[:, 0, :] + rand_ini
        rad = F.interpolate(rad.transpose(1, 2), scale_factor=1 / up, mode="linear").transpose(1, 2)
        phase = torch.cumsum(rad, dim=1) * 2 * np.pi
        phase = F.interpolate(phase.transpose(1, 2) * up, scale_factor=up, mode="linear").transpose(1, 2)
        sines = torch.sin(phase) * c.nsf_alpha
        uv = (f0s > c.voiced_threshold).to(f0.dtype)
        amp = uv * c.nsf_sigma + (1 - uv) * c.nsf_alpha / 3
        sw = sines * uv + amp * noise
        merged = torch.tanh(F.linear(sw, self.W["m_source.l_linear.weight"], self.W["m_source.l_linear.bias"]))
        return merged.transpose(1, 2)                                                       # [B, 1, L]

    def stft(self, s: torch.Tensor) -> torch.Tensor:
        """s [B, L] -> [B, 2 * n_bins, L / hop + 1] (real rows, then imaginary rows)   hifigan.py:542-552"""
        c = self.cfg
        spec = torch.stft(s, c.n_fft, c.hop_len, c.n_fft, window=self.window, return_complex=True)
        return torch.cat([spec.real, spec.imag], dim=1)

    def istft(self, mag: torch.Tensor, phase: torch.Tensor) -> torch.Tensor:
        """hifigan.py:554-594: irfft of every frame, window, overlap-add, divide by the overlap-added squared window, trim n_fft/2."""
        c = self.cfg
        mag = torch.clip(mag, max=1e2)
        comp = torch.complex(mag * torch.cos(phase), mag * torch.sin(phase))
        nfr = comp.shape[2]
        L = c.n_fft + c.hop_len * (nfr - 1)
        frames = torch.fft.irfft(comp.permute(0, 2, 1), n=c.n_fft) * self.window
        out = torch.zeros(comp.shape[0], L)
        den = torch.zeros(L)
        for t in range(nfr):
            out[:, t * c.hop_len: t * c.hop_len + c.n_fft] += frames[:, t]
            den[t * c.hop_len: t * c.hop_len + c.n_fft] += self.window ** 2
        den = torch.where(den > 1e-8, den, torch.ones_like(den))
        return (out / den)[:, c.n_fft // 2: L - c.n_fft // 2]

    def _resblock(self, x, p, k):
        W, w = self.W, self.w
        for j, d in enumerate(self.cfg.resblock_dilations):
            xt = snake(x, W[f"{p}.activations1.{j}.alpha"])
            xt = F.conv1d(xt, w[f"{p}.convs1.{j}"], W[f"{p}.convs1.{j}.bias"], dilation=d, padding=(k * d - d) // 2)
            xt = snake(xt, W[f"{p}.activations2.{j}.alpha"])
            xt = F.conv1d(xt, w[f"{p}.convs2.{j}"], W[f"{p}.convs2.{j}.bias"], padding=(k - 1) // 2)
            x = xt + x
        return x

    # hifigan.py:596-628
    def decode(self, mel: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
        c, W, w = self.cfg, self.W, self.w
        s_stft = self.stft(s.squeeze(1))
        x = F.conv1d(mel, w["conv_pre"], W["conv_pre.bias"], padding=3)
        nk, nst = len(c.resblock_kernel_sizes), len(c.upsample_rates)
        geo = _source_down_geometry(c)
        for i, (u, k) in enumerate(zip(c.upsample_rates, c.upsample_kernel_sizes)):
            x = F.leaky_relu(x, c.lrelu_slope)
            x = F.conv_transpose1d(x, w[f"ups.{i}"], W[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
            if i == nst - 1:
                x = F.pad(x, (1, 0), mode="reflect")
            kk, st, pd = geo[i]
            si = F.conv1d(s_stft, W[f"source_downs.{i}.weight"], W[f"source_downs.{i}.bias"], stride=st, padding=pd)
            si = self._resblock(si, f"source_resblocks.{i}", c.source_resblock_kernel_sizes[i])
            x = x + si
            xs = None
            for j, rk in enumerate(c.resblock_kernel_sizes):
                y = self._resblock(x, f"resblocks.{i * nk + j}", rk)
                xs = y if xs is None else xs + y
            x = xs / nk
        x = F.leaky_relu(x)                                   # (default slope 0.01, as in the reference)
        x = F.conv1d(x, w["conv_post"], W["conv_post.bias"], padding=3)
        mag = torch.exp(x[:, : c.n_bins])
        ph = torch.sin(x[:, c.n_bins:])
        return torch.clamp(self.istft(mag, ph), -c.audio_limit, c.audio_limit)

    # hifigan.py:641-665 (cache_source None, as CosyVoice2Decoder.decode_chunk calls it)
    def forward_chunk(self, mel: torch.Tensor, rand_ini: torch.Tensor, noise: torch.Tensor):
        """mel [B, in_channels, T] -> (wav [B, T * upsample_scale], source [B, 1, T * upsample_scale])"""
        s = self.source(self.f0_predict(mel), rand_ini, noise)
        return self.decode(mel, s), s


def fade_in_out(new: torch.Tensor, old: torch.Tensor, window: torch.Tensor) -> torch.Tensor:
    """cosyvoice2.py:46-54: the first half-window of `new` cross-fades with the tail of `old`."""
    n = window.shape[0] // 2
    out = new.clone()
    out[..., :n] = out[..., :n] * window[:n] + old[..., -n:] * window[n:]
    return out
