"""Speaker encoder of a Qwen3-TTS voice-clone request on the native engine (vox_spkenc_*).

Mirrors mel_spectrogram + Qwen3TTSSpeakerEncoder of /root/reference/vox_serve/model/qwen3_tts.py:21-88, 835-891 as used by
Qwen3TTSModel._extract_speaker_embedding (:1288-1328): 24 kHz clip -> log-mel (n_fft 1024, hop 256, fmin 0, fmax 12000) ->
ECAPA-TDNN -> x-vector [enc_dim].  The weights keep the reference state_dict names (`speaker_encoder.` prefix stripped).

The mel filterbank is librosa's (`librosa.filters.mel`, Slaney scale + normalisation): a constant table, built here on the host
from its published definition (librosa is not a dependency of this package).
"""
import ctypes
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch

from .. import _native as N
from ..tokenizer.qwen3_codec import ConvW


@dataclass
class Qwen3TTSSpeakerEncoderConfig:
    """qwen3_tts.py:91-102 (+ the front-end constants of :1306-1320)."""
    enc_dim: int = 2048
    sample_rate: int = 24000
    mel_dim: int = 128
    enc_channels: List[int] = field(default_factory=lambda: [512, 512, 512, 512, 1536])
    enc_kernel_sizes: List[int] = field(default_factory=lambda: [5, 3, 3, 3, 1])
    enc_dilations: List[int] = field(default_factory=lambda: [1, 2, 3, 4, 1])
    enc_res2net_scale: int = 8
    enc_se_channels: int = 128
    enc_attention_channels: int = 128
    n_fft: int = 1024
    hop_size: int = 256
    fmin: float = 0.0
    fmax: float = 12000.0


class SpkBlockW(ctypes.Structure):
    _fields_ = [("tdnn1", ConvW), ("res2", ConvW * 7), ("tdnn2", ConvW), ("se1", ConvW), ("se2", ConvW)]


class SpkWeights(ctypes.Structure):
    _fields_ = [("mel_basis", ctypes.c_void_p), ("window", ctypes.c_void_p), ("conv0", ConvW), ("blocks", SpkBlockW * 4),
                ("mfa", ConvW), ("asp_tdnn", ConvW), ("asp_conv", ConvW), ("fc", ConvW)]


class SpkConfigC(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("n_mels", "n_mels_padded", "n_fft", "hop", "n_blocks", "channels", "scale", "se_channels",
                                              "mfa_channels", "att_channels", "enc_dim", "kernel0", "dilation0")] + \
               [("kernels", ctypes.c_int32 * 4), ("dilations", ctypes.c_int32 * 4)]


def _bind(L):
    if getattr(L, "_spk_bound", False):
        return
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.vox_spkenc_create.restype, L.vox_spkenc_create.argtypes = ci, [vp, ctypes.POINTER(SpkConfigC), ctypes.POINTER(SpkWeights), ci,
                                                                    ctypes.POINTER(vp)]
    L.vox_spkenc_destroy.restype, L.vox_spkenc_destroy.argtypes = None, [vp]
    L.vox_spkenc_embed.restype, L.vox_spkenc_embed.argtypes = ci, [vp, vp, vp, ci, vp, ctypes.POINTER(ctypes.c_int32), vp]
    L._spk_bound = True


def param_shapes(c: Qwen3TTSSpeakerEncoderConfig) -> Dict[str, tuple]:
    """state_dict names and shapes of Qwen3TTSSpeakerEncoder (qwen3_tts.py:835-878)."""
    S: Dict[str, tuple] = {}

    def conv(name, cout, cin, k):
        S[name + ".weight"], S[name + ".bias"] = (cout, cin, k), (cout,)

    ch, ks = c.enc_channels, c.enc_kernel_sizes
    conv("blocks.0.conv", ch[0], c.mel_dim, ks[0])
    for i in range(1, len(ch) - 1):
        p = f"blocks.{i}"
        conv(p + ".tdnn1.conv", ch[i], ch[i - 1], 1)
        for j in range(c.enc_res2net_scale - 1):
            conv(f"{p}.res2net_block.blocks.{j}.conv", ch[i] // c.enc_res2net_scale, ch[i] // c.enc_res2net_scale, ks[i])
        conv(p + ".tdnn2.conv", ch[i], ch[i], 1)
        conv(p + ".se_block.conv1", c.enc_se_channels, ch[i], 1)
        conv(p + ".se_block.conv2", ch[i], c.enc_se_channels, 1)
    conv("mfa.conv", ch[-1], ch[-1], ks[-1])
    conv("asp.tdnn.conv", c.enc_attention_channels, ch[-1] * 3, 1)
    conv("asp.conv", ch[-1], c.enc_attention_channels, 1)
    conv("fc", c.enc_dim, ch[-1] * 2, 1)
    return S


def slaney_mel_filterbank(sr: int, n_fft: int, n_mels: int, fmin: float, fmax: float) -> np.ndarray:
    """librosa.filters.mel(sr=, n_fft=, n_mels=, fmin=, fmax=) with its defaults (htk=False, norm="slaney"): [n_mels, n_fft/2+1] float32."""
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0

    def to_mel(f):
        f = np.asarray(f, np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, f / f_sp)

    def to_hz(m):
        m = np.asarray(m, np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    freqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    edges = to_hz(np.linspace(to_mel(fmin), to_mel(fmax), n_mels + 2))
    width = np.diff(edges)
    ramps = edges[:, None] - freqs[None, :]
    tri = np.maximum(0, np.minimum(-ramps[:-2] / width[:-1, None], ramps[2:] / width[1:, None]))
    return (tri * (2.0 / (edges[2:] - edges[:-2]))[:, None]).astype(np.float32)


class Qwen3TTSSpeakerEncoder:
    def __init__(self, weights: Dict[str, torch.Tensor], config: Optional[Qwen3TTSSpeakerEncoderConfig] = None, device="cuda",
                 max_seconds: float = 30.0):
        self.cfg = c = config or Qwen3TTSSpeakerEncoderConfig()
        ch, ks, dl = c.enc_channels, c.enc_kernel_sizes, c.enc_dilations
        nb = len(ch) - 2
        if not (1 <= nb <= 4) or len(set(ch[:-1])) != 1 or ch[-1] != nb * ch[0] or ks[-1] != 1 or c.enc_res2net_scale > 8:
            raise ValueError("Qwen3TTSSpeakerEncoder: 1-4 SE-Res2Net blocks of equal width, aggregation width = their sum, scale <= 8")
        self.device = torch.device(device)
        self.L = N.lib()
        _bind(self.L)
        self._keep = []
        W, dev = weights, self.device

        def f32(t):
            t = torch.as_tensor(t).detach().to(device=dev, dtype=torch.float32).contiguous()
            self._keep.append(t)
            return t.data_ptr()

        def conv(name, cin_pad=0):          # Conv1d [Cout, Cin, k] -> one bf16 plane of taps [k][Cout][Cin] (the checkpoint is bf16)
            w = W[name + ".weight"].float().permute(2, 0, 1)
            if cin_pad:
                w = torch.nn.functional.pad(w, (0, cin_pad))
            pl = w.to(torch.bfloat16).to(dev).contiguous()
            self._keep.append(pl)
            return ConvW(pl.data_ptr(), f32(W[name + ".bias"]), pl.shape[0], pl.shape[1], pl.shape[2], 0)

        mp = c.mel_dim + (-c.mel_dim) % 32
        sw = SpkWeights()
        sw.mel_basis = f32(slaney_mel_filterbank(c.sample_rate, c.n_fft, c.mel_dim, c.fmin, c.fmax))
        sw.window = f32(torch.hann_window(c.n_fft, dtype=torch.float32))
        sw.conv0 = conv("blocks.0.conv", mp - c.mel_dim)
        for b in range(nb):
            p = f"blocks.{b + 1}"
            bw = sw.blocks[b]
            bw.tdnn1, bw.tdnn2 = conv(p + ".tdnn1.conv"), conv(p + ".tdnn2.conv")
            for j in range(c.enc_res2net_scale - 1):
                bw.res2[j] = conv(f"{p}.res2net_block.blocks.{j}.conv")
            bw.se1, bw.se2 = conv(p + ".se_block.conv1"), conv(p + ".se_block.conv2")
        sw.mfa, sw.asp_tdnn, sw.asp_conv, sw.fc = conv("mfa.conv"), conv("asp.tdnn.conv"), conv("asp.conv"), conv("fc")
        i4 = lambda xs: (ctypes.c_int32 * 4)(*(list(xs) + [1] * (4 - len(xs))))
        sc = SpkConfigC(c.mel_dim, mp, c.n_fft, c.hop_size, nb, ch[0], c.enc_res2net_scale, c.enc_se_channels, ch[-1],
                        c.enc_attention_channels, c.enc_dim, ks[0], dl[0], i4(ks[1:-1]), i4(dl[1:-1]))
        self.max_samples = int(max_seconds * c.sample_rate)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            N.check(self.L.vox_spkenc_create(N.ctx(), ctypes.byref(sc), ctypes.byref(sw), self.max_samples, ctypes.byref(h)))
        self.h, self._sw = h, sw

    def forward(self, audio: torch.Tensor, return_mel: bool = False):
        """audio [N] float (24 kHz, [-1, 1]) -> x-vector [enc_dim] fp32 on the encoder's device (and the log-mel [T, mel_dim])."""
        c = self.cfg
        a = torch.as_tensor(audio).reshape(-1).to(self.device, torch.float32).contiguous()
        if a.numel() > self.max_samples:
            raise ValueError(f"reference clip of {a.numel()} samples exceeds the encoder capacity ({self.max_samples})")
        emb = torch.empty(c.enc_dim, dtype=torch.float32, device=self.device)
        T = (a.numel() + 2 * ((c.n_fft - c.hop_size) // 2) - c.n_fft) // c.hop_size + 1
        mel = torch.empty(max(T, 1), c.mel_dim, dtype=torch.float32, device=self.device) if return_mel else None
        nf = ctypes.c_int32(0)
        with torch.cuda.device(self.device):
            N.check(self.L.vox_spkenc_embed(self.h, N.stream(), a.data_ptr(), a.numel(), mel.data_ptr() if return_mel else None,
                                            ctypes.byref(nf), emb.data_ptr()))
            torch.cuda.current_stream().synchronize()
        return (emb, mel) if return_mel else emb

    __call__ = forward

    def close(self):
        if getattr(self, "h", None):
            self.L.vox_spkenc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
