"""ORACLE (test infrastructure, never imported by the product path): CPU restatement of the Qwen3-TTS speaker encoder —
the log-mel front end and the ECAPA-TDNN x-vector network that `Qwen3TTSModel._extract_speaker_embedding` runs over the
reference clip of a voice-clone request.

Follows /root/reference/vox_serve/model/qwen3_tts.py:
  mel_spectrogram                     :21-88      (reflect pad (n_fft-hop)/2, Hann STFT, |.| with +1e-9, mel matmul, log clamp 1e-5)
  TimeDelayNetBlock                   :470-490    (Conv1d, padding "same", reflect, + ReLU)
  Res2NetBlock                        :317-348
  SqueezeExcitationBlock              :351-378
  AttentiveStatisticsPooling          :381-467
  SqueezeExcitationRes2NetBlock       :493-532
  Qwen3TTSSpeakerEncoder.forward      :880-891
  _extract_speaker_embedding          :1288-1328  (n_fft 1024, hop 256, win 1024, fmin 0, fmax 12000, 24 kHz)

Third-party piece absent from the image: `librosa.filters.mel` (librosa is a dependency of the reference, pyproject.toml; not
installed here).  `mel_filterbank` restates its published algorithm (Slaney mel scale, triangular filters, Slaney area
normalisation, float32 result); the golden generator injects this restatement where the reference imports librosa, so the
filterbank itself is "parity unpinned" — everything downstream of it is pinned to the reference modules (g15).

Arithmetic: float32 activations, as the reference module run in fp32.  (The reference serves the encoder in bf16; the HIP
path keeps fp32 activations over the checkpoint's bf16 weights, like the codec decoder.)
"""
from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np


@dataclass
class SpkCfg:
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
    hop: int = 256
    fmin: float = 0.0
    fmax: float = 12000.0


def tiny_spk_cfg() -> SpkCfg:
    return SpkCfg(enc_dim=64, mel_dim=32, enc_channels=[64, 64, 64, 128], enc_kernel_sizes=[5, 3, 3, 1], enc_dilations=[1, 2, 3, 1],
                  enc_res2net_scale=2, enc_se_channels=32, enc_attention_channels=32)


def test_audio(seed: int, n: int, sr: int = 24000) -> np.ndarray:
    """Deterministic speech-like clip in [-1, 1]: a few gliding harmonics under a slow envelope plus a noise floor."""
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64) / sr
    f0 = 110.0 + 40.0 * np.sin(2 * np.pi * 0.7 * t + rng.uniform(0, 6.28))
    ph = 2 * np.pi * np.cumsum(f0) / sr
    y = sum(a * np.sin(h * ph + rng.uniform(0, 6.28)) for h, a in zip(range(1, 9), [0.5, 0.3, 0.2, 0.15, 0.1, 0.08, 0.05, 0.03]))
    env = 0.55 + 0.45 * np.sin(2 * np.pi * 2.3 * t)
    y = 0.4 * env * y + 0.01 * rng.standard_normal(n)
    return np.clip(y, -1, 1).astype(np.float32)


# ---- librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm="slaney") ----
def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr: int, n_fft: int, n_mels: int, fmin: float, fmax: float) -> np.ndarray:
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2), np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return (w * enorm[:, None]).astype(np.float32)


def hann_window(n: int) -> np.ndarray:           # torch.hann_window(n): periodic
    return (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n, dtype=np.float64) / n)).astype(np.float32)


def mel_spectrogram(y: np.ndarray, cfg: SpkCfg) -> np.ndarray:
    """y [N] float32 -> log-mel [T, mel_dim] float32 (time-major, the layout the encoder consumes after its transpose)."""
    pad = (cfg.n_fft - cfg.hop) // 2
    yp = np.pad(y.astype(np.float32), (pad, pad), mode="reflect")
    T = (len(yp) - cfg.n_fft) // cfg.hop + 1
    idx = np.arange(cfg.n_fft)[None, :] + cfg.hop * np.arange(T)[:, None]
    frames = yp[idx] * hann_window(cfg.n_fft)[None, :]
    spec = np.fft.rfft(frames.astype(np.float64), axis=1)
    mag = np.sqrt((spec.real ** 2 + spec.imag ** 2).astype(np.float32) + np.float32(1e-9))
    mel = mag @ mel_filterbank(cfg.sample_rate, cfg.n_fft, cfg.mel_dim, cfg.fmin, cfg.fmax).T
    return np.log(np.maximum(mel, np.float32(1e-5))).astype(np.float32)


# ---- weights (reference state_dict names of Qwen3TTSSpeakerEncoder) ----
def param_shapes(c: SpkCfg) -> Dict[str, tuple]:
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


def random_spk_weights(c: SpkCfg, seed: int = 0) -> Dict[str, np.ndarray]:
    """bf16-representable float32 weights (the checkpoint is bf16), fan-in scaled so activations stay O(1)."""
    rng = np.random.default_rng(seed)
    W = {}
    for k, shp in param_shapes(c).items():
        if k.endswith(".bias"):
            w = 0.1 * rng.standard_normal(shp)
        else:
            w = rng.standard_normal(shp) * (1.5 / np.sqrt(shp[1] * shp[2]))
        u = np.ascontiguousarray(w, dtype=np.float32).view(np.uint32).astype(np.uint64)
        u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
        W[k] = u.astype(np.uint32).view(np.float32).reshape(shp)
    return W


class SpkRef:
    def __init__(self, cfg: SpkCfg, W: Dict[str, np.ndarray]):
        self.c, self.W = cfg, {k: np.asarray(v, np.float32) for k, v in W.items()}

    def conv(self, name, x, dil=1):
        """Conv1d, padding "same" with reflect mode; x [T, Cin] -> [T, Cout]."""
        w, b = self.W[name + ".weight"], self.W[name + ".bias"]
        k = w.shape[2]
        p = dil * (k - 1) // 2
        xp = np.pad(x, ((p, p), (0, 0)), mode="reflect") if p else x
        T = x.shape[0]
        y = np.zeros((T, w.shape[0]), np.float32)
        for j in range(k):
            y += xp[j * dil: j * dil + T] @ w[:, :, j].T
        return y + b

    def tdnn(self, name, x, dil=1):
        return np.maximum(self.conv(name + ".conv", x, dil), 0)

    def se_res2net(self, p, x, dil):
        c = self.c
        h = self.tdnn(p + ".tdnn1", x)
        parts = np.split(h, c.enc_res2net_scale, axis=1)
        outs = [parts[0]]
        for i in range(1, c.enc_res2net_scale):
            inp = parts[i] if i == 1 else parts[i] + outs[-1]
            outs.append(self.tdnn(f"{p}.res2net_block.blocks.{i - 1}", inp, dil))
        h = self.tdnn(p + ".tdnn2", np.concatenate(outs, axis=1))
        m = h.mean(axis=0, keepdims=True, dtype=np.float32)
        s = np.maximum(self.conv(p + ".se_block.conv1", m), 0)
        s = 1.0 / (1.0 + np.exp(-self.conv(p + ".se_block.conv2", s)))
        return (h * s + x).astype(np.float32)

    def asp(self, x):
        T = x.shape[0]
        m = np.full((T, 1), np.float32(1.0) / np.float32(T), np.float32)

        def stats(w):
            mean = (w * x).sum(axis=0, dtype=np.float32)
            std = np.sqrt(np.maximum((w * (x - mean[None]) ** 2).sum(axis=0, dtype=np.float32), np.float32(1e-12)))
            return mean, std

        mean, std = stats(m)
        att = np.concatenate([x, np.repeat(mean[None], T, 0), np.repeat(std[None], T, 0)], axis=1)
        att = self.conv("asp.conv", np.tanh(self.tdnn("asp.tdnn", att)))
        att = np.exp(att - att.max(axis=0, keepdims=True))
        att = (att / att.sum(axis=0, keepdims=True, dtype=np.float32)).astype(np.float32)
        mean, std = stats(att)
        return np.concatenate([mean, std])[None]

    def forward(self, mels: np.ndarray) -> np.ndarray:
        """mels [T, mel_dim] -> embedding [enc_dim]."""
        c = self.c
        h = self.tdnn("blocks.0", mels.astype(np.float32), c.enc_dilations[0])
        outs = []
        for i in range(1, len(c.enc_channels) - 1):
            h = self.se_res2net(f"blocks.{i}", h, c.enc_dilations[i])
            outs.append(h)
        h = self.tdnn("mfa", np.concatenate(outs, axis=1), c.enc_dilations[-1])
        return self.conv("fc", self.asp(h))[0]

    def embed(self, audio: np.ndarray) -> np.ndarray:
        return self.forward(mel_spectrogram(audio, self.c))
