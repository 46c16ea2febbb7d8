"""CosyVoice2 token -> mel flow on libvoxhip (drop-in surface of the reference's `CausalMaskedDiffWithXvec.forward_chunk`,
/root/reference/vox_serve/tokenizer/cosyvoice_flow.py:2909-2980, as CosyVoice2Decoder.init_cache / decode_chunk use it in the plugin's
default shared-prompt mode, tokenizer/cosyvoice2.py:862-1046).

Weights are the reference checkpoint's state_dict names (flow.pt: input_embedding, spk_embed_affine_layer, encoder.*, encoder_proj,
decoder.estimator.*).  Packing is layout only: a Linear [O, I] / Conv1d [O, I, k] becomes implicit-GEMM taps [k][O][I] in bf16 (the
reference casts this module to bf16 at load time, cosyvoice2.py:837, so a real checkpoint's weights ARE bf16 values); q / k / v
projections are concatenated along the output dimension.
The time schedule (cosine t_span, Euler step sizes) and the sinusoidal timestep embedding are computed here with the reference's own
torch expressions and handed to the library; their MLPs run on the device once at construction.
"""
import ctypes
import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from .. import _native as N
from .base import device_bound
from .qwen3_codec import ConvW


@dataclass
class FlowConfig:
    """CosyVoice2 (tokenizer/cosyvoice2.py:812-835)"""
    vocab_size: int = 6561
    dim: int = 512
    mel: int = 80
    spk_embed_dim: int = 192
    enc_layers: int = 6
    up_layers: int = 4
    enc_heads: int = 8
    enc_ffn: int = 2048
    pre_lookahead_len: int = 3
    est_channels: int = 256
    est_heads: int = 8
    est_head_dim: int = 64
    est_blocks: int = 4
    est_mid_blocks: int = 12
    n_timesteps: int = 10
    inference_cfg_rate: float = 0.7
    max_cache_len: int = 128
    prefix_len: int = 16


class ConformerW(ctypes.Structure):
    _fields_ = [("qkv", ConvW), ("out", ConvW), ("pos", ConvW), ("bias_u", ctypes.c_void_p), ("bias_v", ctypes.c_void_p), ("w1", ConvW),
                ("w2", ConvW), ("ln_mha_w", ctypes.c_void_p), ("ln_mha_b", ctypes.c_void_p), ("ln_ff_w", ctypes.c_void_p),
                ("ln_ff_b", ctypes.c_void_p)]


class ResnetW(ctypes.Structure):
    _fields_ = [("conv1", ConvW), ("conv2", ConvW), ("res", ConvW), ("ln1_w", ctypes.c_void_p), ("ln1_b", ctypes.c_void_p),
                ("ln2_w", ctypes.c_void_p), ("ln2_b", ctypes.c_void_p), ("mlp", ConvW)]


class TBlockW(ctypes.Structure):
    _fields_ = [("ln1_w", ctypes.c_void_p), ("ln1_b", ctypes.c_void_p), ("ln3_w", ctypes.c_void_p), ("ln3_b", ctypes.c_void_p),
                ("qkv", ConvW), ("out", ConvW), ("ff1", ConvW), ("ff2", ConvW)]


class FlowWeights(ctypes.Structure):
    _fields_ = [("embedding", ctypes.c_void_p), ("spk", ConvW), ("embed_lin", ConvW), ("up_embed_lin", ConvW),
                ("embed_ln_w", ctypes.c_void_p), ("embed_ln_b", ctypes.c_void_p), ("up_embed_ln_w", ctypes.c_void_p),
                ("up_embed_ln_b", ctypes.c_void_p), ("after_w", ctypes.c_void_p), ("after_b", ctypes.c_void_p),
                ("pre1", ConvW), ("pre2", ConvW), ("up_conv", ConvW), ("enc", ctypes.POINTER(ConformerW)), ("enc_proj", ConvW),
                ("time1", ConvW), ("time2", ConvW), ("resnets", ctypes.POINTER(ResnetW)), ("tblocks", ctypes.POINTER(TBlockW)),
                ("down_conv", ConvW), ("up_conv2", ConvW), ("final_conv", ConvW), ("final_proj", ConvW),
                ("final_ln_w", ctypes.c_void_p), ("final_ln_b", ctypes.c_void_p)]


class FlowConfigC(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("vocab", "dim", "mel", "spk_dim", "enc_layers", "up_layers", "enc_heads", "enc_ffn", "pre_lookahead",
                                              "est_ch", "est_heads", "est_head_dim", "est_blocks", "est_mid", "n_steps", "max_cache", "prefix")] + \
               [("cfg_rate", ctypes.c_float)]


def _bind(L):
    if getattr(L, "_flow_bound", False):
        return
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.vox_flow_create.restype = ci
    L.vox_flow_create.argtypes = [vp, ctypes.POINTER(FlowConfigC), ctypes.POINTER(FlowWeights), ci, ci, ci, vp, vp, ctypes.POINTER(vp)]
    L.vox_flow_destroy.restype, L.vox_flow_destroy.argtypes = None, [vp]
    L.vox_flow_set_prompt.restype = ci
    L.vox_flow_set_prompt.argtypes = [vp, vp, vp, ci, vp, vp, vp, ctypes.c_uint64, ctypes.c_uint32, vp]
    L.vox_flow_decode_chunk.restype = ci
    L.vox_flow_decode_chunk.argtypes = [vp, vp, vp, ci, ci, vp, ctypes.c_uint64, ctypes.c_uint32, vp, vp]
    L.vox_flow_enable_slots.restype, L.vox_flow_enable_slots.argtypes = ci, [vp, ci]
    L.vox_flow_slot_reset.restype, L.vox_flow_slot_reset.argtypes = ci, [vp, vp, ci]
    L.vox_flow_slot_state.restype, L.vox_flow_slot_state.argtypes = ci, [vp, ci, ctypes.POINTER(ctypes.c_int32)]
    L.vox_flow_decode_chunk_slots.restype = ci
    L.vox_flow_decode_chunk_slots.argtypes = [vp, vp, vp, ci, ci, ctypes.POINTER(ctypes.c_int32), vp, ctypes.c_uint64, ctypes.c_uint32, vp, vp]
    L._flow_bound = True


def time_schedule(cfg: FlowConfig):
    """(sinusoidal embeddings [n_steps, 4 mel], dt [n_steps]) with the reference's torch expressions (cosyvoice_flow.py:2673-2675,
    2733-2783 for t / dt, :1762-1772 for the embedding)."""
    ts = torch.linspace(0, 1, cfg.n_timesteps + 1)
    ts = 1 - torch.cos(ts * 0.5 * torch.pi)
    t, dt = ts[0], ts[1] - ts[0]
    tt, dts = [], []
    for step in range(1, len(ts)):
        tt.append(t.clone())
        dts.append(dt.clone())
        t = t + dt
        if step < len(ts) - 1:
            dt = ts[step + 1] - t
    tt = torch.stack(tt)
    half = (4 * cfg.mel) // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half).float() * -e)
    e = 1000 * tt.unsqueeze(1) * e.unsqueeze(0)
    return torch.cat((e.sin(), e.cos()), dim=-1).contiguous(), torch.stack(dts).contiguous()


@device_bound
class CosyVoice2Flow:
    def __init__(self, weights: Dict[str, torch.Tensor], config: Optional[FlowConfig] = None, device="cuda", max_batch=8, max_T=32,
                 max_prompt_T=256, seed: int = 0):
        self.cfg = c = config or FlowConfig()
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

        def conv(wp, bias=None):              # wp [taps, N, Cin] -> bf16
            pl = wp.detach().to(torch.bfloat16).to(dev).contiguous()
            self._keep.append(pl)
            return ConvW(pl.data_ptr(), f32(bias) if bias is not None else None, pl.shape[0], pl.shape[1], pl.shape[2], 0)

        def lin(name, bias=True):
            return conv(W[name + ".weight"][None], W[name + ".bias"] if bias else None)

        def conv1d(name):
            return conv(W[name + ".weight"].permute(2, 0, 1), W[name + ".bias"])

        def fused(names, bias=True):
            w = torch.cat([W[n + ".weight"] for n in names], 0)
            b = torch.cat([W[n + ".bias"] for n in names], 0) if bias else None
            return conv(w[None], b)

        fw = FlowWeights()
        fw.embedding = f32(W["input_embedding.weight"])
        fw.spk = lin("spk_embed_affine_layer")
        fw.embed_lin, fw.up_embed_lin = lin("encoder.embed.out.0"), lin("encoder.up_embed.out.0")
        fw.embed_ln_w, fw.embed_ln_b = f32(W["encoder.embed.out.1.weight"]), f32(W["encoder.embed.out.1.bias"])
        fw.up_embed_ln_w, fw.up_embed_ln_b = f32(W["encoder.up_embed.out.1.weight"]), f32(W["encoder.up_embed.out.1.bias"])
        fw.after_w, fw.after_b = f32(W["encoder.after_norm.weight"]), f32(W["encoder.after_norm.bias"])
        fw.pre1, fw.pre2 = conv1d("encoder.pre_lookahead_layer.conv1"), conv1d("encoder.pre_lookahead_layer.conv2")
        fw.up_conv = conv1d("encoder.up_layer.conv")
        enc = (ConformerW * (c.enc_layers + c.up_layers))()
        k = 0
        for grp, nl in (("encoder.encoders", c.enc_layers), ("encoder.up_encoders", c.up_layers)):
            for i in range(nl):
                p, e = f"{grp}.{i}.", enc[k]
                e.qkv = fused([p + "self_attn.linear_q", p + "self_attn.linear_k", p + "self_attn.linear_v"])
                e.out, e.pos = lin(p + "self_attn.linear_out"), lin(p + "self_attn.linear_pos", bias=False)
                e.bias_u, e.bias_v = f32(W[p + "self_attn.pos_bias_u"]), f32(W[p + "self_attn.pos_bias_v"])
                e.w1, e.w2 = lin(p + "feed_forward.w_1"), lin(p + "feed_forward.w_2")
                e.ln_mha_w, e.ln_mha_b = f32(W[p + "norm_mha.weight"]), f32(W[p + "norm_mha.bias"])
                e.ln_ff_w, e.ln_ff_b = f32(W[p + "norm_ff.weight"]), f32(W[p + "norm_ff.bias"])
                k += 1
        fw.enc = ctypes.cast(enc, ctypes.POINTER(ConformerW))
        fw.enc_proj = lin("encoder_proj")
        es = "decoder.estimator."
        fw.time1, fw.time2 = lin(es + "time_mlp.linear_1"), lin(es + "time_mlp.linear_2")
        groups = [es + "down_blocks.0."] + [f"{es}mid_blocks.{i}." for i in range(c.est_mid_blocks)] + [es + "up_blocks.0."]
        res = (ResnetW * len(groups))()
        tbs = (TBlockW * (len(groups) * c.est_blocks))()
        for gi, gp in enumerate(groups):
            p, r = gp + "0.", res[gi]
            r.conv1, r.conv2, r.res = conv1d(p + "block1.block.0"), conv1d(p + "block2.block.0"), conv1d(p + "res_conv")
            r.ln1_w, r.ln1_b = f32(W[p + "block1.block.2.weight"]), f32(W[p + "block1.block.2.bias"])
            r.ln2_w, r.ln2_b = f32(W[p + "block2.block.2.weight"]), f32(W[p + "block2.block.2.bias"])
            r.mlp = lin(p + "mlp.1")
            for j in range(c.est_blocks):
                p, t = f"{gp}1.{j}.", tbs[gi * c.est_blocks + j]
                t.ln1_w, t.ln1_b = f32(W[p + "norm1.weight"]), f32(W[p + "norm1.bias"])
                t.ln3_w, t.ln3_b = f32(W[p + "norm3.weight"]), f32(W[p + "norm3.bias"])
                t.qkv = fused([p + "attn1.to_q", p + "attn1.to_k", p + "attn1.to_v"], bias=False)
                t.out, t.ff1, t.ff2 = lin(p + "attn1.to_out.0"), lin(p + "ff.net.0.proj"), lin(p + "ff.net.2")
        fw.resnets, fw.tblocks = ctypes.cast(res, ctypes.POINTER(ResnetW)), ctypes.cast(tbs, ctypes.POINTER(TBlockW))
        fw.down_conv, fw.up_conv2 = conv1d(es + "down_blocks.0.2"), conv1d(es + "up_blocks.0.2")
        fw.final_conv, fw.final_proj = conv1d(es + "final_block.block.0"), conv1d(es + "final_proj")
        fw.final_ln_w, fw.final_ln_b = f32(W[es + "final_block.block.2.weight"]), f32(W[es + "final_block.block.2.bias"])
        self._arrays = (enc, res, tbs)
        fc = FlowConfigC(c.vocab_size, c.dim, c.mel, c.spk_embed_dim, c.enc_layers, c.up_layers, c.enc_heads, c.enc_ffn, c.pre_lookahead_len,
                         c.est_channels, c.est_heads, c.est_head_dim, c.est_blocks, c.est_mid_blocks, c.n_timesteps, c.max_cache_len,
                         c.prefix_len, c.inference_cfg_rate)
        emb, dt = time_schedule(c)
        h = ctypes.c_void_p()
        N.check(self.L.vox_flow_create(N.ctx(), ctypes.byref(fc), ctypes.byref(fw), max_batch, max_T, max_prompt_T, emb.data_ptr(),
                                       dt.data_ptr(), ctypes.byref(h)))
        self.h, self._fw = h, fw
        self._chunk = 0

    def set_prompt(self, prompt_token: torch.Tensor, prompt_feat: torch.Tensor, embedding: torch.Tensor, noise: Optional[torch.Tensor] = None,
                   noise_stream: int = 0) -> torch.Tensor:
        """The flow half of CosyVoice2Decoder.init_cache: prompt_token [1, Np], prompt_feat [1, 2 Np, mel], embedding [1, spk] ->
        prompt mels [1, mel, 2 (Np + 3)]; the static caches stay on the device."""
        c = self.cfg
        tok = prompt_token.reshape(-1).to(self.device, torch.int32).contiguous()
        feat = prompt_feat.reshape(-1, c.mel).to(self.device, torch.float32).contiguous()
        if feat.shape[0] != 2 * tok.numel():
            raise ValueError("set_prompt: prompt_feat must hold two mel frames per prompt token")
        emb = embedding.reshape(-1).to(self.device, torch.float32).contiguous()
        T2 = 2 * (tok.numel() + 3)
        nz = noise.reshape(c.mel, T2).to(self.device, torch.float32).contiguous() if noise is not None else None
        out = torch.empty(1, c.mel, T2, dtype=torch.float32, device=self.device)
        N.check(self.L.vox_flow_set_prompt(self.h, N.stream(), tok.data_ptr(), tok.numel(), feat.data_ptr(), emb.data_ptr(),
                                           nz.data_ptr() if nz is not None else None, ctypes.c_uint64(self.seed), noise_stream, out.data_ptr()))
        return out

    def forward_chunk(self, token: torch.Tensor, noise: Optional[torch.Tensor] = None, noise_stream: Optional[int] = None,
                      return_mu: bool = False):
        """token [B, T] -> mels fp32 [B, mel, 2T] against the prompt's static caches (shared-prompt mode: nothing is updated)."""
        c = self.cfg
        tok = token.to(self.device, torch.int32).contiguous()
        B, T = tok.shape
        mel = torch.empty(B, c.mel, 2 * T, dtype=torch.float32, device=self.device)
        mu = torch.empty(B, 2 * T, c.mel, dtype=torch.float32, device=self.device) if return_mu else None
        nz = noise.reshape(c.mel, 2 * T).to(self.device, torch.float32).contiguous() if noise is not None else None
        if noise is None and noise_stream is None:
            self._chunk += 1
            noise_stream = self._chunk
        for b0 in range(0, B, self.max_batch):
            nb = min(self.max_batch, B - b0)
            N.check(self.L.vox_flow_decode_chunk(self.h, N.stream(), tok[b0:b0 + nb].data_ptr(), nb, T, nz.data_ptr() if nz is not None else None,
                                                 ctypes.c_uint64(self.seed), int(noise_stream or 0), mel[b0:b0 + nb].data_ptr(),
                                                 mu[b0:b0 + nb].data_ptr() if mu is not None else None))
        return (mel, mu) if return_mu else mel

    # ---- per-request evolving caches (CosyVoice2Decoder with shared_prompt_cache_mode=False) ----
    def enable_slots(self, n_slots: int):
        N.check(self.L.vox_flow_enable_slots(self.h, int(n_slots)))

    def slot_reset(self, slot: int):
        """The request that takes `slot` starts from a copy of the prompt's caches (set_prompt must have run)."""
        N.check(self.L.vox_flow_slot_reset(self.h, N.stream(), int(slot)))

    def slot_state(self, slot: int):
        """[encoder, up-encoder, estimator cache lengths, then their ring offsets]"""
        out = (ctypes.c_int32 * 6)()
        N.check(self.L.vox_flow_slot_state(self.h, int(slot), out))
        return list(out)

    def slot_lens(self, slot: int):
        return self.slot_state(slot)[:3]

    def forward_chunk_slots(self, token: torch.Tensor, slots, noise: Optional[torch.Tensor] = None, noise_stream: Optional[int] = None):
        """token [B, T], slots [B] -> mels fp32 [B, mel, 2T]; the slots' caches are read and then advanced by this chunk.  Requests whose
        caches are in different states (started in different chunks) are decoded in groups of equal state, each group one native call."""
        c = self.cfg
        tok = token.to(self.device, torch.int32).contiguous()
        B, T = tok.shape
        slots = [int(s) for s in slots]
        mel = torch.empty(B, c.mel, 2 * T, dtype=torch.float32, device=self.device)
        nz = noise.reshape(c.mel, 2 * T).to(self.device, torch.float32).contiguous() if noise is not None else None
        if noise is None and noise_stream is None:
            self._chunk += 1
            noise_stream = self._chunk
        groups = {}
        for i, sl in enumerate(slots):
            groups.setdefault(tuple(self.slot_state(sl)), []).append(i)
        for rows in groups.values():
            for g0 in range(0, len(rows), self.max_batch):
                idx = rows[g0:g0 + self.max_batch]
                sub = tok[idx].contiguous()
                out = torch.empty(len(idx), c.mel, 2 * T, dtype=torch.float32, device=self.device)
                sl = (ctypes.c_int32 * len(idx))(*[slots[i] for i in idx])
                N.check(self.L.vox_flow_decode_chunk_slots(self.h, N.stream(), sub.data_ptr(), len(idx), T, sl, nz.data_ptr() if nz is not None else None,
                                                           ctypes.c_uint64(self.seed), int(noise_stream or 0), out.data_ptr(), None))
                mel[idx] = out
        return mel

    def close(self):
        if self.h:
            self.L.vox_flow_destroy(self.h)
            self.h = None
