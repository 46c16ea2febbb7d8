"""GLM-4-Voice detokenizer (speech tokens -> waveform) on libvoxhip: drop-in surface of the reference's `GLMAudioDecoder`
(/root/reference/vox_serve/tokenizer/glm.py:2616-2651): `forward(audio_ids [B, T], token_len) -> speech [B, Tm * 256]` =
GLMFlowModel.inference (block conformer encoder, length regulator, 10-step CFM over a non-causal U-Net) + GLMHiFTModel (HiFT with two x8
stages and SineGen v1, 22.05 kHz).  Stateless per call.

Weights are the reference checkpoints' state_dict names (flow: input_embedding, spk_embed_affine_layer, encoder.*, encoder_proj,
length_regulator.model.*, decoder.estimator.*; hift: see tokenizer/hifigan.py, weight_g / weight_v form).  Packing is layout only (bf16
GEMM operands like the CosyVoice2 flow); the 80 mel channels of the regulator are padded to 96 with zero rows / columns.
"""
import ctypes
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from .. import _native as N
from .base import device_bound
from .cosyvoice_flow import ConformerW, FlowConfig, ResnetW, TBlockW, time_schedule
from .hifigan import HiFTConfig, HiFTGenerator, tconv_taps
from .qwen3_codec import ConvW


@dataclass
class GLMFlowConfig:
    """GLMFlowModel / BlockConformerEncoder / ConditionalDecoder defaults (glm.py:2032-2047, 1005-1034, 1694-1707)"""
    vocab_size: int = 16384
    dim: int = 512
    mel: int = 80
    spk_embed_dim: int = 192
    enc_layers: int = 6
    enc_heads: int = 8
    enc_ffn: int = 2048
    block_size: int = 10
    est_channels: int = 256
    est_heads: int = 8
    est_head_dim: int = 64
    est_blocks: int = 4
    est_mid_blocks: int = 12
    n_timesteps: int = 10
    inference_cfg_rate: float = 0.7
    input_frame_rate: float = 12.5
    sampling_rate: int = 22050
    hop: int = 256
    reg_layers: int = 4
    groups: int = 8

    def mel_len(self, n_tokens: int) -> int:
        """(token_len / self.input_frame_rate * 22050 / 256).int()   (glm.py:2084), with the reference's tensor arithmetic"""
        return int((torch.tensor([n_tokens], dtype=torch.int32) / self.input_frame_rate * self.sampling_rate / self.hop).int().item())


def glm_hift_config(**kw) -> HiFTConfig:
    d = dict(sampling_rate=22050, upsample_rates=[8, 8], upsample_kernel_sizes=[16, 16], source_resblock_kernel_sizes=[7, 11], sine_gen_v1=True)
    d.update(kw)
    return HiFTConfig(**d)


class GlmFlowWeights(ctypes.Structure):
    _fields_ = [("embedding", ctypes.c_void_p), ("spk", ConvW), ("embed_lin", ConvW), ("embed_ln_w", ctypes.c_void_p), ("embed_ln_b", ctypes.c_void_p),
                ("after_w", ctypes.c_void_p), ("after_b", ctypes.c_void_p), ("enc", ctypes.POINTER(ConformerW)), ("enc_proj", ConvW),
                ("reg_conv", ConvW * 4), ("reg_gn_w", ctypes.c_void_p * 4), ("reg_gn_b", ctypes.c_void_p * 4), ("reg_out", ConvW),
                ("time1", ConvW), ("time2", ConvW), ("resnets", ctypes.POINTER(ResnetW)), ("tblocks", ctypes.POINTER(TBlockW)),
                ("down_s2", ConvW), ("down_conv1", ConvW), ("up_tconv", ConvW), ("up_conv1", ConvW), ("final_conv", ConvW), ("final_proj", ConvW),
                ("final_gn_w", ctypes.c_void_p), ("final_gn_b", ctypes.c_void_p)]


class GlmFlowConfigC(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("vocab", "dim", "mel", "mel_padded", "spk_dim", "enc_layers", "enc_heads", "enc_ffn", "block_size",
                                              "est_ch", "est_heads", "est_head_dim", "est_blocks", "est_mid", "n_steps", "reg_layers", "groups")] + \
               [("cfg_rate", ctypes.c_float)]


def _bind(L):
    if getattr(L, "_glmflow_bound", False):
        return
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.vox_glmflow_create.restype = ci
    L.vox_glmflow_create.argtypes = [vp, ctypes.POINTER(GlmFlowConfigC), ctypes.POINTER(GlmFlowWeights), ci, ci, ci, vp, vp, ctypes.POINTER(vp)]
    L.vox_glmflow_destroy.restype, L.vox_glmflow_destroy.argtypes = None, [vp]
    L.vox_glmflow_decode.restype = ci
    L.vox_glmflow_decode.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp, ctypes.c_uint64, ctypes.c_uint32, vp]
    L._glmflow_bound = True


@device_bound
class GLMFlow:
    def __init__(self, weights: Dict[str, torch.Tensor], config: Optional[GLMFlowConfig] = None, device="cuda", max_batch=8, max_T=32, seed: int = 0):
        self.cfg = c = config or GLMFlowConfig()
        self.device = torch.device(device)
        self.max_batch, self.max_T, self.seed = max_batch, max_T, seed
        self.L = N.lib()
        _bind(self.L)
        self._keep = []
        W, dev = weights, self.device
        MP = c.mel + (-c.mel) % 32

        def f32(t):
            t = t.detach().to(device=dev, dtype=torch.float32).contiguous()
            self._keep.append(t)
            return t.data_ptr()

        def conv(wp, bias=None, bias_mod=0, pad_n=0, pad_c=0):          # wp [taps, N, Cin] -> bf16 (zero rows / columns appended)
            wp = wp.detach().float()
            if pad_n or pad_c:
                wp = torch.nn.functional.pad(wp, (0, pad_c, 0, pad_n))
            if bias is not None and pad_n:
                bias = torch.nn.functional.pad(bias.detach().float(), (0, pad_n))
            pl = wp.to(torch.bfloat16).to(dev).contiguous()
            self._keep.append(pl)
            return ConvW(pl.data_ptr(), f32(bias) if bias is not None else None, pl.shape[0], pl.shape[1], pl.shape[2], bias_mod)

        lin = lambda name, bias=True, **kw: conv(W[name + ".weight"][None], W[name + ".bias"] if bias else None, **kw)
        conv1d = lambda name, **kw: conv(W[name + ".weight"].permute(2, 0, 1), W[name + ".bias"], **kw)

        def fused(names, bias=True):
            return conv(torch.cat([W[n + ".weight"] for n in names], 0)[None], torch.cat([W[n + ".bias"] for n in names], 0) if bias else None)

        fw = GlmFlowWeights()
        fw.embedding = f32(W["input_embedding.weight"])
        fw.spk, fw.embed_lin = lin("spk_embed_affine_layer"), lin("encoder.embed.out.0")
        fw.embed_ln_w, fw.embed_ln_b = f32(W["encoder.embed.out.1.weight"]), f32(W["encoder.embed.out.1.bias"])
        fw.after_w, fw.after_b = f32(W["encoder.after_norm.weight"]), f32(W["encoder.after_norm.bias"])
        enc = (ConformerW * c.enc_layers)()
        for i in range(c.enc_layers):
            p, e = f"encoder.encoders.{i}.", enc[i]
            e.qkv = fused([p + "self_attn.linear_q", p + "self_attn.linear_k", p + "self_attn.linear_v"])
            e.out, e.pos = lin(p + "self_attn.linear_out"), lin(p + "self_attn.linear_pos", bias=False)
            e.bias_u, e.bias_v = f32(W[p + "self_attn.pos_bias_u"]), f32(W[p + "self_attn.pos_bias_v"])
            e.w1, e.w2 = lin(p + "feed_forward.w_1"), lin(p + "feed_forward.w_2")
            e.ln_mha_w, e.ln_mha_b = f32(W[p + "norm_mha.weight"]), f32(W[p + "norm_mha.bias"])
            e.ln_ff_w, e.ln_ff_b = f32(W[p + "norm_ff.weight"]), f32(W[p + "norm_ff.bias"])
        fw.enc = ctypes.cast(enc, ctypes.POINTER(ConformerW))
        fw.enc_proj = lin("encoder_proj", pad_n=MP - c.mel)
        for i in range(c.reg_layers):
            fw.reg_conv[i] = conv1d(f"length_regulator.model.{3 * i}", pad_n=MP - c.mel, pad_c=MP - c.mel)
            fw.reg_gn_w[i], fw.reg_gn_b[i] = f32(W[f"length_regulator.model.{3 * i + 1}.weight"]), f32(W[f"length_regulator.model.{3 * i + 1}.bias"])
        fw.reg_out = conv1d(f"length_regulator.model.{3 * c.reg_layers}", pad_c=MP - c.mel)
        es = "decoder.estimator."
        fw.time1, fw.time2 = lin(es + "time_mlp.linear_1"), lin(es + "time_mlp.linear_2")
        groups = [es + "down_blocks.0.", es + "down_blocks.1."] + [f"{es}mid_blocks.{i}." for i in range(c.est_mid_blocks)] + \
                 [es + "up_blocks.0.", es + "up_blocks.1."]
        res, tbs = (ResnetW * len(groups))(), (TBlockW * (len(groups) * c.est_blocks))()
        for gi, gp in enumerate(groups):
            p, r = gp + "0.", res[gi]
            r.conv1, r.conv2, r.res = conv1d(p + "block1.block.0"), conv1d(p + "block2.block.0"), conv1d(p + "res_conv")
            r.ln1_w, r.ln1_b = f32(W[p + "block1.block.1.weight"]), f32(W[p + "block1.block.1.bias"])
            r.ln2_w, r.ln2_b = f32(W[p + "block2.block.1.weight"]), f32(W[p + "block2.block.1.bias"])
            r.mlp = lin(p + "mlp.1")
            for j in range(c.est_blocks):
                p, t = f"{gp}1.{j}.", tbs[gi * c.est_blocks + j]
                t.ln1_w, t.ln1_b = f32(W[p + "norm1.weight"]), f32(W[p + "norm1.bias"])
                t.ln3_w, t.ln3_b = f32(W[p + "norm3.weight"]), f32(W[p + "norm3.bias"])
                t.qkv = fused([p + "attn1.to_q", p + "attn1.to_k", p + "attn1.to_v"], bias=False)
                t.out, t.ff1, t.ff2 = lin(p + "attn1.to_out.0"), lin(p + "ff.net.0.proj"), lin(p + "ff.net.2")
        fw.resnets, fw.tblocks = ctypes.cast(res, ctypes.POINTER(ResnetW)), ctypes.cast(tbs, ctypes.POINTER(TBlockW))
        wd = W[es + "down_blocks.0.2.conv.weight"]                                            # [C, C, 3], stride 2: one tap over (j, ci)
        fw.down_s2 = conv(wd.permute(0, 2, 1).reshape(wd.shape[0], -1)[None], W[es + "down_blocks.0.2.conv.bias"])
        fw.down_conv1 = conv1d(es + "down_blocks.1.2")
        wt = W[es + "up_blocks.0.2.conv.weight"]                                              # ConvTranspose1d [Cin, Cout, 4]
        fw.up_tconv = conv(tconv_taps(wt.float().cpu(), 2), W[es + "up_blocks.0.2.conv.bias"], bias_mod=wt.shape[1])
        fw.up_conv1, fw.final_conv, fw.final_proj = conv1d(es + "up_blocks.1.2"), conv1d(es + "final_block.block.0"), conv1d(es + "final_proj")
        fw.final_gn_w, fw.final_gn_b = f32(W[es + "final_block.block.1.weight"]), f32(W[es + "final_block.block.1.bias"])
        self._arrays = (enc, res, tbs)
        fc = GlmFlowConfigC(c.vocab_size, c.dim, c.mel, MP, c.spk_embed_dim, c.enc_layers, c.enc_heads, c.enc_ffn, c.block_size, c.est_channels,
                            c.est_heads, c.est_head_dim, c.est_blocks, c.est_mid_blocks, c.n_timesteps, c.reg_layers, c.groups, c.inference_cfg_rate)
        emb, dt = time_schedule(FlowConfig(mel=c.mel, n_timesteps=c.n_timesteps))
        h = ctypes.c_void_p()
        N.check(self.L.vox_glmflow_create(N.ctx(), ctypes.byref(fc), ctypes.byref(fw), max_batch, max_T, c.mel_len(max_T) + 2, emb.data_ptr(),
                                          dt.data_ptr(), ctypes.byref(h)))
        self.h, self._fw = h, fw
        self._call = 0

    def inference(self, token: torch.Tensor, token_len=None, embedding: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None,
                  first_stream: Optional[int] = None) -> torch.Tensor:
        """token [B, T] -> mel fp32 [B, mel, Tm]   (glm.py:2065-2112; embedding None = zeros, as GLMAudioDecoder passes)"""
        c = self.cfg
        tok = token.to(self.device, torch.int32).contiguous()
        B, T = tok.shape
        Tm = c.mel_len(T)
        mel = torch.empty(B, c.mel, Tm, dtype=torch.float32, device=self.device)
        emb = embedding.to(self.device, torch.float32).contiguous() if embedding is not None else None
        nz = noise.to(self.device, torch.float32).contiguous() if noise is not None else None
        if noise is None and first_stream is None:
            self._call += 1
            first_stream = self._call * 65536
        for b0 in range(0, B, self.max_batch):
            nb = min(self.max_batch, B - b0)
            N.check(self.L.vox_glmflow_decode(self.h, N.stream(), tok[b0:b0 + nb].data_ptr(), nb, T, Tm, emb[b0:b0 + nb].data_ptr() if emb is not None else None,
                                              nz[b0:b0 + nb].data_ptr() if nz is not None else None, ctypes.c_uint64(self.seed),
                                              int(first_stream or 0) + b0, mel[b0:b0 + nb].data_ptr()))
        return mel

    def close(self):
        if self.h:
            self.L.vox_glmflow_destroy(self.h)
            self.h = None


@device_bound
class GLMAudioDecoder:
    def __init__(self, flow_weights: Dict[str, torch.Tensor], hift_weights: Dict[str, torch.Tensor], device="cuda",
                 flow_config: Optional[GLMFlowConfig] = None, hift_config: Optional[HiFTConfig] = None, max_batch: int = 8, max_tokens: int = 25,
                 seed: int = 0):
        self.device = torch.device(device)
        self.flow = GLMFlow(flow_weights, flow_config, device=device, max_batch=max_batch, max_T=max_tokens, seed=seed)
        self.hift = HiFTGenerator(hift_weights, hift_config or glm_hift_config(), device=device, max_batch=max_batch,
                                  max_T=self.flow.cfg.mel_len(max_tokens) + 2, seed=seed)
        self.seed, self.use_graph, self._graphs, self._call = seed, True, {}, 0
        L = self.flow.L
        L.vox_flow_fill_noise.restype = ctypes.c_int
        L.vox_flow_fill_noise.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]

    @torch.inference_mode()
    def forward(self, audio_ids: torch.Tensor, token_len=None, flow_noise=None, hift_noise=None, hift_rand_ini=None, first_stream=None,
                hift_stream_base=None) -> torch.Tensor:
        """audio_ids [B, T] -> speech fp32 [B, Tm * 256]   (glm.py:2640-2651)"""
        if (self.use_graph and flow_noise is None and hift_noise is None and hift_rand_ini is None and first_stream is None
                and hift_stream_base is None and audio_ids.shape[0] <= self.flow.max_batch):
            return self._forward_graph(audio_ids)
        mel = self.flow.inference(audio_ids, token_len, None, noise=flow_noise, first_stream=first_stream)
        speech, _ = self.hift.forward_chunk(mel, noise=hift_noise, rand_ini=hift_rand_ini, stream_base=hift_stream_base)
        return speech

    __call__ = forward

    def _forward_graph(self, audio_ids: torch.Tensor) -> torch.Tensor:
        """The ~8 000 launches of a window as one hipGraph per (requests, tokens): inputs in graph-stable buffers, the per-request CFM start
        noise drawn into one before the replay (a graph would freeze the stream ids), the vocoder's streams read from a device array.  The
        first window of a shape runs eagerly, the second is captured."""
        B, T = audio_ids.shape
        c, L = self.flow.cfg, self.flow.L
        Tm = c.mel_len(T)
        ent = self._graphs.get((B, T))
        if ent is None:
            ent = self._graphs[(B, T)] = {"tok": torch.empty(B, T, dtype=torch.int32, device=self.device),
                                          "z": torch.empty(B, c.mel, Tm, dtype=torch.float32, device=self.device),
                                          "sb": torch.empty(B, dtype=torch.int32, device=self.device),
                                          "mel": torch.empty(B, c.mel, Tm, dtype=torch.float32, device=self.device),
                                          "wav": torch.empty(B, Tm * self.hift.upsample_scale, dtype=torch.float32, device=self.device),
                                          "g": None, "calls": 0}
        self._call += 1
        # on the caller's current stream (N.graph_capture: why the decoder owns none)
        ent["tok"].copy_(audio_ids.to(self.device, torch.int32), non_blocking=True)
        ent["sb"].copy_(((torch.arange(B, dtype=torch.int64) + self._call * 65536) * 2).to(torch.int32), non_blocking=True)
        for b in range(B):
            N.check(L.vox_flow_fill_noise(N.stream(), ctypes.c_uint64(self.seed), self._call * 65536 + b, c.mel, Tm, ent["z"][b].data_ptr()))

        def body():
            st = N.stream()
            N.check(L.vox_glmflow_decode(self.flow.h, st, ent["tok"].data_ptr(), B, T, Tm, None, ent["z"].data_ptr(), ctypes.c_uint64(self.seed), 0,
                                         ent["mel"].data_ptr()))
            N.check(self.hift.L.vox_hift_decode(self.hift.h, st, ent["mel"].data_ptr(), B, Tm, None, ctypes.c_uint64(self.seed ^ 0x5A5A),
                                                ent["sb"].data_ptr(), ent["wav"].data_ptr(), None, None))
        ent["calls"] += 1
        if ent["calls"] == 1:
            body()
        else:
            if ent["g"] is None:
                with N.graph_capture() as cap:
                    body()
                ent["g"] = cap.graph
            N.check(L.vox_graph_launch(ent["g"], N.stream()))
        out = ent["wav"].clone()
        return out

    def close(self):
        self.flow.close()
        self.hift.close()
