"""Orpheus-3B speech LM + SNAC detokenizer on the native engine (drop-in surface of
/root/reference/vox_serve/model/orpheus.py:224-507).

Architecture facts taken from the reference: a Llama-3.2-3B body (3072 hidden, 28 layers, 24 heads / 8 KV of 128, FFN
8192, no biases, llama-3.1 RoPE scaling factor 32 / low 1 / high 4 / 8192, theta 5e5, RMSNorm eps 1e-5: :36-130, :62-66)
with tied input/output embeddings (:189) over 156 940 ids; one token per step; audio ids are
128256 + 10 + 4096 * position-in-frame + code, 7 per SNAC frame (:479-481).  Defaults: top_p 0.8 / temperature 0.6 /
repetition penalty 1.3 over all generated tokens (:261-269); stop id 128258 (:259); detokenizer windows of 28 tokens
(4 frames) advancing by 7, each yielding the 2048 samples [2048:4096] of the 8192 decoded (:295-326, :483-507).
forward + sampling collapse into one native call per step (single_stack.py); postprocess runs the SNAC decoder
(tokenizer/snac.py) on the device.
"""
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from ..engine import LMCfg, StackCfg
from ..sampling import SamplingConfig
from ..tokenizer.snac import SNACConfig, SNACDecoder
from .base import PreprocessOutput
from .single_stack import SingleStackLM


@dataclass
class OrpheusConfig:
    """canopylabs/orpheus-3b-0.1-ft config.json (a LlamaConfig)"""
    hidden_size: int = 3072
    intermediate_size: int = 8192
    num_hidden_layers: int = 28
    num_attention_heads: int = 24
    num_key_value_heads: int = 8
    head_dim: int = 128
    vocab_size: int = 156940
    rms_norm_eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_factor: float = 32.0
    rope_low_freq_factor: float = 1.0
    rope_high_freq_factor: float = 4.0
    rope_original_max_position_embeddings: int = 8192
    audio_token_base: int = 128256 + 10          # id of code 0 of frame position 0 (orpheus.py:479-481)
    stop_token_id: int = 128258

    def lm_cfg(self, max_pos=8192) -> LMCfg:
        st = StackCfg(self.hidden_size, self.num_hidden_layers, self.num_attention_heads, self.num_key_value_heads, self.head_dim,
                      self.intermediate_size, eps=self.rms_norm_eps, rope_theta=self.rope_theta, rope_scale=self.rope_factor,
                      rope_llama31=(self.rope_low_freq_factor, self.rope_high_freq_factor, self.rope_original_max_position_embeddings),
                      qk_norm=False, qkv_bias=False)
        return LMCfg(st, self.vocab_size, self.vocab_size, 1, 0, max_pos)


def pack_orpheus_weights(S: Dict[str, torch.Tensor], c: OrpheusConfig):
    layers = []
    for i in range(c.num_hidden_layers):
        p = f"model.layers.{i}."
        layers.append(dict(wqkv=torch.cat([S[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0).contiguous(),
                           wo=S[p + "self_attn.o_proj.weight"], wgate=S[p + "mlp.gate_proj.weight"], wup=S[p + "mlp.up_proj.weight"],
                           wdown=S[p + "mlp.down_proj.weight"], ln1=S[p + "input_layernorm.weight"],
                           ln2=S[p + "post_attention_layernorm.weight"]))
    emb = S["model.embed_tokens.weight"]
    return layers, S["model.norm.weight"], emb, S.get("lm_head.weight", emb)      # tied when the checkpoint has no lm_head


def orpheus_codes(token_ids: torch.Tensor, codebook_size: int = 4096, base: int = 128256 + 10) -> List[torch.Tensor]:
    """LM ids of whole frames [B, 7 * F] -> the three SNAC code levels [B,F], [B,2F], [B,4F]  (orpheus.py:479-499:
    `(id - 128256 - 10) % 4096`; positions 0 | 1,4 | 2,3,5,6 of every frame)."""
    mf = (token_ids.reshape(token_ids.shape[0], -1, 7).long() - base) % codebook_size
    F = mf.shape[1]
    return [mf[:, :, 0], mf[:, :, [1, 4]].reshape(-1, 2 * F), mf[:, :, [2, 3, 5, 6]].reshape(-1, 4 * F)]


class OrpheusModel(SingleStackLM):
    available_voices = ["tara", "leah", "jess", "leo", "dan", "mia", "zac", "zoe"]

    def __init__(self, model_name: str, weights: Dict[str, torch.Tensor], codec_weights: Dict[str, torch.Tensor],
                 config: Optional[OrpheusConfig] = None, codec_config: Optional[SNACConfig] = None, text_tokenizer=None,
                 device="cuda:0", dtype=torch.bfloat16, audio_decoder_device=None, sampling: Optional[SamplingConfig] = None,
                 max_pos=8192, sampling_overrides=None, noise_seed: int = 0, **engine_kw):
        self.orpheus_config = config or OrpheusConfig()
        layers, norm, emb, head = pack_orpheus_weights(weights, self.orpheus_config)
        sampling = sampling or SamplingConfig(top_k=None, top_p=0.8, min_p=None, temperature=0.6, repetition_penalty=1.3,
                                              repetition_window=-1, cfg_scale=None)
        if sampling_overrides is not None:
            sampling = sampling_overrides(sampling)
        super().__init__(model_name, self.orpheus_config.lm_cfg(max_pos), layers, norm, emb, head, None, sampling, device=device,
                         dtype=dtype, audio_decoder_device=audio_decoder_device, **engine_kw)
        self.text_tokenizer = text_tokenizer
        self.stop_token_id = self.orpheus_config.stop_token_id
        self.stop_token_ids = [self.stop_token_id]
        self.audio_decoder = SNACDecoder(codec_weights, codec_config, device=self.audio_decoder_device,
                                         max_batch=engine_kw.get("max_batch_size", 8), max_T=16, seed=noise_seed)

    detokenize_interval = property(lambda self: 28)
    detokenize_overlap = property(lambda self: 21)
    output_audio_length = property(lambda self: 4 * self.audio_decoder.hop)      # 2048 for snac_24khz

    @property
    def max_tokens(self) -> int:
        mt = self.default_sampling_config.max_tokens
        return mt if mt is not None else 1200

    def _validate_voice(self, voice):
        if voice and voice not in self.available_voices:
            raise ValueError(f"Voice {voice} is not available for model {self.model_name}")

    def format_prompt_ids(self, prompt: str, voice: Optional[str] = "tara") -> List[int]:
        """orpheus.py:353-372 ("larger" model type): <start> + tokenizer("{voice}: {prompt}") + end tokens"""
        if self.text_tokenizer is None:
            raise RuntimeError("no text tokenizer loaded (offline): pass model_kwargs['prompt_token_ids']")
        text = f"{voice}: {prompt}" if voice else prompt
        ids = list(self.text_tokenizer(text).input_ids)
        return [128259] + ids + [128009, 128260, 128261, 128257] if voice else ids

    def preprocess(self, prompt: str = None, audio_path: str = None, voice="tara", model_type="larger",
                   prompt_token_ids: Optional[List[int]] = None, **kwargs) -> PreprocessOutput:
        assert audio_path is None
        self._validate_voice(voice)
        if prompt_token_ids is None:
            prompt_token_ids = self.format_prompt_ids(prompt, voice)
        ids = torch.tensor(list(prompt_token_ids), dtype=torch.long).view(-1, 1)
        return PreprocessOutput(input_tokens=ids, repetition_cache=self._new_repetition_cache())

    def update_requests(self, requests, out: torch.Tensor):
        """orpheus.py:447-470: every sampled id is an audio token except the stop id (which is dropped)."""
        for i, req in enumerate(requests):
            row = out[i:i + 1].clone()
            tok = int(row[0, 0])
            req.input_tokens = row
            req.lm_output_tokens.append(row)
            if tok == self.stop_token_id:
                req.done_lm_generation = True
                req.finish_reason = "stop_id_encountered"
            else:
                req.lm_output_audio_tokens.append(row)
            if req.next_position_id > self.max_tokens:
                req.done_lm_generation = True
                req.finish_reason = "max_tokens_reached"

    def postprocess(self, token_ids: torch.Tensor, **kwargs) -> torch.Tensor:
        """token_ids [B, 28] or [B, 28, 1] (4 frames of 7 ids) -> audio [B, 1, 2048] = samples [2048:4096] of the window."""
        B = token_ids.shape[0]
        codes = orpheus_codes(token_ids.reshape(B, -1).to(self.audio_decoder.device), self.audio_decoder.cfg.codebook_size,
                              self.orpheus_config.audio_token_base)
        q = 4 * self.audio_decoder.hop          # one frame = 4 latent steps; the window's second frame is kept (2048 of 8192 samples)
        return self.audio_decoder.decode(codes, out_off=q, out_len=q)
