"""GLM-4-Voice-9B speech LM on the native engine (drop-in surface of /root/reference/vox_serve/model/glm_voice.py).

Architecture facts taken from the reference: fused query_key_value with bias split [q | k | v] (:137-144), fused
swiglu dense_h_to_4h split in halves (:95-97), RoPE on the first half of each head, interleaved pairs, theta 1e4
(:150-158), RMSNorm eps 3.90625e-08, 40 layers x 4096, 32 heads / 2 KV groups, vocab 168960, untied output_layer.
Defaults: top_p 0.8 / temperature 0.8 (:358-366) -> the full-vocabulary sampler; stop ids (:355);
audio tokens are ids >= <|audio_0|> (:356, :569); 25 tokens per detokenizer call (:402-404).
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from ..engine import LMCfg, StackCfg
from ..sampling import SamplingConfig
from .base import PreprocessOutput
from .single_stack import SingleStackLM


@dataclass
class GLMVoiceConfig:
    add_bias_linear: bool = False
    add_qkv_bias: bool = True
    eos_token_id: List[int] = field(default_factory=lambda: [151329, 151336, 151338])
    ffn_hidden_size: int = 13696
    hidden_size: int = 4096
    layernorm_epsilon: float = 3.90625e-08
    multi_query_group_num: int = 2
    num_attention_heads: int = 32
    num_layers: int = 40
    pad_token_id: int = 151329
    padded_vocab_size: int = 168960
    rope_ratio: int = 1
    vocab_size: int = 168960
    audio_offset: int = 152353          # id of <|audio_0|> in the glm-4-voice tokenizer

    def lm_cfg(self, max_pos=8192) -> LMCfg:
        d = self.hidden_size // self.num_attention_heads
        st = StackCfg(self.hidden_size, self.num_layers, self.num_attention_heads, self.multi_query_group_num, d,
                      self.ffn_hidden_size, eps=self.layernorm_epsilon, rope_theta=1e4, rope_scale=float(self.rope_ratio),
                      rope_dim=d // 2, rope_interleave=True, qk_norm=False, qkv_bias=True)
        return LMCfg(st, self.padded_vocab_size, self.padded_vocab_size, 1, 0, max_pos)


def pack_glm_weights(S: Dict[str, torch.Tensor], c: GLMVoiceConfig):
    """Reference state_dict -> engine layer dicts.  Layout only: the fused QKV is already [q | k | v] rows; the fused
    swiglu matrix is viewed as its two halves (no copy)."""
    layers = []
    for i in range(c.num_layers):
        p = f"transformer.encoder.layers.{i}."
        h4 = S[p + "mlp.dense_h_to_4h.weight"]
        layers.append(dict(wqkv=S[p + "self_attention.query_key_value.weight"],
                           bqkv=S[p + "self_attention.query_key_value.bias"], wo=S[p + "self_attention.dense.weight"],
                           wgate=h4[:c.ffn_hidden_size], wup=h4[c.ffn_hidden_size:],
                           wdown=S[p + "mlp.dense_4h_to_h.weight"], ln1=S[p + "input_layernorm.weight"],
                           ln2=S[p + "post_attention_layernorm.weight"]))
    return (layers, S["transformer.encoder.final_layernorm.weight"], S["transformer.embedding.word_embeddings.weight"],
            S["transformer.output_layer.weight"])


class GLMVoiceModel(SingleStackLM):
    def __init__(self, model_name: str, weights: Dict[str, torch.Tensor], config: Optional[GLMVoiceConfig] = None,
                 text_tokenizer=None, device="cuda:0", dtype=torch.bfloat16, audio_decoder_device=None,
                 sampling: Optional[SamplingConfig] = None, max_pos=8192, sampling_overrides=None, codec_weights: Optional[dict] = None,
                 codec_config: Optional[dict] = None, codec_seed: int = 0, **engine_kw):
        self.glm_config = config or GLMVoiceConfig()
        layers, norm, emb, head = pack_glm_weights(weights, self.glm_config)
        sampling = sampling or SamplingConfig(top_k=None, top_p=0.8, min_p=None, temperature=0.8, repetition_penalty=None,
                                              repetition_window=None, cfg_scale=None)
        if sampling_overrides is not None:      # load_model's per-field overrides, applied before the engine sizes its caches
            sampling = sampling_overrides(sampling)
        super().__init__(model_name, self.glm_config.lm_cfg(max_pos), layers, norm, emb, head, None, sampling,
                         device=device, dtype=dtype, audio_decoder_device=audio_decoder_device, **engine_kw)
        self.text_tokenizer = text_tokenizer
        self.stop_token_ids = list(self.glm_config.eos_token_id)
        self.audio_offset = (text_tokenizer.convert_tokens_to_ids("<|audio_0|>") if text_tokenizer is not None
                             else self.glm_config.audio_offset)
        # detokenizer (glm_voice.py:355-369: GLMAudioDecoder = flow + HiFT, stateless per 25-token window)
        self.audio_decoder = None
        if codec_weights is not None:
            from ..tokenizer.glm import GLMAudioDecoder
            cc = codec_config or {}
            self.audio_decoder = GLMAudioDecoder(codec_weights["flow"], codec_weights["hift"], device=self.audio_decoder_device or device,
                                                 flow_config=cc.get("flow"), hift_config=cc.get("hift"),
                                                 max_batch=engine_kw.get("max_batch_size", 8), max_tokens=self.detokenize_interval, seed=codec_seed)

    def postprocess(self, token_ids: torch.Tensor, **kwargs) -> torch.Tensor:
        """token_ids [B, 25, 1] (LM ids of audio tokens) -> audio [B, 1, 44032]   (glm_voice.py:594-596)"""
        if self.audio_decoder is None:
            raise NotImplementedError("GLMVoiceModel: no detokenizer weights were given (codec_weights={'flow': ..., 'hift': ...})")
        audio = self.audio_decoder(token_ids[:, :, 0] - self.audio_offset, None)
        return audio[:, None, :]

    supports_audio_input = property(lambda self: True)
    detokenize_interval = property(lambda self: 25)
    detokenize_overlap = property(lambda self: 0)
    output_audio_length = property(lambda self: 44032)

    @property
    def max_tokens(self) -> int:
        mt = self.default_sampling_config.max_tokens
        return mt if mt is not None else 512

    def is_audio_token(self, tok: int) -> bool:
        return tok >= self.audio_offset

    @staticmethod
    def format_prompt(prompt: str, audio_tokens: Optional[List[int]] = None) -> str:
        """glm_voice.py:463-483"""
        if audio_tokens is not None:
            user = "<|begin_of_audio|>" + "".join(f"<|audio_{x}|>" for x in audio_tokens) + "<|end_of_audio|>"
            system = ("User will provide you with a speech instruction. Do it step by step. First, think about the "
                      "instruction and respond in a interleaved manner, with 13 text token followed by 26 audio tokens. ")
        else:
            user = prompt
            system = ("User will provide you with a text instruction. Do it step by step. First, think about the "
                      "instruction and respond in a interleaved manner, with 13 text token followed by 26 audio tokens.")
        return f"<|system|>\n{system}<|user|>\n{user}<|assistant|>streaming_transcription\n"

    def preprocess(self, prompt: str = None, audio_path: str = None, prompt_token_ids: Optional[List[int]] = None,
                   audio_tokens: Optional[List[int]] = None, **kwargs) -> PreprocessOutput:
        if audio_path is not None and audio_tokens is None:
            raise NotImplementedError("audio prompts need the GLM-4-Voice whisper-VQ encoder (prompt side, out of the hot "
                                      "path): pass model_kwargs['audio_tokens']")
        if prompt_token_ids is None:
            if self.text_tokenizer is None:
                raise RuntimeError("no text tokenizer loaded (offline): pass model_kwargs['prompt_token_ids']")
            prompt_token_ids = self.text_tokenizer(self.format_prompt(prompt, audio_tokens)).input_ids
        ids = torch.tensor(list(prompt_token_ids), dtype=torch.long).view(-1, 1)
        return PreprocessOutput(input_tokens=ids, repetition_cache=self._new_repetition_cache())
