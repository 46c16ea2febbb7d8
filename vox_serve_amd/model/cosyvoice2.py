"""CosyVoice2-0.5B speech LM on the native engine (drop-in surface of /root/reference/vox_serve/model/cosyvoice2.py).

Architecture facts taken from the reference: Qwen2-0.5B body (896 hidden, 24 layers, 14 heads / 2 KV, head 64, FFN
4864, q/k/v bias, NeoX RoPE theta 1e6: :27-38, :122-168); the prompt rows are precomputed embeddings
[sos | text | task_id | prompt speech] handed over as input_features with mask 1 (:933-986); decode rows are
speech_embedding[clamp(id)] (:1019-1024); logits = llm_decoder (with bias) over 6561+3 speech ids (:313);
top_k 25 (:396-404); stop ids 6561..6563 (:389); 28-token detokenizer windows overlapping by 3 (:595-602).
"""
from dataclasses import dataclass
from typing import Dict, Optional

import torch

from ..engine import LMCfg, StackCfg
from ..sampling import SamplingConfig
from .base import PreprocessOutput
from .single_stack import SingleStackLM


@dataclass
class CosyVoice2Config:
    llm_input_size: int = 896
    llm_output_size: int = 896
    speech_token_size: int = 6561
    hidden_size: int = 896
    intermediate_size: int = 4864
    num_attention_heads: int = 14
    num_key_value_heads: int = 2
    num_hidden_layers: int = 24
    vocab_size: int = 151936
    rope_theta: float = 1000000.0
    rms_norm_eps: float = 1e-06

    def lm_cfg(self, max_pos=8192) -> LMCfg:
        st = StackCfg(self.hidden_size, self.num_hidden_layers, self.num_attention_heads, self.num_key_value_heads,
                      self.hidden_size // self.num_attention_heads, self.intermediate_size, eps=self.rms_norm_eps,
                      rope_theta=self.rope_theta, qk_norm=False, qkv_bias=True)
        return LMCfg(st, self.speech_token_size + 3, self.speech_token_size + 3, 1, 1, max_pos)


def pack_cosyvoice2_weights(S: Dict[str, torch.Tensor], c: CosyVoice2Config):
    layers = []
    for i in range(c.num_hidden_layers):
        p = f"llm.model.model.layers.{i}."
        cat = lambda kind: torch.cat([S[p + f"self_attn.{n}_proj.{kind}"] for n in "qkv"], 0).contiguous()
        layers.append(dict(wqkv=cat("weight"), bqkv=cat("bias"), wo=S[p + "self_attn.o_proj.weight"],
                           wgate=S[p + "mlp.gate_proj.weight"], wup=S[p + "mlp.up_proj.weight"],
                           wdown=S[p + "mlp.down_proj.weight"], ln1=S[p + "input_layernorm.weight"],
                           ln2=S[p + "post_attention_layernorm.weight"]))
    return (layers, S["llm.model.model.norm.weight"], S["speech_embedding.weight"], S["llm_decoder.weight"],
            S["llm_decoder.bias"])


class CosyVoice2Model(SingleStackLM):
    sos, task_id = 0, 1            # rows of llm_embedding (cosyvoice2.py:351-352)

    def __init__(self, model_name: str, weights: Dict[str, torch.Tensor], config: Optional[CosyVoice2Config] = None,
                 text_tokenizer=None, speaker_ref: Optional[dict] = None, device="cuda:0", dtype=torch.bfloat16,
                 audio_decoder_device=None, sampling: Optional[SamplingConfig] = None, max_pos=8192, sampling_overrides=None,
                 codec_weights: Optional[dict] = None, codec_config: Optional[dict] = None, use_detokenizer_cache: bool = False,
                 codec_seed: int = 0, **engine_kw):
        self.cv_config = config or CosyVoice2Config()
        layers, norm, emb, head, head_b = pack_cosyvoice2_weights(weights, self.cv_config)
        sampling = sampling or SamplingConfig(top_k=25, top_p=None, min_p=None, temperature=1.0, repetition_penalty=None,
                                              repetition_window=None, cfg_scale=None)
        if sampling_overrides is not None:      # load_model's per-field overrides, applied before the engine sizes its caches
            sampling = sampling_overrides(sampling)
        super().__init__(model_name, self.cv_config.lm_cfg(max_pos), layers, norm, emb, head, head_b, sampling,
                         device=device, dtype=dtype, audio_decoder_device=audio_decoder_device, **engine_kw)
        dev = torch.device(device)
        self.text_embedding = weights["llm.model.model.embed_tokens.weight"].to(dev)
        self.llm_embedding = weights["llm_embedding.weight"].to(dev)
        self.speech_embedding = weights["speech_embedding.weight"].to(dev)
        self.text_tokenizer = text_tokenizer
        # default speaker reference (cosyvoice2.py:885-922 builds it from a prompt wav through the S3 tokenizer and the
        # campplus ONNX model: prompt side).  {"ref_text_ids": LongTensor[n], "prompt_speech_token": LongTensor[m]}
        self.speaker_ref = speaker_ref or {"ref_text_ids": torch.zeros(0, dtype=torch.long),
                                           "prompt_speech_token": torch.zeros(0, dtype=torch.long)}
        self.stop_token_ids = [self.cv_config.speech_token_size + i for i in range(3)]
        # detokenizer (cosyvoice2.py:377-418): flow + HiFT; the plugin's default is the shared prompt cache — the speaker prompt is run
        # through the flow once here and every chunk of every request is decoded against its caches
        # use_detokenizer_cache=True: every request owns evolving caches (cosyvoice2.py:325-335, 514-560, 1104-1117)
        self.use_detokenizer_cache = bool(use_detokenizer_cache)
        self.audio_decoder = None
        if codec_weights is not None:
            from ..tokenizer.cosyvoice2 import CosyVoice2Decoder
            cc = codec_config or {}
            need = [k for k in ("prompt_speech_token", "prompt_feat", "embedding") if k not in self.speaker_ref]
            if need:
                raise ValueError(f"CosyVoice2Model: speaker_ref lacks {need} (the detokenizer's prompt: speech tokens, mel, x-vector)")
            self.audio_decoder = CosyVoice2Decoder(codec_weights["flow"], codec_weights["hift"], device=self.audio_decoder_device or device,
                                                   flow_config=cc.get("flow"), hift_config=cc.get("hift"),
                                                   max_batch=engine_kw.get("max_batch_size", 8), max_tokens_per_chunk=self.detokenize_interval,
                                                   max_prompt_tokens=max(64, int(self.speaker_ref["prompt_speech_token"].numel()) + 8),
                                                   seed=codec_seed, shared_prompt_cache_mode=not self.use_detokenizer_cache,
                                                   max_slots=max(16, 2 * engine_kw.get("max_batch_size", 8)))
            self._shared_prompt_cache = self.audio_decoder.init_cache(self.speaker_ref)

    def audio_decoder_initial_cache(self, batch_size: int):
        if not self.use_detokenizer_cache:
            return None     # shared prompt cache mode: nothing per request (cosyvoice2.py:521-524)
        return self.audio_decoder.new_request_cache(batch_size)      # a copy of the prompt's caches per request (:526-560)

    def postprocess(self, token_ids: torch.Tensor, decoder_cache=None, **kwargs) -> torch.Tensor:
        """token_ids [B, 28, 1] -> audio [B, 1, 24000]   (cosyvoice2.py:1093-1103)"""
        if self.audio_decoder is None:
            raise NotImplementedError("CosyVoice2Model: no detokenizer weights were given (codec_weights={'flow': ..., 'hift': ...})")
        if self.use_detokenizer_cache:      # cosyvoice2.py:1104-1117: the request's caches are used and updated in place
            audio, _ = self.audio_decoder.decode_chunk(token_ids[:, :, 0], speech_token_lens=self.detokenize_interval,
                                                       decoder_cache=decoder_cache, ref_dict=self.speaker_ref)
            return audio[:, None, :]
        audio, _ = self.audio_decoder.decode_chunk(token_ids[:, :, 0], speech_token_lens=self.detokenize_interval,
                                                   decoder_cache=self._shared_prompt_cache, ref_dict=self.speaker_ref)
        return audio[:, None, :]

    supports_audio_input = property(lambda self: True)
    needs_input_features = property(lambda self: True)
    needs_input_masks = property(lambda self: True)
    detokenize_interval = property(lambda self: 28)
    detokenize_overlap = property(lambda self: 3)
    output_audio_length = property(lambda self: 24000)

    @property
    def max_tokens(self) -> int:
        mt = self.default_sampling_config.max_tokens
        return mt if mt is not None else 4096

    def preprocess(self, prompt: str = None, audio_path: str = None, prompt_token_ids=None, **kwargs) -> PreprocessOutput:
        """cosyvoice2.py:924-1006: rows [sos | ref text + prompt text | task_id | prompt speech tokens], every row carried
        as an embedding in input_features with mask 1."""
        assert audio_path is None, "audio_path is not supported yet for this model"
        if prompt_token_ids is None:
            if self.text_tokenizer is None:
                raise RuntimeError("no text tokenizer loaded (offline): pass model_kwargs['prompt_token_ids']")
            prompt_token_ids = self.text_tokenizer.encode(prompt)
        dev = self.text_embedding.device
        ref_text = self.speaker_ref["ref_text_ids"].to(dev).long().view(-1)
        speech = self.speaker_ref["prompt_speech_token"].to(dev).long().view(-1)
        text = torch.cat([ref_text, torch.tensor(list(prompt_token_ids), dtype=torch.long, device=dev)])
        ids = torch.cat([torch.tensor([self.sos], device=dev), text, torch.tensor([self.task_id], device=dev), speech])
        feats = torch.cat([self.llm_embedding[self.sos][None], self.text_embedding[text],
                           self.llm_embedding[self.task_id][None], self.speech_embedding[speech]], 0)
        masks = torch.ones(ids.shape[0], 1, dtype=torch.bool)
        return PreprocessOutput(input_tokens=ids.view(-1, 1).cpu(), repetition_cache=self._new_repetition_cache(),
                                input_masks=masks, input_features=feats,
                                decoder_cache=self.audio_decoder_initial_cache(1) if self.audio_decoder is not None else None)
