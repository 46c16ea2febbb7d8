"""Qwen3-TTS model plugin on the native engine.

Mirrors Qwen3TTSModel of /root/reference/vox_serve/model/qwen3_tts.py:947-2045 — same properties, the same
prompt layout (preprocess, :1373-1803, custom-voice / voice-design / x-vector paths; ICL voice cloning needs the
speaker + Mimi encoders which are prompt-side and out of the hot path), is_stop_id, postprocess — while forward /
sampling / depth_forward / depth_sampling (:1805-2004) collapse into one native call per frame
(`Qwen3Engine.frame / prefill`), executed by the worker.
"""
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

import torch

from ..engine import Qwen3Cfg, Qwen3Engine
from ..sampling import SamplingConfig
from ..tokenizer.qwen3_codec import Qwen3CodecConfig, Qwen3TTSDecoder, Qwen3TTSDecoderCache
from .base import BaseLMWithDepth, PreprocessOutput


@dataclass
class Qwen3TTSTokens:
    """Special ids of Qwen3TTSConfig / Qwen3TTSTalkerConfig (qwen3_tts.py:204-262)."""
    tts_bos: int = 151672
    tts_eos: int = 151673
    tts_pad: int = 151671
    codec_bos: int = 2149
    codec_eos: int = 2150
    codec_pad: int = 2148
    codec_think: int = 2154
    codec_nothink: int = 2155
    codec_think_bos: int = 2156
    codec_think_eos: int = 2157
    codec_language_id: Dict[str, int] = field(default_factory=lambda: {
        "chinese": 2055, "english": 2050, "german": 2053, "italian": 2070, "portuguese": 2071, "spanish": 2054,
        "japanese": 2058, "korean": 2064, "french": 2061, "russian": 2069})
    spk_id: Dict[str, int] = field(default_factory=dict)
    spk_is_dialect: Dict[str, Any] = field(default_factory=dict)


class Qwen3TTSModel(BaseLMWithDepth):
    def __init__(self, model_name: str, weights: Dict[str, torch.Tensor], codec_weights: Dict[str, torch.Tensor],
                 config: Optional[Qwen3Cfg] = None, codec_config: Optional[Qwen3CodecConfig] = None,
                 tokens: Optional[Qwen3TTSTokens] = None, text_tokenizer=None, dtype=torch.bfloat16, device="cuda:0",
                 audio_decoder_device=None, detokenize_interval=None, tts_model_type="custom_voice", max_batch_size=8,
                 page_size=128, max_num_pages=2048, max_seq_len=2304, max_prefill_tokens=1024):
        super().__init__(model_name, device, dtype, False, audio_decoder_device)
        self._detokenize_interval = detokenize_interval if detokenize_interval is not None else 10
        self.config = config or Qwen3Cfg()
        self.tokens = tokens or Qwen3TTSTokens(tts_pad=self.config.tts_pad_id, codec_eos=self.config.eos_id)
        self.tts_model_type = tts_model_type
        self.text_tokenizer = text_tokenizer
        self.stop_token_id = self.config.eos_id
        self.suppress_tokens = [i for i in range(self.config.vocab - 1024, self.config.vocab) if i != self.config.eos_id]
        self.default_sampling_config = SamplingConfig(top_k=50, top_p=1.0, min_p=None, temperature=0.9,
                                                      repetition_penalty=1.05, repetition_window=-1, cfg_scale=None)
        self.engine = Qwen3Engine(self.config, weights, max_batch=max_batch_size, page_size=page_size,
                                  max_pages=max_num_pages, max_seq_len=max_seq_len, max_prefill_rows=max_prefill_tokens,
                                  device=device)
        self.audio_decoder = Qwen3TTSDecoder(codec_weights, codec_config, device=self.audio_decoder_device,
                                             max_batch=max_batch_size, max_slots=max(64, 2 * max_batch_size),
                                             detokenize_interval=self._detokenize_interval)

    # ---- properties (qwen3_tts.py:1098-1240) ----
    n_codebooks = property(lambda self: self.config.n_groups + 1)
    depth_n_codebooks = property(lambda self: self.config.n_groups)
    num_attention_heads = property(lambda self: self.config.talker.heads)
    num_key_value_heads = property(lambda self: self.config.talker.kv_heads)
    num_hidden_layers = property(lambda self: self.config.talker.layers)
    hidden_size = property(lambda self: self.config.talker.hidden)
    head_dim = property(lambda self: self.config.talker.head_dim)
    depth_num_attention_heads = property(lambda self: self.config.depth.heads)
    depth_num_key_value_heads = property(lambda self: self.config.depth.kv_heads)
    depth_num_hidden_layers = property(lambda self: self.config.depth.layers)
    depth_hidden_size = property(lambda self: self.config.depth.hidden)
    depth_head_dim = property(lambda self: self.config.depth.head_dim)
    depth_vocab_size = property(lambda self: self.config.depth_vocab)
    vocab_size = property(lambda self: self.config.vocab)
    detokenize_interval = property(lambda self: self._detokenize_interval)
    detokenize_overlap = property(lambda self: 0)
    n_channels = property(lambda self: 1)
    output_audio_length = property(lambda self: self._detokenize_interval * 1920)
    supports_audio_input = property(lambda self: self.tts_model_type == "base")
    needs_input_features = property(lambda self: True)
    needs_input_masks = property(lambda self: True)
    supports_input_streaming = property(lambda self: True)

    @property
    def max_tokens(self) -> int:
        if self.default_sampling_config.max_tokens is not None:
            return self.default_sampling_config.max_tokens
        return 2048

    def is_stop_id(self, token_ids) -> bool:
        return int(token_ids) == self.stop_token_id

    def audio_decoder_initial_cache(self, batch_size: int) -> Qwen3TTSDecoderCache:
        return self.audio_decoder.init_cache(batch_size=batch_size)

    # ---- prompt layout (qwen3_tts.py:1561-1778) ----
    def layout(self, prompt_ids: List[int], language: str = "auto", speaker: Optional[str] = None,
               instruct_ids: Optional[List[int]] = None, is_input_streaming: bool = False, speaker_id: Optional[int] = None):
        """prompt_ids = tokenizer ids of '<|im_start|>assistant\\n{prompt}<|im_end|>\\n<|im_start|>assistant\\n' (first 3 =
        role tokens, last 5 = template tail).  Returns (input_tokens [n,C] int64, input_masks [n,C] bool)."""
        t, C = self.tokens, self.n_codebooks
        language_id = t.codec_language_id.get(language.lower()) if language.lower() != "auto" else None
        design = self.tts_model_type == "voice_design"
        if not design and speaker_id is None:
            sp = speaker.lower() if speaker else None
            if sp is not None and sp in t.spk_id:
                speaker_id = t.spk_id[sp]
                if language.lower() in ("chinese", "auto") and t.spk_is_dialect.get(sp, False):
                    language_id = t.codec_language_id[t.spk_is_dialect[sp]]
            elif t.spk_id:
                speaker_id = next(iter(t.spk_id.values()))
            else:
                speaker_id = t.codec_pad
        prefix = ([t.codec_nothink, t.codec_think_bos, t.codec_think_eos] if language_id is None
                  else [t.codec_think, t.codec_think_bos, language_id, t.codec_think_eos])
        rows = []   # (text_id, codec_id, mask)
        for i in (instruct_ids or []):
            rows.append((i, 0, False))
        for i in range(3):
            rows.append((prompt_ids[i], 0, False))
        for c in prefix:
            rows.append((t.tts_pad, c, True))
        if not design:
            rows.append((t.tts_pad, speaker_id, True))
        rows.append((t.tts_bos, t.codec_pad, True))
        end = len(prompt_ids) if is_input_streaming else len(prompt_ids) - 5
        for k, i in enumerate(range(3, end)):
            last = is_input_streaming and k == end - 3 - 1
            rows.append((prompt_ids[i], t.codec_bos if last else t.codec_pad, True))
        if not is_input_streaming:
            rows.append((t.tts_eos, t.codec_pad, True))
            rows.append((t.tts_pad, t.codec_bos, True))
        n = len(rows)
        toks = torch.zeros(n, C, dtype=torch.long)
        masks = torch.zeros(n, C, dtype=torch.bool)
        for j, (ti, ci, m) in enumerate(rows):
            toks[j, -1], toks[j, 0], masks[j, -1] = ti, ci, m
        return toks, masks

    def preprocess(self, prompt: str = None, audio_path: str = None, language: str = "auto", speaker: str = None,
                   instruct: str = None, prompt_token_ids: Optional[List[int]] = None, is_input_streaming: bool = False,
                   **kwargs) -> PreprocessOutput:
        if audio_path is not None:
            raise NotImplementedError("ICL voice cloning needs the prompt-side speaker / Mimi encoders (out of the hot path)")
        if prompt_token_ids is None:
            if self.text_tokenizer is None:
                raise RuntimeError("no text tokenizer loaded (offline): pass model_kwargs['prompt_token_ids']")
            tpl = "<|im_start|>assistant\n{prompt}" if is_input_streaming else \
                "<|im_start|>assistant\n{prompt}<|im_end|>\n<|im_start|>assistant\n"
            prompt_token_ids = list(self.text_tokenizer.encode(tpl.format(prompt=prompt)))
        instruct_ids = None
        if instruct:
            if self.text_tokenizer is None:
                raise RuntimeError("instruct needs the text tokenizer")
            instruct_ids = list(self.text_tokenizer.encode(f"<|im_start|>user\n{instruct}<|im_end|>\n"))
        toks, masks = self.layout(list(prompt_token_ids), language, speaker, instruct_ids, is_input_streaming)
        feats = torch.zeros(toks.shape[0], self.hidden_size, dtype=self.dtype)
        rep = None
        c = self.default_sampling_config
        if c.repetition_penalty is not None and c.repetition_window is not None and c.repetition_penalty != 1.0:
            rep = torch.zeros(c.repetition_window if c.repetition_window > 0 else 1, self.n_codebooks, self.vocab_size,
                              dtype=torch.bool)
        return PreprocessOutput(input_tokens=toks, input_masks=masks, input_features=feats, repetition_cache=rep,
                                decoder_cache=self.audio_decoder_initial_cache(1))

    def postprocess(self, token_ids: torch.Tensor, decoder_cache: Optional[Qwen3TTSDecoderCache] = None, **kwargs):
        """token_ids [B, interval, n_codebooks] (last column = text token, dropped) -> audio [B,1,interval*1920]
        (qwen3_tts.py:2006-2044; the clamp to [0, depth_vocab-1] happens in the RVQ kernel)."""
        if decoder_cache is None:
            cache = self.audio_decoder.init_cache(token_ids.shape[0])
            try:
                return self.audio_decoder.decode_chunk(token_ids, cache, code_layout="BTQ")[0].clone()
            finally:
                self.audio_decoder.release_cache(cache)
        return self.audio_decoder.decode_chunk(token_ids, decoder_cache, code_layout="BTQ")[0]
