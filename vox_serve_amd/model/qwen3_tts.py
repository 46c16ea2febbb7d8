"""Qwen3-TTS model plugin on the native engine.

Mirrors Qwen3TTSModel of /root/reference/vox_serve/model/qwen3_tts.py:947-2045 — same properties, the same
prompt layout (preprocess, :1373-1803: custom-voice / voice-design / x-vector-only / ICL voice cloning), is_stop_id,
postprocess — while forward /
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
                 page_size=128, max_num_pages=2048, max_seq_len=2304, max_prefill_tokens=1024, tts_model_size="1b7",
                 speaker_encoder_weights: Optional[Dict[str, torch.Tensor]] = None, speaker_encoder_config=None,
                 audio_encoder_weights: Optional[Dict[str, torch.Tensor]] = None, audio_encoder_config=None,
                 max_reference_seconds: float = 30.0):
        super().__init__(model_name, device, dtype, False, audio_decoder_device)
        self._detokenize_interval = detokenize_interval if detokenize_interval is not None else 10
        self.config = config or Qwen3Cfg()
        self.tokens = tokens or Qwen3TTSTokens(tts_pad=self.config.tts_pad_id, codec_eos=self.config.eos_id)
        self.tts_model_type = tts_model_type
        self.tts_model_size = tts_model_size
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
        # prompt-side encoders of the "base" (voice clone) checkpoints: speaker_encoder.* of the LM checkpoint (qwen3_tts.py:897-901)
        # and the encoder half of the speech tokenizer (qwen3_codec.py:1681-1741)
        self.speaker_encoder = self.audio_encoder = None
        if speaker_encoder_weights is not None:
            from .qwen3_tts_speaker import Qwen3TTSSpeakerEncoder
            self.speaker_encoder = Qwen3TTSSpeakerEncoder(speaker_encoder_weights, speaker_encoder_config, device=device,
                                                          max_seconds=max_reference_seconds)
            if self.speaker_encoder.cfg.enc_dim != self.config.talker.hidden:
                raise ValueError("speaker encoder enc_dim must equal the talker hidden size")
        if audio_encoder_weights is not None:
            from ..tokenizer.qwen3_codec_encoder import Qwen3TTSTokenizerV2Encoder
            self.audio_encoder = Qwen3TTSTokenizerV2Encoder(audio_encoder_weights, audio_encoder_config, device=self.audio_decoder_device,
                                                            max_seconds=max_reference_seconds)

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
               instruct_ids: Optional[List[int]] = None, is_input_streaming: bool = False, speaker_id: Optional[int] = None,
               ref_text_ids: Optional[List[int]] = None, ref_codes0: Optional[List[int]] = None, return_info: bool = False):
        """prompt_ids = tokenizer ids of '<|im_start|>assistant\\n{prompt}<|im_end|>\\n<|im_start|>assistant\\n' (first 3 =
        role tokens, last 5 = template tail).  Returns (input_tokens [n,C] int64, input_masks [n,C] bool); with
        return_info also {"speaker_row": row of the speaker-embedding position or None, "icl_row": first reference-code row
        or None}.  Model type "base" (voice cloning, :1656-1672) puts (tts_pad, codec_pad) at the speaker position — the
        speaker embedding arrives through input_features; with ref_text_ids + ref_codes0 (codebook 0 of the reference codes)
        the ICL layout of :1698-1744 follows: reference text, text, tts_eos, (tts_pad, codec_bos), one row per reference frame."""
        t, C = self.tokens, self.n_codebooks
        language_id = t.codec_language_id.get(language.lower()) if language.lower() != "auto" else None
        design, clone = self.tts_model_type == "voice_design", self.tts_model_type == "base"
        icl = clone and ref_codes0 is not None
        if icl and is_input_streaming:
            raise ValueError("Input streaming is not supported with ICL mode (voice cloning with reference audio). "
                             "Please use x_vector_only_mode=True or disable input streaming.")
        if clone:
            speaker_id = t.codec_pad
        elif not design and speaker_id is None:
            sp = speaker.lower() if speaker else None
            if sp is None or sp not in t.spk_id:
                sp = next(iter(t.spk_id), None)           # the reference falls back to its first speaker (:1542-1544)
            if sp is not None:
                speaker_id = t.spk_id[sp]
                if language.lower() in ("chinese", "auto") and t.spk_is_dialect.get(sp, False):
                    language_id = t.codec_language_id[t.spk_is_dialect[sp]]
            else:
                speaker_id = t.codec_pad
        prefix = ([t.codec_nothink, t.codec_think_bos, t.codec_think_eos] if language_id is None
                  else [t.codec_think, t.codec_think_bos, language_id, t.codec_think_eos])
        rows = []   # (text_id, codec_id, mask)
        info = {"speaker_row": None, "icl_row": None}
        for i in (instruct_ids or []):
            rows.append((i, 0, False))
        for i in range(3):
            rows.append((prompt_ids[i], 0, False))
        for c in prefix:
            rows.append((t.tts_pad, c, True))
        if not design:
            if clone:
                info["speaker_row"] = len(rows)
            rows.append((t.tts_pad, speaker_id, True))
        rows.append((t.tts_bos, t.codec_pad, True))
        end = len(prompt_ids) if is_input_streaming else len(prompt_ids) - 5
        if icl:
            for i in range(3, len(ref_text_ids) - 2):
                rows.append((ref_text_ids[i], t.codec_pad, True))
            for i in range(3, end):
                rows.append((prompt_ids[i], t.codec_pad, True))
            rows.append((t.tts_eos, t.codec_pad, True))
            rows.append((t.tts_pad, t.codec_bos, True))
            info["icl_row"] = len(rows)
            for c0 in ref_codes0:
                rows.append((t.tts_pad, int(c0), True))
        else:
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
        return (toks, masks, info) if return_info else (toks, masks)

    def prompt_features(self, n_rows: int, info: dict, speaker_embedding: Optional[torch.Tensor],
                        ref_codes: Optional[torch.Tensor]) -> torch.Tensor:
        """input_features [n_rows, hidden] of a voice-clone prompt (qwen3_tts.py:1657-1672, 1733-1744), computed by
        vox_qwen3_prompt_features on the LM device: speaker row = speaker_embedding - codec_embedding[codec_pad]; reference-code
        rows = running bf16 sum of the code predictor's embeddings of codebooks 1..15."""
        from .. import _native as N
        dev = torch.device(self.device)
        feats = torch.zeros(n_rows, self.hidden_size, dtype=torch.bfloat16, device=dev)
        spk = spk_out = codes = icl_out = None
        T = 0
        if info["speaker_row"] is not None:
            if speaker_embedding is None or speaker_embedding.numel() != self.hidden_size:
                raise ValueError(f"voice cloning needs a speaker embedding of {self.hidden_size} values")
            spk = speaker_embedding.to(dev, torch.bfloat16).contiguous()
            spk_out = feats[info["speaker_row"]]
        if info["icl_row"] is not None:
            codes = ref_codes.to(dev, torch.int32).contiguous()
            T = codes.shape[0]
            icl_out = feats[info["icl_row"]: info["icl_row"] + T]
        with torch.cuda.device(dev):
            N.check(N.lib().vox_qwen3_prompt_features(
                self.engine.h, N.stream(), codes.data_ptr() if codes is not None else None, T,
                spk.data_ptr() if spk is not None else None, self.tokens.codec_pad,
                spk_out.data_ptr() if spk_out is not None else None, icl_out.data_ptr() if icl_out is not None else None))
            torch.cuda.current_stream().synchronize()
        return feats

    def preprocess(self, prompt: str = None, audio_path: str = None, language: str = "english", speaker: str = "ryan",
                   instruct: str = None, prompt_token_ids: Optional[List[int]] = None, is_input_streaming: bool = False,
                   ref_text: str = None, x_vector_only_mode: bool = False, ref_text_token_ids: Optional[List[int]] = None,
                   speaker_embedding: Optional[torch.Tensor] = None, ref_codes: Optional[torch.Tensor] = None,
                   **kwargs) -> PreprocessOutput:
        """Qwen3TTSModel.preprocess (qwen3_tts.py:1373-1803).  Voice cloning ("base" model type): the speaker embedding and the
        reference codes come from the prompt-side encoders over `audio_path`, or are passed precomputed
        (`speaker_embedding` [hidden], `ref_codes` [T, 16]) — e.g. a voice registered once and reused.  Without network
        access there is no default reference clip (:1494-1508): cloning without audio / codes raises."""
        clone = self.tts_model_type == "base"
        if speaker_embedding is not None and not torch.is_tensor(speaker_embedding):
            speaker_embedding = torch.tensor(speaker_embedding, dtype=torch.float32)      # JSON model_kwargs
        if ref_codes is not None and not torch.is_tensor(ref_codes):
            ref_codes = torch.tensor(ref_codes, dtype=torch.int64)
        language = "auto" if language is None else language
        if language.lower() != "auto" and language.lower() not in self.tokens.codec_language_id:
            language = "auto"
        if instruct == "" or self.tts_model_size == "0b6":        # the 0.6B checkpoints take no instruct (qwen3_tts.py:1463-1466)
            instruct = None
        if prompt_token_ids is None:
            if self.text_tokenizer is None:
                raise RuntimeError("no text tokenizer loaded (offline): pass model_kwargs['prompt_token_ids']")
            tpl = "<|im_start|>assistant\n{prompt}" if is_input_streaming else \
                "<|im_start|>assistant\n{prompt}<|im_end|>\n<|im_start|>assistant\n"
            prompt_token_ids = list(self.text_tokenizer.encode(tpl.format(prompt=prompt)))
        instruct_ids = kwargs.get("instruct_token_ids")
        if instruct and instruct_ids is None:
            if self.text_tokenizer is None:
                raise RuntimeError("instruct needs the text tokenizer")
            instruct_ids = list(self.text_tokenizer.encode(f"<|im_start|>user\n{instruct}<|im_end|>\n"))
        ref_codes0 = None
        if clone:
            if audio_path is not None:
                audio, sr = self._load_audio_to_np(audio_path)
                if speaker_embedding is None:
                    speaker_embedding = self._extract_speaker_embedding(audio, sr)
                if not x_vector_only_mode and ref_codes is None:
                    ref_codes = self._encode_audio_to_codes(audio, sr)
            if speaker_embedding is None:
                raise ValueError("voice cloning needs audio_path (or a precomputed speaker_embedding); the reference's "
                                 "default clip is a network download")
            if x_vector_only_mode:
                ref_codes = None
            if ref_codes is not None:
                if ref_text_token_ids is None:
                    if ref_text is None or self.text_tokenizer is None:
                        raise ValueError("ICL voice cloning needs ref_text (and the text tokenizer) or ref_text_token_ids")
                    ref_text_token_ids = list(self.text_tokenizer.encode(f"<|im_start|>assistant\n{ref_text}<|im_end|>\n"))
                ref_codes0 = [int(v) for v in ref_codes[:, 0].tolist()]
        toks, masks, info = self.layout(list(prompt_token_ids), language, speaker, instruct_ids, is_input_streaming,
                                        ref_text_ids=ref_text_token_ids if ref_codes0 is not None else None,
                                        ref_codes0=ref_codes0, return_info=True)
        if clone:
            feats = self.prompt_features(toks.shape[0], info, speaker_embedding, ref_codes)
        else:
            feats = torch.zeros(toks.shape[0], self.hidden_size, dtype=self.dtype)
        rep = None
        c = self.default_sampling_config
        if c.repetition_penalty is not None and c.repetition_window is not None and c.repetition_penalty != 1.0:
            rep = torch.zeros(c.repetition_window if c.repetition_window > 0 else 1, self.n_codebooks, self.vocab_size,
                              dtype=torch.bool)
        return PreprocessOutput(input_tokens=toks, input_masks=masks, input_features=feats, repetition_cache=rep,
                                decoder_cache=self.audio_decoder_initial_cache(1))

    # ---- prompt-side encoders (qwen3_tts.py:1271-1371) ----
    def _load_audio_to_np(self, x):
        """(waveform float32 mono, sample rate) from a (waveform, sr) pair, a .npy path, or a 16-bit PCM .wav path
        (qwen3_tts.py:1271-1286 reads files / URLs / base64 through librosa + soundfile, which this image lacks)."""
        import numpy as np
        if isinstance(x, (tuple, list)) and len(x) == 2:
            return np.asarray(x[0], dtype=np.float32), int(x[1])
        if isinstance(x, str) and x.endswith(".npy"):
            return np.load(x).astype(np.float32), 24000
        if isinstance(x, str) and x.endswith(".wav"):
            import wave
            with wave.open(x, "rb") as f:
                if f.getsampwidth() != 2:
                    raise ValueError("only 16-bit PCM wav files are read natively")
                a = np.frombuffer(f.readframes(f.getnframes()), dtype="<i2").astype(np.float32) / 32768.0
                if f.getnchannels() > 1:
                    a = a.reshape(-1, f.getnchannels()).mean(axis=1)
                return a, f.getframerate()
        raise ValueError("audio_path: pass (waveform, sample_rate), a .npy or a 16-bit .wav path")

    @staticmethod
    def _to_24k(audio, sr: int):
        """Reference clip at 24 kHz.  The reference resamples with librosa (soxr) on the host (qwen3_tts.py:1513-1518, 1350-1357); here a
        polyphase FIR (scipy.signal.resample_poly) does — same band-limited signal, not the same filter taps, so a non-24 kHz clip gives
        codes / an x-vector close to, not identical with, the reference's."""
        import numpy as np
        if int(sr) == 24000:
            return np.asarray(audio, dtype=np.float32)
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(24000, int(sr))
        return resample_poly(np.asarray(audio, dtype=np.float64), 24000 // g, int(sr) // g).astype(np.float32)

    def _extract_speaker_embedding(self, audio, sr: int) -> torch.Tensor:
        if getattr(self, "speaker_encoder", None) is None:
            raise NotImplementedError("no speaker encoder weights loaded: pass speaker_embedding=")
        wav = torch.from_numpy(self._to_24k(audio, sr))
        return self.speaker_encoder(wav).to(self.dtype)      # the reference's encoder output dtype

    def _encode_audio_to_codes(self, audio, sr: int) -> torch.Tensor:
        if getattr(self, "audio_encoder", None) is None:
            raise NotImplementedError("no codec encoder weights loaded: pass ref_codes=")
        wav = torch.from_numpy(self._to_24k(audio, sr))
        return self.audio_encoder(wav).to(self.device)

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
