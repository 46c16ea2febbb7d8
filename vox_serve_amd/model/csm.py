"""CSM-1B speech LM on the native engine (drop-in surface of /root/reference/vox_serve/model/csm.py:315-790).

forward / sampling / depth_forward / depth_sampling (:637-770) collapse into one native call per frame
(`CSMEngine.frame / prefill`: 33-column masked embedding sum, 16-layer llama-3.1-RoPE backbone, codebook-0 sampling,
31 depth steps with per-codebook heads, feedback of the next inputs); `update_requests` is the request-state half of
`sampling` incl. its quirks: the stop test reads the row BEFORE the depth loop fills it (every column = codebook 0, so
"token_ids[-2] == 0" is "codebook 0 == 0", :604-606,697-721), and max_tokens is only examined on a stop token (:717-722).
`postprocess` (:772-787) runs the native Mimi decoder (tokenizer/mimi.py), stateless per chunk like the reference."""
from typing import Dict, List, Optional, Tuple

import torch

from ..engine import CSMCfg, CSMEngine
from ..sampling import SamplingConfig
from .base import BaseLMWithDepth, PreprocessOutput


class CSMModel(BaseLMWithDepth):
    def __init__(self, model_name: str, weights: Dict[str, torch.Tensor], config: Optional[CSMCfg] = None, text_tokenizer=None,
                 device="cuda:0", dtype=torch.bfloat16, audio_decoder_device=None, sampling: Optional[SamplingConfig] = None,
                 codec_weights: Optional[Dict[str, torch.Tensor]] = None, codec_config=None, max_batch_size=8, page_size=128, max_num_pages=2048, max_seq_len=2304, max_prefill_tokens=1024,
                 stateful_codec: bool = False, **kw):
        super().__init__(model_name, device, dtype, False, audio_decoder_device)
        self.config = config or CSMCfg()
        self.text_tokenizer = text_tokenizer
        self.stop_token_id = 0
        self.default_sampling_config = sampling or SamplingConfig(top_k=50, top_p=None, min_p=None, temperature=0.9,
                                                                  repetition_penalty=None, repetition_window=None, cfg_scale=None)
        self.engine = CSMEngine(self.config, weights, max_batch=max_batch_size, page_size=page_size, max_pages=max_num_pages,
                                max_seq_len=max_seq_len, max_prefill_rows=max_prefill_tokens, device=device)
        self.engine.keep_hidden = False
        self.default_context = {"tokens": [], "tokens_mask": []}     # csm.py:511-569 builds it from prompt audio (Mimi encoder)
        self.audio_decoder = None
        if codec_weights is not None:
            from ..tokenizer.mimi import MimiDecoder
            self.audio_decoder = MimiDecoder(codec_weights, codec_config, num_codebooks=self.config.n_codebooks,
                                             device=self.audio_decoder_device, max_batch=max_batch_size, max_frames=10)
            # option, NOT the reference's behaviour (it decodes every chunk from a fresh state: seams at chunk borders): each
            # request keeps its Mimi streaming state in a slot, chunks continue seamlessly
            if stateful_codec:
                self.audio_decoder.enable_streaming(max(64, 2 * max_batch_size))
        self.stateful_codec = bool(stateful_codec and self.audio_decoder is not None)

    n_codebooks = property(lambda self: self.config.n_codebooks + 1)
    depth_n_codebooks = property(lambda self: self.config.n_codebooks)
    num_attention_heads = property(lambda self: self.config.backbone.heads)
    num_key_value_heads = property(lambda self: self.config.backbone.kv_heads)
    num_hidden_layers = property(lambda self: self.config.backbone.layers)
    hidden_size = property(lambda self: self.config.backbone.hidden)
    head_dim = property(lambda self: self.config.backbone.head_dim)
    depth_num_attention_heads = property(lambda self: self.config.depth.heads)
    depth_num_key_value_heads = property(lambda self: self.config.depth.kv_heads)
    depth_num_hidden_layers = property(lambda self: self.config.depth.layers)
    depth_hidden_size = property(lambda self: self.config.depth.hidden)
    depth_head_dim = property(lambda self: self.config.depth.head_dim)
    depth_vocab_size = property(lambda self: self.config.vocab)
    vocab_size = property(lambda self: self.config.vocab)
    needs_watermarking = property(lambda self: True)
    watermarker_type = property(lambda self: "silentcipher")
    needs_input_masks = property(lambda self: True)
    detokenize_interval = property(lambda self: 10)
    detokenize_overlap = property(lambda self: 0)
    n_channels = property(lambda self: 1)
    output_audio_length = property(lambda self: 19200)

    @property
    def max_tokens(self) -> int:
        mt = self.default_sampling_config.max_tokens
        return mt if mt is not None else 1200

    def is_stop_id(self, token_ids: List[int]) -> bool:
        return token_ids[-2] == self.stop_token_id           # the last audio codebook, before the text column

    def _tokenize_text_segment(self, text: str, speaker: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """csm.py:473-486: `[speaker]text` -> rows with the text id in the last column"""
        if self.text_tokenizer is None:
            raise RuntimeError("no text tokenizer loaded (offline): pass model_kwargs['prompt_token_ids']")
        return self.frames_from_text_ids(self.text_tokenizer.encode(f"[{speaker}]{text}"))

    def frames_from_text_ids(self, text_ids) -> Tuple[torch.Tensor, torch.Tensor]:
        C1 = self.n_codebooks
        toks = torch.zeros(len(text_ids), C1, dtype=torch.long)
        mask = torch.zeros(len(text_ids), C1, dtype=torch.bool)
        toks[:, -1] = torch.as_tensor(list(text_ids), dtype=torch.long)
        mask[:, -1] = True
        return toks, mask

    def frames_from_audio_codes(self, codes: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """csm.py:488-509: context audio [n_codebooks, T] -> T+1 rows (an all-zero EOS frame appended), audio columns"""
        C1 = self.n_codebooks
        codes = torch.cat([codes.long(), torch.zeros(codes.shape[0], 1, dtype=torch.long)], dim=1)
        toks = torch.zeros(codes.shape[1], C1, dtype=torch.long)
        mask = torch.zeros(codes.shape[1], C1, dtype=torch.bool)
        toks[:, :-1] = codes.transpose(0, 1)
        mask[:, :-1] = True
        return toks, mask

    def preprocess(self, prompt: str = None, audio_path: str = None, speaker=0, context=None, prompt_token_ids=None,
                   **kwargs) -> PreprocessOutput:
        assert audio_path is None
        toks, mask = (self.frames_from_text_ids(prompt_token_ids) if prompt_token_ids is not None
                      else self._tokenize_text_segment(prompt, speaker))
        if context is None:
            toks = torch.cat(self.default_context["tokens"] + [toks], dim=0)
            mask = torch.cat(self.default_context["tokens_mask"] + [mask], dim=0)
        return PreprocessOutput(input_tokens=toks, input_masks=mask, repetition_cache=None,
                                decoder_cache=self.audio_decoder_initial_cache(1) if self.stateful_codec else None)

    def update_requests(self, requests, out: torch.Tensor):
        """out [B, 33] int64 on the host (all 32 codes already sampled): csm.py:699-725 + 760-768."""
        C1 = self.n_codebooks
        for i, req in enumerate(requests):
            row = out[i:i + 1].clone()
            c0 = int(row[0, 0])
            req.input_tokens = torch.zeros(1, C1, dtype=torch.long)
            req.input_tokens[0, :C1 - 1] = row[0, :C1 - 1]
            req.input_masks = torch.ones(1, C1, dtype=torch.bool)
            req.input_masks[:, -1] = False
            req.lm_output_tokens.append(row)
            if c0 != self.stop_token_id:               # evaluated on the pre-depth row: every column == codebook 0
                req.lm_output_audio_tokens.append(row)
            elif req.next_position_id > self.max_tokens:
                req.done_lm_generation, req.finish_reason = True, "max_tokens_reached"
            else:
                req.done_lm_generation, req.finish_reason = True, "stop_id_encountered"

    def postprocess(self, token_ids: torch.Tensor, **kwargs) -> torch.Tensor:
        """token_ids [B, interval, 33] (last column = text, dropped; codes clamped to [0, 2047] in the RVQ kernel) ->
        audio [B, 1, interval*1920]; Mimi is stateless per chunk like the reference (csm.py:772-787)."""
        if self.audio_decoder is None:
            raise RuntimeError("CSMModel: no Mimi weights loaded (pass codec_weights=...)")
        cache = kwargs.get("decoder_cache")
        if self.stateful_codec and cache is not None:
            return self.audio_decoder.decode_chunk(token_ids, cache.slot.tolist(), code_layout="BTQ")
        return self.audio_decoder.decode(token_ids, code_layout="BTQ")

    def audio_decoder_initial_cache(self, batch_size: int):
        return self.audio_decoder.init_cache(batch_size) if self.stateful_codec else None
