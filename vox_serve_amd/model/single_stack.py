"""Shared plugin body for the single-stack speech LMs (one decoder stack, one codebook per step): GLM-4-Voice
and the CosyVoice2 LLM.  forward + sampling of the reference plugins (glm_voice.py:517-592,
cosyvoice2.py:1008-1091) collapse into one native call per step (`LMEngine.frame / prefill`): embedding gather
(or the per-row input_features override), the decoder stack, final norm + head, repetition penalty, sampler and
the next-step feedback all run inside one hipGraph; `update_requests` is the request-state half of `sampling`.
"""
from typing import List, Optional

import torch

from ..engine import LMCfg, LMEngine
from ..sampling import SamplingConfig
from .base import BaseLM, PreprocessOutput


class SingleStackLM(BaseLM):
    stop_token_ids: List[int] = []

    def __init__(self, model_name, cfg: LMCfg, layers, final_norm, embedding, head_w, head_b, sampling: SamplingConfig,
                 device="cuda:0", dtype=torch.bfloat16, audio_decoder_device=None, max_batch_size=8, page_size=128,
                 max_num_pages=2048, max_seq_len=4096, max_prefill_tokens=1024):
        super().__init__(model_name, device, dtype, False, audio_decoder_device)
        self.config = cfg
        self.default_sampling_config = sampling
        s = sampling
        self._rep_window = None
        if s.repetition_penalty is not None and s.repetition_window is not None and s.repetition_penalty != 1.0:
            self._rep_window = s.repetition_window
        self.engine = LMEngine(cfg, layers, final_norm, embedding, head_w, head_b, max_batch=max_batch_size,
                               page_size=page_size, max_pages=max_num_pages, max_seq_len=max_seq_len,
                               max_prefill_rows=max_prefill_tokens, rep_window=self._rep_window, device=device)

    n_codebooks = property(lambda self: 1)
    num_attention_heads = property(lambda self: self.config.stack.heads)
    num_key_value_heads = property(lambda self: self.config.stack.kv_heads)
    num_hidden_layers = property(lambda self: self.config.stack.layers)
    hidden_size = property(lambda self: self.config.stack.hidden)
    head_dim = property(lambda self: self.config.stack.head_dim)
    vocab_size = property(lambda self: self.config.vocab_out)
    n_channels = property(lambda self: 1)

    def is_stop_id(self, token_ids) -> bool:
        t = token_ids[0] if hasattr(token_ids, "__len__") else token_ids
        return int(t) in self.stop_token_ids

    def _new_repetition_cache(self) -> Optional[torch.Tensor]:
        """glm_voice.py:499-513 / cosyvoice2.py:981-996"""
        if self._rep_window is None:
            return None
        return torch.zeros(self._rep_window if self._rep_window > 0 else 1, 1, self.vocab_size, dtype=torch.bool,
                           device=self.device)

    def is_audio_token(self, tok: int) -> bool:
        return True

    def update_requests(self, requests, out: torch.Tensor):
        """out [B,1] int64 on the host: the per-request tail of `sampling` (glm_voice.py:561-590)."""
        for i, req in enumerate(requests):
            row = out[i:i + 1].clone()
            tok = int(row[0, 0])
            req.input_tokens = row
            if self.needs_input_masks:
                req.input_masks = torch.zeros(1, 1, dtype=torch.bool)            # cosyvoice2.py:1062-1064
            if self.needs_input_features:
                req.input_features = torch.zeros(1, self.hidden_size, dtype=self.dtype)
            req.lm_output_tokens.append(row)
            stop = tok in self.stop_token_ids
            if self.is_audio_token(tok) and not stop:
                req.lm_output_audio_tokens.append(row)
            if stop:
                req.done_lm_generation = True
                req.finish_reason = "stop_id_encountered"
            if req.next_position_id > self.max_tokens:
                req.done_lm_generation = True
                req.finish_reason = "max_tokens_reached"

    def postprocess(self, token_ids: torch.Tensor, **kwargs) -> torch.Tensor:
        raise NotImplementedError(f"{type(self).__name__}: the flow-matching + HiFT detokenizer is not built yet "
                                  "(SURVEY.md §8f-3); the speech-LM step is native")


def _split_rows(w: torch.Tensor, sizes):
    out, o = [], 0
    for n in sizes:
        out.append(w[o:o + n].contiguous())
        o += n
    return out
