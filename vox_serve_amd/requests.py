"""Request / LMInputs boundary types — same fields as /root/reference/vox_serve/requests.py:12-91 so schedulers and
model plugins written against the reference keep working."""
from dataclasses import dataclass, field
from queue import Queue
from typing import Any, Dict, List, Optional, TypedDict

import torch

from .sampling import SamplingConfig
from .tokenizer.base import DecoderCache


@dataclass
class Request:
    request_id: str
    prompt: str = None
    audio_path: str = None
    sampling_config: SamplingConfig = None
    model_kwargs: Dict[str, Any] = field(default_factory=dict)

    # next_position_id == len(input_tokens) + len(lm_output_tokens) + 1   (reference quirk Q1 kept)
    next_position_id: int = None

    kv_pages: List[int] = None
    kv_token_len: int = None
    kv_last_page_len: int = None

    input_tokens: torch.Tensor = None
    input_length: int = None
    lm_output_tokens: List[torch.Tensor] = field(default_factory=list)
    lm_output_audio_tokens: List[torch.Tensor] = field(default_factory=list)
    output_audio: Queue = field(default_factory=Queue)

    input_features: torch.Tensor = None
    input_masks: torch.Tensor = None
    repetition_cache: torch.Tensor = None
    decoder_cache: DecoderCache = None

    done_lm_prefill: bool = False
    audio_decode_idx: List[int] = field(default_factory=list)
    next_audio_decode_idx: List[int] = field(default_factory=list)
    done_lm_generation: bool = False
    done_all: bool = False
    finish_reason: str = None

    is_pressing: bool = False
    is_streaming: bool = False

    is_input_streaming: bool = False
    input_text_buffer: str = ""
    pending_text_tokens: Queue = field(default_factory=Queue)
    text_token_cursor: int = 0
    total_text_tokens: int = 0
    text_complete: bool = False
    waiting_for_text: bool = False
    prefill_ready: bool = False
    eos_injected: bool = False

    chunk_send_timestamps: List[float] = field(default_factory=list)
    chunk_durations: List[float] = field(default_factory=list)


class LMInputs(TypedDict):
    qo_indptr: List[int]
    paged_kv_indptr: List[int]
    paged_kv_indices: List[int]
    paged_kv_last_page_len: List[int]
    input_ids: torch.Tensor
    position_ids: torch.Tensor
    input_features: Optional[torch.Tensor]
    input_masks: Optional[torch.Tensor]
    repetition_cache: Optional[torch.Tensor]
    is_prefill: bool
