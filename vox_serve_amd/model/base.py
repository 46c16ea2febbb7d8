"""Model-plugin contract — the property / method surface of /root/reference/vox_serve/model/base.py:13-447
(BaseLM :29, BaseLMWithDepth :280, PreprocessOutput :13) that workers and schedulers program against."""
from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import List, Optional

import torch

from ..sampling import SamplingConfig
from ..tokenizer.base import DecoderCache


@dataclass
class PreprocessOutput:
    input_tokens: List[List[int]]
    repetition_cache: Optional[torch.Tensor] = None
    input_masks: Optional[torch.Tensor] = None
    input_features: Optional[torch.Tensor] = None
    decoder_cache: Optional[DecoderCache] = None


class BaseLM(ABC):
    def __init__(self, model_name: str, device: str = "cuda", dtype: torch.dtype = torch.bfloat16,
                 enable_torch_compile: bool = False, audio_decoder_device: str = None):
        self.model_name, self.device, self.dtype = model_name, device, dtype
        self.enable_torch_compile = enable_torch_compile
        self.audio_decoder_device = audio_decoder_device or device
        if torch.device(self.audio_decoder_device) != torch.device(device) and torch.device(self.audio_decoder_device).type == "cuda" \
                and torch.cuda.is_available() and (torch.device(self.audio_decoder_device).index or 0) >= torch.cuda.device_count():
            raise ValueError(f"audio_decoder_device {self.audio_decoder_device}: only {torch.cuda.device_count()} GPU(s) visible")

    def decoder_guard(self):
        """Detokenizer on its own GPU (worker/base.py:55-78, 641-644 of the reference): every native detokenizer call — creation,
        cache init / reset, decode — runs with audio_decoder_device current, so it uses that device's libvoxhip context, workspace
        and streams; the token windows cross with one peer copy per chunk (a few hundred bytes per request)."""
        from .. import _native as N
        return N.device_guard(self.audio_decoder_device)

    # ---- architecture ----
    @property
    @abstractmethod
    def n_codebooks(self) -> int: ...

    @property
    @abstractmethod
    def num_attention_heads(self) -> int: ...

    @property
    @abstractmethod
    def num_key_value_heads(self) -> int: ...

    @property
    @abstractmethod
    def num_hidden_layers(self) -> int: ...

    @property
    @abstractmethod
    def hidden_size(self) -> int: ...

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    @abstractmethod
    def vocab_size(self) -> int: ...

    # ---- capabilities (defaults as in the reference) ----
    @property
    def has_depth_transformer(self) -> bool:
        return False

    @property
    def supports_audio_input(self) -> bool:
        return False

    @property
    def needs_watermarking(self) -> bool:
        return False

    @property
    def watermarker_type(self) -> str:
        return None

    @property
    def needs_input_features(self) -> bool:
        return False

    @property
    def needs_input_masks(self) -> bool:
        return False

    @property
    def use_repetition_penalty(self) -> bool:
        return (hasattr(self, "default_sampling_config")
                and self.default_sampling_config.repetition_penalty is not None
                and self.default_sampling_config.repetition_penalty != 1.0)

    @property
    def supports_input_streaming(self) -> bool:
        return False

    # ---- streaming / detokenize ----
    @property
    @abstractmethod
    def detokenize_interval(self) -> int: ...

    @property
    @abstractmethod
    def detokenize_overlap(self) -> int: ...

    @property
    @abstractmethod
    def max_tokens(self) -> int: ...

    @property
    def n_channels(self) -> int:
        return 1

    @property
    @abstractmethod
    def output_audio_length(self) -> int: ...

    def audio_decoder_initial_cache(self, batch_size: int) -> Optional[DecoderCache]:
        return None

    @abstractmethod
    def is_stop_id(self, token_ids) -> bool: ...

    @abstractmethod
    def preprocess(self, prompt: str = None, audio_path: str = None, **kwargs) -> PreprocessOutput: ...

    @abstractmethod
    def postprocess(self, token_ids: torch.Tensor, **kwargs) -> torch.Tensor: ...


class BaseLMWithDepth(BaseLM):
    @property
    def has_depth_transformer(self) -> bool:
        return True

    @property
    @abstractmethod
    def depth_n_codebooks(self) -> int: ...

    @property
    @abstractmethod
    def depth_num_attention_heads(self) -> int: ...

    @property
    @abstractmethod
    def depth_num_key_value_heads(self) -> int: ...

    @property
    @abstractmethod
    def depth_num_hidden_layers(self) -> int: ...

    @property
    @abstractmethod
    def depth_hidden_size(self) -> int: ...

    @property
    def depth_head_dim(self) -> int:
        return self.depth_hidden_size // self.depth_num_attention_heads

    @property
    @abstractmethod
    def depth_vocab_size(self) -> int: ...
