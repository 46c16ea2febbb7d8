"""CosyVoice2 detokenizer (speech tokens -> waveform) on libvoxhip: drop-in surface of the reference's `CosyVoice2Decoder`
(/root/reference/vox_serve/tokenizer/cosyvoice2.py:774-1046) in the mode the plugin serves by default — the shared prompt cache
(model/cosyvoice2.py:325, 1093-1103): `init_cache(ref_dict)` runs the flow over the speaker prompt once and keeps its caches,
`decode_chunk(speech_tokens, ...)` decodes every 28-token window of every request against them:
    flow (conformer encoder + 10-step CFM)  ->  HiFT vocoder  ->  fade-in over mel_cache_len frames  ->  trailing mel_cache_len frames cut.
The per-request evolving cache (use_detokenizer_cache=True) is not built.
"""
import ctypes
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .. import _native as N
from .cosyvoice_flow import CosyVoice2Flow, FlowConfig
from .hifigan import HiFTConfig, HiFTGenerator


@dataclass
class CosyVoice2DecoderCache:
    """Handle of the static prompt caches (they live inside the native flow object)."""
    prompt_tokens: int = 0
    prompt_mels: Optional[torch.Tensor] = None


class CosyVoice2Decoder:
    S3GEN_SR = 24000
    MAX_CACHE_LEN = 128
    PREFIX_LEN = 16

    def __init__(self, flow_weights: Dict[str, torch.Tensor], hift_weights: Dict[str, torch.Tensor], device="cuda",
                 flow_config: Optional[FlowConfig] = None, hift_config: Optional[HiFTConfig] = None, shared_prompt_cache_mode: bool = True,
                 max_batch: int = 8, max_tokens_per_chunk: int = 28, max_prompt_tokens: int = 256, seed: int = 0):
        if not shared_prompt_cache_mode:
            raise NotImplementedError("CosyVoice2Decoder: only the shared prompt cache mode (the plugin's default) is built")
        self.device = torch.device(device)
        self.shared_prompt_cache_mode = True
        self.flow = CosyVoice2Flow(flow_weights, flow_config, device=device, max_batch=max_batch, max_T=max_tokens_per_chunk,
                                   max_prompt_T=max_prompt_tokens, seed=seed)
        self.hift = HiFTGenerator(hift_weights, hift_config, device=device, max_batch=max_batch, max_T=2 * max_tokens_per_chunk, seed=seed)
        self.mel_cache_len = 6
        self.source_cache_len = int(self.mel_cache_len * 480)
        self.speech_window = torch.from_numpy(np.hamming(2 * self.source_cache_len)).to(self.device)      # float64, like the reference
        L = N.lib()
        L.vox_fade_in_out.restype = ctypes.c_int
        L.vox_fade_in_out.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        self.L = L

    def init_cache(self, ref_dict: dict, noise: Optional[torch.Tensor] = None) -> CosyVoice2DecoderCache:
        """cosyvoice2.py:862-942: the flow over prompt tokens (+ the first three again) with the prompt mel as condition; the caches
        are truncated to the sliding window and kept on the device."""
        tok = ref_dict["prompt_speech_token"]
        mels = self.flow.set_prompt(tok, ref_dict["prompt_feat"].float(), ref_dict["embedding"].float(), noise=noise)
        return CosyVoice2DecoderCache(prompt_tokens=int(tok.numel()), prompt_mels=mels)

    @torch.inference_mode()
    def decode_chunk(self, speech_tokens: torch.Tensor, speech_token_lens: int = None, decoder_cache: CosyVoice2DecoderCache = None,
                     ref_dict: Optional[dict] = None, last_chunk: bool = False, flow_noise: Optional[torch.Tensor] = None,
                     hift_noise: Optional[torch.Tensor] = None, hift_stream_base: Optional[torch.Tensor] = None,
                     flow_noise_stream: Optional[int] = None) -> Tuple[torch.Tensor, CosyVoice2DecoderCache]:
        """speech_tokens [B, T] -> (audio fp32 [B, 2 T * 480 - 2880], the same cache)   (cosyvoice2.py:944-1063, shared mode)"""
        if speech_tokens.dim() == 1:
            speech_tokens = speech_tokens.unsqueeze(0)
        mels = self.flow.forward_chunk(speech_tokens, noise=flow_noise, noise_stream=flow_noise_stream)
        wav, _ = self.hift.forward_chunk(mels, noise=hift_noise, stream_base=hift_stream_base)
        B, Lw = wav.shape
        N.check(self.L.vox_fade_in_out(N.stream(), wav.data_ptr(), B, Lw, None, self.speech_window.data_ptr(), self.source_cache_len))
        return wav[:, : Lw - self.source_cache_len], decoder_cache

    def close(self):
        self.flow.close()
        self.hift.close()
