"""CosyVoice2 detokenizer (speech tokens -> waveform) on libvoxhip: drop-in surface of the reference's `CosyVoice2Decoder`
(/root/reference/vox_serve/tokenizer/cosyvoice2.py:774-1046) in the mode the plugin serves by default — the shared prompt cache
(model/cosyvoice2.py:325, 1093-1103): `init_cache(ref_dict)` runs the flow over the speaker prompt once and keeps its caches,
`decode_chunk(speech_tokens, ...)` decodes every 28-token window of every request against them:
    flow (conformer encoder + 10-step CFM)  ->  HiFT vocoder  ->  fade-in over mel_cache_len frames  ->  trailing mel_cache_len frames cut.
With shared_prompt_cache_mode=False (the plugin's use_detokenizer_cache=True, cosyvoice2.py:1010-1083) every request owns evolving caches:
a native slot that starts as a copy of the prompt's caches, is read and then advanced by each of its chunks (sliding window), plus the tail of
its previous chunk's audio for the fade-in.
"""
import ctypes
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .. import _native as N
from .base import DecoderCache, device_bound
from .cosyvoice_flow import CosyVoice2Flow, FlowConfig
from .hifigan import HiFTConfig, HiFTGenerator


@dataclass
class CosyVoice2DecoderCache(DecoderCache):
    """Shared mode: handle of the static prompt caches (they live inside the native flow object).  Per-request mode: `slot` [B] int32 names
    the native slots holding each request's evolving flow caches, `speech_cache` [B, mel_cache_len * 480] is the tail of each request's
    previous chunk (the HiFTGeneratorCache.speech_cache of the reference)."""
    prompt_tokens: int = 0
    prompt_mels: Optional[torch.Tensor] = None
    slot: Optional[torch.Tensor] = None
    speech_cache: Optional[torch.Tensor] = None


@device_bound
class CosyVoice2Decoder:
    S3GEN_SR = 24000
    MAX_CACHE_LEN = 128
    PREFIX_LEN = 16

    def __init__(self, flow_weights: Dict[str, torch.Tensor], hift_weights: Dict[str, torch.Tensor], device="cuda",
                 flow_config: Optional[FlowConfig] = None, hift_config: Optional[HiFTConfig] = None, shared_prompt_cache_mode: bool = True,
                 max_batch: int = 8, max_tokens_per_chunk: int = 28, max_prompt_tokens: int = 256, seed: int = 0, max_slots: int = 16):
        self.device = torch.device(device)
        self.shared_prompt_cache_mode = bool(shared_prompt_cache_mode)
        self.max_slots, self._free_slots = max_slots, list(range(max_slots))
        self.flow = CosyVoice2Flow(flow_weights, flow_config, device=device, max_batch=max_batch, max_T=max_tokens_per_chunk,
                                   max_prompt_T=max_prompt_tokens, seed=seed)
        self.hift = HiFTGenerator(hift_weights, hift_config, device=device, max_batch=max_batch, max_T=2 * max_tokens_per_chunk, seed=seed)
        self.mel_cache_len = 6
        self.source_cache_len = int(self.mel_cache_len * 480)
        self.speech_window = torch.from_numpy(np.hamming(2 * self.source_cache_len)).to(self.device)      # float64, like the reference
        self.seed, self.use_graph, self._graphs, self._chunk = seed, True, {}, 0
        L = N.lib()
        L.vox_flow_fill_noise.restype = ctypes.c_int
        L.vox_flow_fill_noise.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.vox_fade_in_out.restype = ctypes.c_int
        L.vox_fade_in_out.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        self.L = L
        if not self.shared_prompt_cache_mode:
            self.flow.enable_slots(max_slots)

    # ---- per-request caches (shared_prompt_cache_mode=False) ----
    def new_request_cache(self, batch_size: int = 1) -> CosyVoice2DecoderCache:
        """model/cosyvoice2.py:514-560 `audio_decoder_initial_cache`: each new request gets a copy of the prompt's caches (init_cache must
        have run) and a silent speech tail."""
        if self.shared_prompt_cache_mode:
            raise RuntimeError("CosyVoice2Decoder: per-request caches need shared_prompt_cache_mode=False")
        slots = []
        for _ in range(batch_size):
            if not self._free_slots:
                raise RuntimeError("CosyVoice2Decoder: no free detokenizer cache slot (raise max_slots)")
            sl = self._free_slots.pop(0)
            self.flow.slot_reset(sl)
            slots.append(sl)
        return CosyVoice2DecoderCache(slot=torch.tensor(slots, dtype=torch.int32),
                                      speech_cache=torch.zeros(batch_size, self.source_cache_len, dtype=torch.float32, device=self.device))

    def release_cache(self, cache: CosyVoice2DecoderCache):
        if cache is not None and cache.slot is not None:
            for sl in cache.slot.tolist():
                if sl not in self._free_slots:
                    self._free_slots.append(int(sl))

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
        if not self.shared_prompt_cache_mode:
            return self._decode_chunk_evolving(speech_tokens, decoder_cache, flow_noise, hift_noise, hift_stream_base, flow_noise_stream)
        if (self.use_graph and flow_noise is None and hift_noise is None and hift_stream_base is None and flow_noise_stream is None
                and speech_tokens.shape[0] <= self.flow.max_batch):
            return self._decode_chunk_graph(speech_tokens), decoder_cache
        mels = self.flow.forward_chunk(speech_tokens, noise=flow_noise, noise_stream=flow_noise_stream)
        wav, _ = self.hift.forward_chunk(mels, noise=hift_noise, stream_base=hift_stream_base)
        B, Lw = wav.shape
        N.check(self.L.vox_fade_in_out(N.stream(), wav.data_ptr(), B, Lw, None, self.speech_window.data_ptr(), self.source_cache_len))
        return wav[:, : Lw - self.source_cache_len], decoder_cache

    def _decode_chunk_evolving(self, speech_tokens, cache: CosyVoice2DecoderCache, flow_noise, hift_noise, hift_stream_base, flow_noise_stream):
        """cosyvoice2.py:1010-1083: the flow against each request's own caches (advanced in place by the native call), HiFT, the fade-in
        against the request's previous tail; the new tail (the trimmed last mel_cache_len frames of the faded audio) is written back into
        cache.speech_cache in place, like the reference's `decoder_cache.copy_from(new_decoder_cache)` (model/cosyvoice2.py:1114)."""
        if cache is None or cache.slot is None:
            raise ValueError("CosyVoice2Decoder: the per-request mode needs the request's decoder_cache (new_request_cache)")
        mels = self.flow.forward_chunk_slots(speech_tokens, cache.slot.tolist(), noise=flow_noise, noise_stream=flow_noise_stream)
        wav, _ = self.hift.forward_chunk(mels, noise=hift_noise, stream_base=hift_stream_base)
        B, Lw = wav.shape
        prev = cache.speech_cache.to(self.device, torch.float32).contiguous()
        N.check(self.L.vox_fade_in_out(N.stream(), wav.data_ptr(), B, Lw, prev.data_ptr(), self.speech_window.data_ptr(), self.source_cache_len))
        cache.speech_cache.copy_(wav[:, Lw - self.source_cache_len:])
        return wav[:, : Lw - self.source_cache_len], cache

    def _decode_chunk_graph(self, speech_tokens: torch.Tensor) -> torch.Tensor:
        """The ~7 000 launches of a chunk (flow: 10 estimator passes of 70 blocks; HiFT) as one hipGraph per (requests, tokens): inputs go
        into graph-stable buffers, the CFM start noise is drawn into one before the replay (a graph would freeze its stream id), the
        vocoder's noise streams are read from a device array.  The first chunk of a shape runs eagerly, the second is captured.  The
        returned tensor is a fresh copy."""
        B, T = speech_tokens.shape
        c, L = self.flow.cfg, self.L
        key = (B, T)
        ent = self._graphs.get(key)
        if ent is None:
            Lw = 2 * T * self.hift.upsample_scale
            ent = self._graphs[key] = {"tok": torch.empty(B, T, dtype=torch.int32, device=self.device),
                                       "z": torch.empty(c.mel, 2 * T, dtype=torch.float32, device=self.device),
                                       "sb": torch.empty(B, dtype=torch.int32, device=self.device),
                                       "mel": torch.empty(B, c.mel, 2 * T, dtype=torch.float32, device=self.device),
                                       "wav": torch.empty(B, Lw, dtype=torch.float32, device=self.device), "g": None, "calls": 0}
        self._chunk += 1
        # on the caller's current stream (N.graph_capture: why the decoder owns none)
        ent["tok"].copy_(speech_tokens.to(self.device, torch.int32), non_blocking=True)
        ent["sb"].copy_(((torch.arange(B, dtype=torch.int64) + self._chunk * 65536) * 2).to(torch.int32), non_blocking=True)
        N.check(L.vox_flow_fill_noise(N.stream(), ctypes.c_uint64(self.seed), self._chunk, c.mel, 2 * T, ent["z"].data_ptr()))

        def body():
            st = N.stream()
            N.check(self.flow.L.vox_flow_decode_chunk(self.flow.h, st, ent["tok"].data_ptr(), B, T, ent["z"].data_ptr(), ctypes.c_uint64(self.seed), 0,
                                                      ent["mel"].data_ptr(), None))
            N.check(self.hift.L.vox_hift_decode(self.hift.h, st, ent["mel"].data_ptr(), B, 2 * T, None, ctypes.c_uint64(self.seed),
                                                ent["sb"].data_ptr(), ent["wav"].data_ptr(), None, None))
            N.check(L.vox_fade_in_out(st, ent["wav"].data_ptr(), B, ent["wav"].shape[1], None, self.speech_window.data_ptr(), self.source_cache_len))
        ent["calls"] += 1
        if ent["calls"] == 1:
            body()
        else:
            if ent["g"] is None:
                with N.graph_capture() as cap:
                    body()
                ent["g"] = cap.graph
            N.check(L.vox_graph_launch(ent["g"], N.stream()))
        out = ent["wav"][:, : ent["wav"].shape[1] - self.source_cache_len].clone()
        return out

    def close(self):
        self.flow.close()
        self.hift.close()
