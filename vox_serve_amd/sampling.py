"""Drop-in for /root/reference/vox_serve/sampling.py: SamplingConfig + Sampler with the same dispatch
order (sampling.py:85-118), repetition-penalty semantics (sampling.py:122-146) and cache update incl. the
reference's cross-request leak (sampling.py:150-178).  Runs in libvoxhip's on-device
sampler; stochastic modes draw from a seeded Philox4x32-10 stream (the reference's draw is
flashinfer-internal and not reproducible; the support set / probabilities follow its contract).
"""
import itertools
from dataclasses import dataclass
from typing import Optional

import torch

from . import _native as N


@dataclass
class SamplingConfig:
    top_p: Optional[float] = None
    top_k: Optional[int] = None
    min_p: Optional[float] = None
    temperature: float = 1.0
    max_tokens: Optional[int] = None
    repetition_penalty: Optional[float] = None
    repetition_window: Optional[int] = None  # -1 for global window
    cfg_scale: Optional[float] = None
    greedy: bool = False


def native_config(config: SamplingConfig) -> N.SamplingCfg:
    """SamplingConfig -> vox_sampling_config following Sampler.run_sampling's dispatch order."""
    greedy = config.greedy or config.temperature == 0.0
    top_k, top_p, min_p = 0, 1.0, 0.0
    if not greedy:
        if config.top_k is not None and config.top_p is not None:
            top_k, top_p = config.top_k, config.top_p
        elif config.top_k is not None:
            top_k = config.top_k
        elif config.top_p is not None:
            top_p = config.top_p
        elif config.min_p is not None:
            min_p = config.min_p
        else:
            greedy = True
    return N.SamplingCfg(int(greedy), int(top_k), float(top_p), float(min_p),
                         float(config.temperature if not greedy else 1.0), 1.0)


def _sample(logits: torch.Tensor, nc: N.SamplingCfg) -> torch.Tensor:
    """One on-device sampler call over the rows of `logits` [..., V] (vox_sample); ids shaped like logits.shape[:-1]."""
    lead, v = logits.shape[:-1], logits.shape[-1]
    lg = logits.reshape(-1, v).contiguous()
    if lg.dtype != torch.bfloat16:
        lg = lg.to(torch.bfloat16)
    out = torch.empty(lg.shape[0], dtype=torch.int32, device=lg.device)
    N.check(N.lib().vox_sample(N.ctx(), N.stream(), N.ptr(lg), lg.shape[0], v, nc, Sampler.seed, next(Sampler._offset), N.ptr(out)))
    return (out.long() if nc.greedy else out).reshape(lead)


# Module-level helpers of the reference (sampling.py:21-80: thin wrappers that `Sampler.run_sampling` dispatches to; nothing else in
# the reference imports them, they are here so that the module is a drop-in symbol for symbol).  Same arguments; the stochastic ones
# draw from the seeded Philox stream of `Sampler` (manual_seed), their support set and probabilities follow flashinfer's contract.
def greedy_sampling(logits):
    """sampling.py:21-27: index of the (first) maximum logit, int64."""
    return _sample(logits, N.SamplingCfg(1, 0, 1.0, 0.0, 1.0, 1.0))


def top_k_sampling(logits, top_k, temperature):
    """sampling.py:30-40."""
    return _sample(logits, N.SamplingCfg(0, int(top_k), 1.0, 0.0, float(temperature), 1.0))


def top_p_sampling(logits, top_p, temperature):
    """sampling.py:43-53."""
    return _sample(logits, N.SamplingCfg(0, 0, float(top_p), 0.0, float(temperature), 1.0))


def top_k_top_p_sampling(logits, top_k, top_p, temperature, filter_apply_order="top_k_first"):
    """sampling.py:56-67 (the reference only ever passes "top_k_first": top-p renormalised inside the top-k set)."""
    if filter_apply_order != "top_k_first":
        raise NotImplementedError("filter_apply_order='joint' is not part of the serving path (sampling.py:56 default only)")
    return _sample(logits, N.SamplingCfg(0, int(top_k), float(top_p), 0.0, float(temperature), 1.0))


def min_p_sampling(logits, min_p, temperature):
    """sampling.py:70-80."""
    return _sample(logits, N.SamplingCfg(0, 0, 1.0, float(min_p), float(temperature), 1.0))


class Sampler:
    seed = 0
    _offset = itertools.count()

    @classmethod
    def manual_seed(cls, seed: int):
        cls.seed, cls._offset = int(seed), itertools.count()

    @classmethod
    def run_sampling(cls, logits: torch.Tensor, config: SamplingConfig) -> torch.Tensor:
        lg = logits.contiguous()
        if lg.dtype != torch.bfloat16:
            lg = lg.to(torch.bfloat16)
        b, v = lg.shape
        out = torch.empty(b, dtype=torch.int32, device=lg.device)
        nc = native_config(config)
        N.check(N.lib().vox_sample(N.ctx(), N.stream(), N.ptr(lg), b, v, nc, cls.seed, next(cls._offset), N.ptr(out)))
        return out.long() if nc.greedy else out

    @classmethod
    def apply_repetition_penalty(cls, logits: torch.Tensor, repetition_cache: torch.Tensor, penalty: float):
        """sampling.py:122-146: logits [B, Cl, V], cache [B, W, C, V]; Cl == 1 reads cache codebook 0, Cl == C codebook c per row."""
        b, c1, v = logits.shape
        out = logits.contiguous().clone()
        cache = repetition_cache.contiguous().view(torch.uint8)
        _, w, c, _ = cache.shape
        if c1 == 1:
            N.check(N.lib().vox_rep_penalty(N.ctx(), N.stream(), N.ptr(out), N.ptr(cache), b, w, c, v, float(penalty)))
        else:
            if c1 != c:
                raise ValueError(f"logits cover {c1} codebooks, the repetition cache {c}")       # (torch.where would not broadcast either)
            N.check(N.lib().vox_rep_penalty_mc(N.ctx(), N.stream(), N.ptr(out), N.ptr(cache), b, c1, w, c, v, float(penalty)))
        return out

    @classmethod
    def update_repetition_penalty_cache(cls, repetition_cache: torch.Tensor, output_ids: torch.Tensor, window_size: int):
        """sampling.py:150-178, in place.  output_ids [B, 1] with a C-codebook cache: the codebook-0 form (incl. the reference's
        cross-request leak); output_ids [B, C]: the reference's `cache[:, w, :, output_ids] = True` — every id in every plane."""
        assert repetition_cache.is_contiguous()
        b, w, c, v = repetition_cache.shape
        cl = output_ids.shape[1]
        cache8 = repetition_cache.view(torch.uint8)
        if cl == 1:
            ids = output_ids[:, 0].to(torch.int32).contiguous()
            N.check(N.lib().vox_rep_update(N.ctx(), N.stream(), N.ptr(cache8), N.ptr(ids), b, w, c, v, int(window_size)))
        else:
            ids = output_ids.to(torch.int32).contiguous()
            N.check(N.lib().vox_rep_update_mc(N.ctx(), N.stream(), N.ptr(cache8), N.ptr(ids), b, cl, w, c, v, int(window_size)))
