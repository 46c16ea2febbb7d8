"""CPU oracle of the multi-codebook forms of Sampler.apply_repetition_penalty / update_repetition_penalty_cache
(/root/reference/vox_serve/sampling.py:122-178 with logits [B, C, V] and output_ids [B, C], C > 1), numpy.

TEST INFRASTRUCTURE ONLY.  Pinned to the reference by tests/golden/g19_sampler_mc.npz (tests/test_oracle_goldens.py)."""
import numpy as np

from . import voxref as vr


def rep_penalty_mc(logits_bits: np.ndarray, cache: np.ndarray, penalty: float) -> np.ndarray:
    """logits [B, C, V] bf16 bits, cache [B, W, C, V] uint8 -> penalised logits (sampling.py:137-146): a token seen in any window
    slot of ITS codebook is divided (logit > 0) or multiplied (logit <= 0) by the penalty, one bf16 rounding."""
    seen = cache.any(axis=1)                                            # [B, C, V]
    x = vr.bf2f(logits_bits)
    y = np.where(x > 0, x / np.float32(penalty), x * np.float32(penalty)).astype(np.float32)
    return np.where(seen, vr.f2bf(y), logits_bits).astype(np.uint16)


def rep_update_mc(cache: np.ndarray, ids: np.ndarray, window: int) -> None:
    """cache [B, W, C, V] uint8 in place, ids [B, C] (sampling.py:166-178): sliding window -> shift left, clear the newest slot,
    then `cache[:, -1, :, ids] = True`; global window -> `cache[:, :, :, ids] = True`.  The advanced index takes every id of the
    step for every batch row and codebook plane (the reference's semantics, leak included)."""
    flat = np.asarray(ids).reshape(-1)
    if window > 1:
        cache[:, :-1] = cache[:, 1:].copy()
        cache[:, -1] = 0
        cache[:, -1][:, :, flat] = 1
    else:
        cache[:, :, :, flat] = 1
