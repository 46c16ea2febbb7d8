"""TEST INFRASTRUCTURE ONLY (see oracle/voxref.c header).  The configuration of the g21 reference fixtures
(tests/golden/make_goldens.py::g21_qwen3_full_width): ONE talker layer + ONE depth layer at the full widths of Qwen3-TTS-1.7B —
every reduction length (K = 2048 / 6144 / 1024 / 3072) and head shape of the real model
(/root/reference/vox_serve/model/qwen3_tts.py:562-704); a small text vocabulary keeps the embedding table out of the
fixture's way.  Kept apart from qwen3_ref.py, whose text is part of the oracle tapes' freshness hash."""
from .qwen3_ref import Qwen3Cfg, StackCfg

WEIGHT_SEED, WEIGHT_STD = 21, 0.02


def wide_cfg() -> Qwen3Cfg:
    return Qwen3Cfg(talker=StackCfg(2048, 1, 16, 8, 128, 6144), depth=StackCfg(1024, 1, 16, 8, 128, 3072),
                    vocab=3072, text_vocab=1024, text_hidden=2048, depth_vocab=2048, n_groups=16, eos_id=2150,
                    tts_pad_id=7, max_pos=512)
