"""TEST INFRASTRUCTURE ONLY (see oracle/voxref.c header).  The configuration of the g22 reference fixtures
(tests/golden/make_goldens.py::g22_glm_full_width): ONE GLM-4-Voice-9B layer at its full width — every reduction length
(K = 4096 / 13696), head shape (32 q heads on 2 kv heads of 128, half-rotary interleaved RoPE, QKV bias) of the real model
(/root/reference/vox_serve/model/glm_voice.py:22-55, 104-305) — with a 4096-entry vocabulary, which keeps the two
168960 x 4096 tables out of the fixture's way (the full tables are covered by the oracle tapes of tests/test_gpu_lm.py).
Kept apart from lm_ref.py, whose text is part of the oracle tapes' freshness hash."""
from .lm_ref import LMCfg, glm_cfg

WEIGHT_SEED, WEIGHT_STD = 22, 0.02


def wide_glm_cfg() -> LMCfg:
    return glm_cfg(layers=1, vocab=4096, max_pos=512)
