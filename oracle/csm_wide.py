"""TEST INFRASTRUCTURE ONLY (see oracle/voxref.c header).  The configuration of the g23 reference fixture
(tests/golden/make_goldens.py::g23_csm_full_width): ONE backbone layer + ONE depth-decoder layer at the full widths of CSM-1B
(/root/reference/vox_serve/model/csm.py:55-313: backbone 2048 wide, 32 q heads on 8 kv heads of 64, FFN 8192; depth decoder 1024
wide, 8 q heads on 2 kv heads of 128, FFN 8192; 32 codebooks of 2051 entries; llama-3.1 RoPE over the real 8192-token original
context) with a 1024-entry text vocabulary, which keeps the 128256 x 2048 text table out of the fixture's way.  Kept apart from
csm_ref.py, whose text is part of the oracle tapes' freshness hash."""
from .csm_ref import CSMCfg, StackCfg

WEIGHT_SEED, WEIGHT_STD = 23, 0.02


def wide_csm_cfg() -> CSMCfg:
    ll = (1.0, 4.0, 8192)
    return CSMCfg(backbone=StackCfg(2048, 1, 32, 8, 64, 8192, eps=1e-5, rope_theta=5e5, rope_scale=32.0, rope_llama31=ll, qk_norm=False),
                  depth=StackCfg(1024, 1, 8, 2, 128, 8192, eps=1e-5, rope_theta=5e5, rope_scale=32.0, rope_llama31=ll, qk_norm=False),
                  vocab=2051, text_vocab=1024, n_codebooks=32, max_pos=512)
