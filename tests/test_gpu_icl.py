"""GPU: voice cloning through the Qwen3-TTS plugin — the prompt layout + input_features of x-vector-only and ICL cloning
equal the reference's `preprocess` (g14) bit for bit, and an ICL request driven through ModelWorker produces the oracle's
token stream."""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def build_base(dev, golden_ids, **override):
    from oracle import qwen3_codec_ref as CR, qwen3_ref as QR, voxref as vr
    from tests.test_gpu_codec import small_cfg
    from tests.test_gpu_qwen3 import to_engine_cfg
    from vox_serve_amd.model.qwen3_tts import Qwen3TTSModel, Qwen3TTSTokens
    from vox_serve_amd.sampling import SamplingConfig
    from vox_serve_amd.tokenizer.qwen3_codec import Qwen3CodecConfig
    ids = dict(golden_ids, **override)
    spk_id, dialect, lang = ids.pop("spk_id"), ids.pop("spk_is_dialect"), ids.pop("codec_language_id")
    cfg = QR.tiny_cfg()
    Wn = QR.random_weights(cfg, seed=0, std=0.08)
    W = {k: vr.to_torch(v).to(dev) for k, v in Wn.items()}
    cc = small_cfg()
    pc = Qwen3CodecConfig(**{k: getattr(cc, k) for k in Qwen3CodecConfig.__dataclass_fields__})
    toks = Qwen3TTSTokens(**ids, codec_eos=cfg.eos_id, codec_language_id=lang, spk_id=spk_id, spk_is_dialect=dialect)
    m = Qwen3TTSModel("tiny-base", W, CR.random_codec_weights(cc, 3), config=to_engine_cfg(cfg), codec_config=pc, tokens=toks,
                      device=str(dev), detokenize_interval=4, max_batch_size=4, page_size=16, max_num_pages=64,
                      max_seq_len=512, max_prefill_tokens=64, tts_model_type="base")
    m.default_sampling_config = SamplingConfig(greedy=True, max_tokens=None)
    return m, cfg, Wn


def test_clone_prompts_equal_the_reference_preprocess(golden):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import voxref as vr
    dev = torch.device("cuda:0")
    g = golden("g14_qwen3_preprocess")
    m, cfg, _ = build_base(dev, json.loads(str(g["special_ids"])))
    spk = vr.to_torch(g["spk_embedding"])
    codes = torch.from_numpy(g["ref_codes"])
    n = 0
    for tag, kind, kw in json.loads(str(g["cases"])):
        if kind != "base":
            continue
        icl = not kw.get("x_vector_only_mode")
        po = m.preprocess(prompt_token_ids=g[f"{tag}_prompt_ids"].tolist(), language=kw["language"],
                          instruct_token_ids=g[f"{tag}_instruct_ids"].tolist() if f"{tag}_instruct_ids" in g else None,
                          is_input_streaming=bool(kw.get("is_input_streaming")), x_vector_only_mode=not icl,
                          speaker_embedding=spk, ref_codes=codes if icl else None,
                          ref_text_token_ids=g[f"{tag}_ref_text_ids"].tolist() if icl else None)
        assert torch.equal(po.input_tokens, torch.from_numpy(g[f"{tag}_tokens"])), tag
        assert torch.equal(po.input_masks, torch.from_numpy(g[f"{tag}_masks"])), tag
        assert np.array_equal(vr.from_torch(po.input_features), g[f"{tag}_features"]), tag
        n += 1
    assert n == 4
    with pytest.raises(ValueError):
        m.preprocess(prompt_token_ids=[1, 2, 3, 4, 5, 6, 7, 8, 9])          # cloning without a reference voice
    m.engine.close()


def test_icl_request_through_the_worker_equals_the_oracle(golden):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import qwen3_ref as QR, voxref as vr
    from tests.test_gpu_worker import _drive_worker
    dev = torch.device("cuda:0")
    g = golden("g14_qwen3_preprocess")
    # text-side ids inside the tiny text vocabulary (512 rows); the layout itself is pinned by the test above
    m, cfg, Wn = build_base(dev, json.loads(str(g["special_ids"])), tts_bos=5, tts_eos=6, tts_pad=7)
    kw = {"prompt_token_ids": [11, 12, 13] + list(range(100, 109)) + [21, 22, 23, 24, 25], "language": "english",
          "speaker_embedding": vr.to_torch(g["spk_embedding"]).float().tolist(), "ref_codes": g["ref_codes"].tolist(),
          "ref_text_token_ids": [11, 12, 13] + list(range(200, 206)) + [21, 22]}
    po = m.preprocess(**kw)
    assert po.input_tokens.shape[0] == 3 + 4 + 1 + 1 + 6 + 9 + 1 + 1 + 7
    reqs, w = _drive_worker(m, [kw], 6)
    ref = QR.Qwen3Ref(cfg, Wn, page_size=16, max_pages=64, max_batch=1)
    q = QR.RefRequest()
    toks = po.input_tokens.numpy().astype(np.int32)
    lg, hid = ref.prefill(q, toks, po.input_masks[:, -1].numpy().astype(np.uint8), vr.from_torch(po.input_features))
    frames = [ref.frame([q], lg, hid)[0][0]]
    for _ in range(6):
        frames.append(ref.frame([q])[0][0])
    got = [t[0, :cfg.n_groups].tolist() for t in reqs[0].lm_output_tokens]
    want = [f[:cfg.n_groups].tolist() for f in frames[: len(got)]]
    assert len(got) >= 5 and got == want, (got, want)
    m.engine.close()
