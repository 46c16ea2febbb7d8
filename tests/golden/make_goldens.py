"""Generate the committed golden fixtures by RUNNING THE REFERENCE (vox-serve) on CPU.

Runs only in the build container (needs /root/reference).  Output: tests/golden/*.npz — data only
(inputs + the reference's outputs).  See _ref_harness.py for how the reference is imported.

  python tests/golden/make_goldens.py [g1 g2 g3 g4 ...]
"""
import json
import os
import queue
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _ref_harness as H  # noqa: E402

from oracle import voxref as vr  # noqa: E402  (bf16<->numpy helpers + weight recipe only)
from oracle import qwen3_ref as QR  # noqa: E402
from oracle import qwen3_wide as QW  # noqa: E402


def bits(t):
    return vr.from_torch(t.to(torch.bfloat16))


# --------------------------------------------------------------------------------------------------
def g1_sampler(ns):
    """Sampler: greedy ids, suppress mask, repetition penalty incl. quirks Q2/Q3 (sampling.py:21-178)."""
    S = ns.sampling
    out = {}
    g = torch.Generator().manual_seed(42)
    for name, (B, V) in {"a": (1, 3072), "b": (8, 3072), "c": (32, 3072), "d": (8, 2048), "e": (4, 168960)}.items():
        logits = torch.randn(B, V, generator=g).to(torch.bfloat16)
        out[f"greedy_{name}_logits"] = bits(logits)
        out[f"greedy_{name}_ids"] = S.Sampler.run_sampling(logits, S.SamplingConfig(greedy=True)).numpy().astype(np.int32)
    # ties: argmax must return the FIRST maximal index
    t = torch.zeros(3, 64, dtype=torch.bfloat16)
    t[0, [5, 9]] = 1.0
    t[1, [63, 0]] = 2.0
    t[2, :] = -1.0
    out["greedy_tie_logits"], out["greedy_tie_ids"] = bits(t), S.greedy_sampling(t).numpy().astype(np.int32)

    # repetition penalty: logits [B,1,V], cache [B,W,C,V]
    B, W, C, V = 3, 2, 5, 256
    logits = torch.randn(B, 1, V, generator=g).to(torch.bfloat16)
    cache = torch.rand(B, W, C, V, generator=g) < 0.1
    pen = S.Sampler.apply_repetition_penalty(logits.clone(), cache, 1.3)
    out["pen_logits"], out["pen_cache"], out["pen_out"] = bits(logits), cache.numpy().astype(np.uint8), bits(pen)
    # cache update, global window (Qwen3: W=1) and sliding window; codebook-0-only form => leak Q2
    for tag, (Wn, window) in {"glob": (1, -1), "win": (3, 3)}.items():
        c = torch.rand(B, Wn, C, V, generator=g) < 0.02
        ids = torch.tensor([[5], [7], [9]])
        c2 = c.clone()
        S.Sampler.update_repetition_penalty_cache(c2, ids, window)
        out[f"upd_{tag}_in"], out[f"upd_{tag}_ids"], out[f"upd_{tag}_out"] = (
            c.numpy().astype(np.uint8), ids.numpy().astype(np.int32)[:, 0], c2.numpy().astype(np.uint8))
    # temperature scaling happens in the logits dtype (sampling.py:31): bf16(l/T)
    l = torch.randn(4, 512, generator=g).to(torch.bfloat16)
    out["temp_logits"], out["temp_out"] = bits(l), bits(l / 0.9)
    np.savez_compressed(os.path.join(HERE, "g1_sampler.npz"), **out)
    print("g1 ok", len(out))


def g19_sampler_mc(ns):
    """Sampler.apply_repetition_penalty / update_repetition_penalty_cache in their multi-codebook forms (logits [B, C, V],
    output_ids [B, C], C > 1: sampling.py:122-178)."""
    S = ns.sampling
    out = {}
    g = torch.Generator().manual_seed(19)
    B, W, C, V = 3, 2, 4, 320
    logits = torch.randn(B, C, V, generator=g).to(torch.bfloat16)
    cache = torch.rand(B, W, C, V, generator=g) < 0.1
    pen = S.Sampler.apply_repetition_penalty(logits.clone(), cache, 1.3)
    out["pen_logits"], out["pen_cache"], out["pen_out"] = bits(logits), cache.numpy().astype(np.uint8), bits(pen)
    for tag, (Wn, window) in {"glob": (1, -1), "win": (3, 3)}.items():
        c = torch.rand(B, Wn, C, V, generator=g) < 0.02
        ids = torch.randint(0, V, (B, C), generator=g)
        c2 = c.clone()
        S.Sampler.update_repetition_penalty_cache(c2, ids, window)
        out[f"upd_{tag}_in"], out[f"upd_{tag}_ids"], out[f"upd_{tag}_out"] = (
            c.numpy().astype(np.uint8), ids.numpy().astype(np.int32), c2.numpy().astype(np.uint8))
    np.savez_compressed(os.path.join(HERE, "g19_sampler_mc.npz"), **out)
    print("g19 ok", len(out))


# --------------------------------------------------------------------------------------------------
def g2_wrappers(ns):
    """plan() page/slot arithmetic and set_kv_cache layout (flashinfer_utils.py:60-145,189-244)."""
    FU = ns.flashinfer_utils
    out = {}
    rng = np.random.default_rng(7)
    page, Hq, Hkv, D, P = 4, 4, 2, 16, 24
    cpu = torch.device("cpu")
    # prefill: 3 requests, ragged; one of them a 1-token "decode piggy-back" with existing context
    q_lens = [6, 1, 9]
    kv_lens = [6, 11, 9]
    pages, indptr, last = [], [0], []
    free = list(rng.permutation(P))
    for n in kv_lens:
        k = (n + page - 1) // page
        pages += [int(free.pop()) for _ in range(k)]
        indptr.append(indptr[-1] + k)
        last.append(n % page or page)
    qo = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
    w = FU.FlashInferPrefillWrapper(torch.empty(1), Hq, Hkv, Hq * D, page, device=cpu)
    w.plan(torch.tensor(qo), torch.tensor(indptr, dtype=torch.int32), torch.tensor(pages, dtype=torch.int32),
           torch.tensor(last, dtype=torch.int32), torch.bfloat16)
    T = int(qo[-1])
    g = torch.Generator().manual_seed(3)
    kv = (torch.randn(P, 2, page, Hkv, D, generator=g)).to(torch.bfloat16)
    k = torch.randn(T, Hkv, D, generator=g).to(torch.bfloat16)
    v = torch.randn(T, Hkv, D, generator=g).to(torch.bfloat16)
    q = torch.randn(T, Hq, D, generator=g).to(torch.bfloat16)
    kv_in = kv.clone()
    w.set_kv_cache(kv, k, v)
    o = w.run(q, kv)
    out.update(pf_qo=qo, pf_indptr=np.array(indptr, np.int32), pf_indices=np.array(pages, np.int32),
               pf_last=np.array(last, np.int32), pf_token_to_page=w.token_to_page.numpy().astype(np.int32),
               pf_token_to_cache=w.token_to_cache.numpy().astype(np.int32), pf_kv_in=bits(kv_in), pf_k=bits(k),
               pf_v=bits(v), pf_q=bits(q), pf_kv_out=bits(kv), pf_out=bits(o), page=np.int32(page))
    # decode
    d = FU.FlashInferDecodeWrapper(torch.empty(1), Hq, Hkv, Hq * D, page, device=cpu)
    d.plan(torch.tensor(indptr, dtype=torch.int32), torch.tensor(pages, dtype=torch.int32),
           torch.tensor(last, dtype=torch.int32), torch.bfloat16)
    B = len(kv_lens)
    k1 = torch.randn(B, Hkv, D, generator=g).to(torch.bfloat16)
    v1 = torch.randn(B, Hkv, D, generator=g).to(torch.bfloat16)
    q1 = torch.randn(B, Hq, D, generator=g).to(torch.bfloat16)
    kv2 = kv.clone()
    d.set_kv_cache(kv2, k1, v1)
    o1 = d.run(q1, kv2)
    out.update(dc_loc=d.kv_cache_locations.numpy().astype(np.int32), dc_k=bits(k1), dc_v=bits(v1), dc_q=bits(q1),
               dc_kv_out=bits(kv2), dc_out=bits(o1))
    # rms_norm / rope (three variants used by the model families)
    x = torch.randn(5, 256, generator=g).to(torch.bfloat16)
    wn = (1 + 0.1 * torch.randn(256, generator=g)).to(torch.bfloat16)
    out.update(rms_x=bits(x), rms_w=bits(wn), rms_y=bits(FU.rms_norm(x, wn, 1e-6)))
    qq = torch.randn(7, 4, 64, generator=g).to(torch.bfloat16)
    kk = torch.randn(7, 2, 64, generator=g).to(torch.bfloat16)
    pos = torch.tensor([0, 1, 2, 17, 100, 1000, 2047], dtype=torch.int32)
    out.update(rope_q=bits(qq), rope_k=bits(kk), rope_pos=pos.numpy())
    a, b = FU.apply_rope_pos_ids(qq, kk, pos, rope_theta=1e6, interleave=False)
    out.update(rope_neox_q=bits(a), rope_neox_k=bits(b))
    a, b = FU.apply_rope_pos_ids(qq, kk, pos, rope_theta=1e4, interleave=True, rotary_dim=32)
    out.update(rope_glm_q=bits(a), rope_glm_k=bits(b))
    a, b = FU.apply_rope_pos_ids(qq, kk, pos, rope_scale=32.0, rope_theta=5e5, interleave=False,
                                 low_freq_factor=1.0, high_freq_factor=4.0, old_context_len=8192)
    out.update(rope_l31_q=bits(a), rope_l31_k=bits(b))
    np.savez_compressed(os.path.join(HERE, "g2_wrappers.npz"), **out)
    print("g2 ok")


# --------------------------------------------------------------------------------------------------
def _ref_qwen3(ns, cfg: QR.Qwen3Cfg, W):
    """Build the reference Qwen3TTSForCausalLM + Qwen3TTSModel plugin (constructors bypassed) on CPU."""
    Q = ns.qwen3_tts
    t, d = cfg.talker, cfg.depth
    cp = Q.Qwen3TTSCodePredictorConfig(hidden_size=d.hidden, intermediate_size=d.ffn, num_hidden_layers=d.layers,
                                       num_attention_heads=d.heads, num_key_value_heads=d.kv_heads,
                                       head_dim=d.head_dim, vocab_size=cfg.depth_vocab, num_code_groups=cfg.n_groups,
                                       rope_theta=int(d.rope_theta))
    tc = Q.Qwen3TTSTalkerConfig(hidden_size=t.hidden, intermediate_size=t.ffn, num_hidden_layers=t.layers,
                                num_attention_heads=t.heads, num_key_value_heads=t.kv_heads, head_dim=t.head_dim,
                                vocab_size=cfg.vocab, text_vocab_size=cfg.text_vocab, text_hidden_size=cfg.text_hidden,
                                num_code_groups=cfg.n_groups, codec_eos_token_id=cfg.eos_id,
                                rope_theta=int(t.rope_theta), code_predictor_config=cp)
    rc = Q.Qwen3TTSConfig(tts_model_type="custom_voice", talker_config=tc, tts_pad_token_id=cfg.tts_pad_id)
    net = Q.Qwen3TTSForCausalLM(rc).to(torch.bfloat16)
    sd = {k: vr.to_torch(v) for k, v in W.items()}
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("lm_head_weight" in m or "inv_freq" in m for m in missing), missing
    net.talker.code_predictor.lm_head_weight.copy_(
        torch.stack([h.weight for h in net.talker.code_predictor.lm_head], dim=0))
    m = Q.Qwen3TTSModel.__new__(Q.Qwen3TTSModel)
    m.model, m.config, m.device, m.dtype = net, rc, "cpu", torch.bfloat16
    m.model_name = "tiny"
    m._detokenize_interval = 10
    m._num_attention_heads, m._num_key_value_heads = t.heads, t.kv_heads
    m._num_hidden_layers, m._hidden_size = t.layers, t.hidden
    m._depth_num_attention_heads, m._depth_num_key_value_heads = d.heads, d.kv_heads
    m._depth_num_hidden_layers, m._depth_hidden_size = d.layers, d.hidden
    m._vocab_size, m.stop_token_id = cfg.vocab, cfg.eos_id
    m.suppress_tokens = cfg.suppress_ids
    m.tts_model_type = "custom_voice"
    m.default_sampling_config = ns.sampling.SamplingConfig(greedy=True, repetition_penalty=1.05, repetition_window=-1)
    return m


def g3_qwen3_lm(ns):
    """Tiny Qwen3-TTS talker+depth through the reference's own worker: prefill + 3 decode frames, B=2."""
    _qwen3_lm_golden(ns, [20, 13], "g3_qwen3_lm.npz", seed=11)


def g18_qwen3_lm_b12(ns):
    """The same at 12 concurrent requests (ragged prompts of 3..31 tokens, 3 decode frames): the row counts at which the HIP
    engine and the oracle leave the fixed-order kernels for the matrix cores meet logits the REFERENCE produced."""
    _qwen3_lm_golden(ns, [9, 5, 12, 6, 31, 10, 7, 11, 13, 3, 15, 6], "g18_qwen3_lm_b12.npz", seed=18, P=64)


def g21_qwen3_full_width(ns):
    """ONE talker layer + ONE depth layer at the widths of Qwen3-TTS-1.7B (hidden 2048 / FFN 6144 / 16+8 heads of 128; depth
    1024 / 3072; 3072 + 15 x 2048 vocabularies, 16 code groups; oracle/qwen3_wide.py) through the reference's own modules and worker, at
    1, 12 and 32 concurrent requests (two decode frames each): the K = 2048 / 6144 / 1024 / 3072 reductions and every rounding
    point of a full-width layer meet numbers the REFERENCE produced — at the row counts that take the fixed-order kernels (1),
    the staged matrix-core linears (12) and the full-K matrix-core GEMMs (32)."""
    for tag, lens, seed in (("b1", [12], 211), ("b12", [3, 5, 12, 2, 4, 6, 3, 9, 2, 5, 4, 3], 212),
                            ("b32", [3, 2, 5, 4, 12, 3, 2, 6, 4, 3, 2, 5, 3, 4, 2, 7, 3, 2, 4, 3, 5, 2, 3, 4, 2, 6, 3, 2, 4, 3, 2, 5], 213)):
        _qwen3_lm_golden(ns, lens, f"g21_qwen3_full_width_{tag}.npz", seed=seed, P=max(8, len(lens) + 2), cfg=QW.wide_cfg(),
                         wseed=QW.WEIGHT_SEED, std=QW.WEIGHT_STD, n_frames=2, keep_kv=False, dl_cols=256)


def _qwen3_lm_golden(ns, prompt_lens, fname, seed, P=32, cfg=None, wseed=0, std=0.08, n_frames=3, keep_kv=True, dl_cols=None):
    torch.cuda.synchronize = lambda *a, **k: None          # worker/base.py calls it unconditionally
    FU, MW = ns.flashinfer_utils, ns.ModelWorker
    from vox_serve.model.base import PreprocessOutput
    cfg = cfg or QR.tiny_cfg()
    W = QR.random_weights(cfg, seed=wseed, std=std)
    m = _ref_qwen3(ns, cfg, W)
    t, d = cfg.talker, cfg.depth
    page = 16
    cpu = torch.device("cpu")
    w = MW.__new__(MW)
    w.model, w.device, w.page_size, w.max_batch_size = m, "cpu", page, max(4, len(prompt_lens))
    w.empty_pages = queue.Queue()
    for i in range(P):
        w.empty_pages.put(i)
    w.prefill_wrapper = FU.FlashInferPrefillWrapper(torch.empty(1), t.heads, t.kv_heads, t.heads * t.head_dim, page, device=cpu)
    w.decode_wrapper = FU.FlashInferDecodeWrapper(torch.empty(1), t.heads, t.kv_heads, t.heads * t.head_dim, page, device=cpu)
    w.kv_cache = torch.zeros(t.layers, P, 2, page, t.kv_heads, t.head_dim, dtype=torch.bfloat16)
    w.has_depth_transformer = True
    w.depth_attn_wrapper = FU.FlashInferPrefillWrapper(torch.empty(1), d.heads, d.kv_heads, d.heads * d.head_dim, page, device=cpu)
    w.depth_kv_cache = torch.zeros(d.layers, P, 2, cfg.n_groups, d.kv_heads, d.head_dim, dtype=torch.bfloat16)
    import logging
    w.logger = logging.getLogger("golden")
    w.nvtx_enabled = False

    rec = {"logits": [], "hidden": [], "dlogits": []}
    f0, df0 = m.forward, m.depth_forward

    def fwd(**kw):
        lg, hs = f0(**kw)
        rec["logits"].append(lg.clone()); rec["hidden"].append(hs.clone())
        return lg, hs

    def dfwd(**kw):
        lg = df0(**kw)
        rec["dlogits"].append(lg.clone())
        return lg
    m.forward, m.depth_forward = fwd, dfwd

    g = torch.Generator().manual_seed(seed)
    out = {"page": np.int32(page), "P": np.int32(P), "prompt_lens": np.array(prompt_lens, np.int32)}
    reqs = []
    R = ns.requests.Request
    for r, n in enumerate(prompt_lens):
        ids = torch.zeros(n, cfg.n_groups + 1, dtype=torch.long)
        ids[:, -1] = torch.randint(0, cfg.text_vocab, (n,), generator=g)
        ids[:, 0] = torch.randint(0, cfg.vocab - 1024, (n,), generator=g)
        masks = torch.zeros(n, cfg.n_groups + 1, dtype=torch.bool)
        masks[n // 2:, -1] = True
        feats = (0.05 * torch.randn(n, t.hidden, generator=g)).to(torch.bfloat16)
        feats[: n // 3] = 0
        out[f"r{r}_ids"], out[f"r{r}_masks"], out[f"r{r}_feats"] = (
            ids.numpy().astype(np.int32), masks[:, -1].numpy().astype(np.uint8), bits(feats))
        req = R(request_id=str(r), prompt="x")
        m.preprocess = (lambda ids=ids, masks=masks, feats=feats: (lambda prompt=None, audio_path=None, **kw:
                        PreprocessOutput(input_tokens=ids, input_masks=masks, input_features=feats,
                                         repetition_cache=torch.zeros(1, cfg.n_groups + 1, cfg.vocab,
                                                                      dtype=torch.bool))))()  # qwen3_tts.py:1786
        # one prefill per step, as the scheduler does (scheduler/base.py:264-298)
        li = w.prepare_lm_inputs([req], [])
        n_before = len(rec["logits"])
        w.run_lm_prefill([req], li)
        out[f"r{r}_prefill_logits"] = bits(rec["logits"][n_before][-1:, 0])
        out[f"r{r}_prefill_hidden"] = bits(rec["hidden"][n_before][-1:])
        out[f"r{r}_kv_pages"] = np.array(req.kv_pages, np.int32)
        out[f"r{r}_frame0"] = req.lm_output_tokens[-1].numpy().astype(np.int32)[0]
        out[f"r{r}_next_pos"] = np.int32(req.next_position_id)
        reqs.append(req)
    dlc = slice(None) if dl_cols is None else slice(0, dl_cols)      # (g21: the first dl_cols columns of every depth head)
    out["prefill_dlogits"] = np.stack([bits(x[1::2] if x.shape[0] == 2 else x)[..., dlc] for x in rec["dlogits"]])
    rec["dlogits"].clear()
    for f in range(n_frames):
        li = w.prepare_lm_inputs(reqs, [])
        out[f"f{f}_pos"] = li["position_ids"].numpy().astype(np.int32)
        out[f"f{f}_indptr"] = np.array(li["paged_kv_indptr"], np.int32)
        out[f"f{f}_indices"] = np.array(li["paged_kv_indices"], np.int32)
        out[f"f{f}_last"] = np.array(li["paged_kv_last_page_len"], np.int32)
        out[f"f{f}_in_ids"] = li["input_ids"].numpy().astype(np.int32)
        out[f"f{f}_in_feats"] = bits(li["input_features"])
        w.run_lm_decode(reqs, li)
        out[f"f{f}_logits"] = bits(rec["logits"][-1][:, 0])
        out[f"f{f}_hidden"] = bits(rec["hidden"][-1])
        out[f"f{f}_dlogits"] = np.stack([bits(x[1::2] if x.shape[0] == 2 * len(reqs) else x)[..., dlc] for x in rec["dlogits"]])
        rec["dlogits"].clear()
        out[f"f{f}_tokens"] = np.stack([r.lm_output_tokens[-1].numpy().astype(np.int32)[0] for r in reqs])
    if keep_kv:
        out["kv_final"] = bits(w.kv_cache)
    out["n_frames"] = np.int32(n_frames)
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, "ok; frame tokens", out[f"f{n_frames - 1}_tokens"][:, :6])



# --------------------------------------------------------------------------------------------------
def _lm_worker(ns, m, c, page, P):
    import logging
    FU, MW = ns.flashinfer_utils, ns.ModelWorker
    cpu = torch.device("cpu")
    w = MW.__new__(MW)
    w.model, w.device, w.page_size, w.max_batch_size = m, "cpu", page, 4
    w.empty_pages = queue.Queue()
    for i in range(P):
        w.empty_pages.put(i)
    w.prefill_wrapper = FU.FlashInferPrefillWrapper(torch.empty(1), c.heads, c.kv_heads, c.heads * c.head_dim, page, device=cpu)
    w.decode_wrapper = FU.FlashInferDecodeWrapper(torch.empty(1), c.heads, c.kv_heads, c.heads * c.head_dim, page, device=cpu)
    w.kv_cache = torch.zeros(c.layers, P, 2, page, c.kv_heads, c.head_dim, dtype=torch.bfloat16)
    w.has_depth_transformer = False
    w.logger = logging.getLogger("golden")
    w.nvtx_enabled = False
    return w


def _drive_lm(ns, w, m, prompts, n_steps, out, tag, rep_shape=None):
    """One prefill per request, then n_steps batched decode steps through the reference worker (greedy)."""
    from vox_serve.model.base import PreprocessOutput
    rec = []
    f0 = m.forward

    def fwd(**kw):
        lg = f0(**kw)
        rec.append(lg.clone())
        return lg
    m.forward = fwd
    reqs = []

    cl = lambda t: None if t is None else t.clone()

    def run(task):
        try:
            task.send(None)
        except StopIteration:
            pass
    for r, pp in enumerate(prompts):
        req = ns.requests.Request(request_id=str(r), prompt="x")
        rc = torch.zeros(*rep_shape, dtype=torch.bool) if rep_shape else None
        m.preprocess = (lambda pp=pp, rc=rc: (lambda prompt=None, audio_path=None, **kw: PreprocessOutput(
            input_tokens=pp["ids"], input_masks=cl(pp.get("masks")), input_features=cl(pp.get("feats")),
            repetition_cache=rc)))()      # clones: CosyVoice2Model.sampling zeroes these in place (cosyvoice2.py:1062-1064)
        li = w.prepare_lm_inputs([req], [])
        run(w.run_lm_prefill([req], li))
        out[f"{tag}_r{r}_prefill_logits"] = bits(rec[-1][-1:, 0])
        out[f"{tag}_r{r}_tok0"] = req.lm_output_tokens[-1].numpy().astype(np.int32)[0]
        out[f"{tag}_r{r}_next_pos"] = np.int32(req.next_position_id)
        reqs.append(req)
    for f in range(n_steps):
        li = w.prepare_lm_inputs(reqs, [])
        out[f"{tag}_f{f}_pos"] = li["position_ids"].numpy().astype(np.int32)
        run(w.run_lm_decode(reqs, li))
        out[f"{tag}_f{f}_logits"] = bits(rec[-1][:, 0])
        out[f"{tag}_f{f}_tokens"] = np.stack([r.lm_output_tokens[-1].numpy().astype(np.int32)[0] for r in reqs])
    out[f"{tag}_kv_final"] = bits(w.kv_cache)


def g7_single_stack_lms(ns):
    """Tiny GLM-4-Voice and CosyVoice2 LMs through the reference's modules + worker (glm_voice.py, cosyvoice2.py)."""
    torch.cuda.synchronize = lambda *a, **k: None
    from vox_serve.model import cosyvoice2 as CV
    from vox_serve.model import glm_voice as GV
    from oracle import lm_ref as LR
    out = {"page": np.int32(16), "P": np.int32(24)}
    g = torch.Generator().manual_seed(5)

    # ---- GLM-4-Voice, greedy ----
    cfg = LR.tiny_glm_cfg()
    c = cfg.stack
    S = LR.random_glm_state_dict(cfg, seed=3, std=0.08)
    gc = GV.GLMVoiceConfig(ffn_hidden_size=c.ffn, hidden_size=c.hidden, multi_query_group_num=c.kv_heads,
                           num_attention_heads=c.heads, num_hidden_layers=c.layers, num_layers=c.layers,
                           padded_vocab_size=cfg.vocab_out, vocab_size=cfg.vocab_out, layernorm_epsilon=c.eps)
    net = GV.GLMVoiceForCausalLM(gc).to(torch.bfloat16)
    missing, unexpected = net.load_state_dict({k: vr.to_torch(v) for k, v in S.items()}, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    prompts = [{"ids": torch.randint(0, cfg.vocab_in, (n, 1), generator=g)} for n in (11, 19)]
    for i, pp in enumerate(prompts):
        out[f"glm_r{i}_ids"] = pp["ids"].numpy().astype(np.int32)[:, 0]
    m = GV.GLMVoiceModel.__new__(GV.GLMVoiceModel)
    m.model, m.config, m.device, m.dtype = net, gc, "cpu", torch.bfloat16
    m._num_attention_heads, m._num_key_value_heads = c.heads, c.kv_heads
    m._num_hidden_layers, m._hidden_size = c.layers, c.hidden
    m.stop_token_ids, m.audio_offset = [cfg.vocab_out - 3, cfg.vocab_out - 2, cfg.vocab_out - 1], cfg.vocab_out // 2
    m.default_sampling_config = ns.sampling.SamplingConfig(greedy=True)
    _drive_lm(ns, _lm_worker(ns, m, c, 16, 24), m, prompts, 4, out, "glm")

    # ---- CosyVoice2: prefill rows are input_features (mask 1), decode rows are speech_embedding[id] ----
    cfg = LR.tiny_cosyvoice2_cfg()
    c = cfg.stack
    S = LR.random_cosyvoice2_state_dict(cfg, seed=4, std=0.08)
    cc = CV.CosyVoice2Config()
    cc.llm_input_size = cc.llm_output_size = cc.hidden_size = c.hidden
    cc.speech_token_size, cc.intermediate_size = cfg.vocab_out - 3, c.ffn
    cc.num_attention_heads, cc.num_key_value_heads, cc.num_hidden_layers = c.heads, c.kv_heads, c.layers
    cc.vocab_size = S["llm.model.model.embed_tokens.weight"].shape[0]
    net = CV.CosyVoice2ForCausalLM(cc).to(torch.bfloat16)
    missing, unexpected = net.load_state_dict({k: vr.to_torch(v) for k, v in S.items()}, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    m = CV.CosyVoice2Model.__new__(CV.CosyVoice2Model)
    m.model, m.config, m.device, m.dtype = net, cc, "cpu", torch.bfloat16
    m._num_attention_heads, m._num_key_value_heads = c.heads, c.kv_heads
    m._num_hidden_layers, m._hidden_size = c.layers, c.hidden
    m.stop_token_ids = [cc.speech_token_size + i for i in range(3)]
    m.default_sampling_config = ns.sampling.SamplingConfig(greedy=True)
    prompts = []
    for i, n in enumerate((9, 14)):
        ids = torch.randint(0, cc.vocab_size, (n, 1), generator=g)
        feats = (0.08 * torch.randn(n, c.hidden, generator=g)).to(torch.bfloat16)
        prompts.append({"ids": ids, "masks": torch.ones(n, 1, dtype=torch.bool), "feats": feats})
        out[f"cosy_r{i}_ids"], out[f"cosy_r{i}_feats"] = ids.numpy().astype(np.int32)[:, 0], bits(feats)
    _drive_lm(ns, _lm_worker(ns, m, c, 16, 24), m, prompts, 4, out, "cosy")
    # same prompts with a persisted per-request repetition cache (sliding window 2, penalty 2.0): the greedy run
    # above repeats a token, so the penalty changes the outcome (sampling.py:121-178, cosyvoice2.py:1046-1058)
    m.forward = type(m).forward.__get__(m)
    m.default_sampling_config = ns.sampling.SamplingConfig(greedy=True, repetition_penalty=2.0, repetition_window=2)
    _drive_lm(ns, _lm_worker(ns, m, c, 16, 24), m, prompts, 4, out, "cosyrep", (2, 1, cfg.vocab_out))
    np.savez_compressed(os.path.join(HERE, "g7_single_stack_lms.npz"), **out)
    for t in ("glm", "cosy", "cosyrep"):
        print("g7", t, [out[f"{t}_f{f}_tokens"].ravel().tolist() for f in range(4)])


def g22_glm_full_width(ns):
    """ONE GLM-4-Voice-9B layer at full width (hidden 4096, 32 q / 2 kv heads of 128, FFN 13696, QKV bias, half-rotary interleaved RoPE;
    a 4096-entry vocabulary keeps the two 168960 x 4096 tables out of the fixture's way — oracle/lm_wide.py) through the reference's
    modules and worker (glm_voice.py:85-305) at 1 and at 8 concurrent requests (BASELINE config 4's per-GPU share), two decode steps:
    the K = 4096 / 13696 reductions and every rounding point of a full-width layer meet numbers the REFERENCE produced — at the row
    counts that take the fixed-order kernels (1) and the matrix-core linears (8).  The g21 recipe, for the single-stack family."""
    torch.cuda.synchronize = lambda *a, **k: None
    from vox_serve.model import glm_voice as GV
    from oracle import lm_ref as LR, lm_wide as LW
    cfg = LW.wide_glm_cfg()
    c = cfg.stack
    S = LR.random_glm_state_dict(cfg, seed=LW.WEIGHT_SEED, std=LW.WEIGHT_STD)
    gc = GV.GLMVoiceConfig(ffn_hidden_size=c.ffn, hidden_size=c.hidden, multi_query_group_num=c.kv_heads,
                           num_attention_heads=c.heads, num_hidden_layers=c.layers, num_layers=c.layers,
                           padded_vocab_size=cfg.vocab_out, vocab_size=cfg.vocab_out, layernorm_epsilon=c.eps)
    net = GV.GLMVoiceForCausalLM(gc).to(torch.bfloat16)
    missing, unexpected = net.load_state_dict({k: vr.to_torch(v) for k, v in S.items()}, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    for tag, lens, seed in (("b1", [9], 221), ("b8", [3, 5, 2, 7, 4, 3, 6, 2], 222)):
        g = torch.Generator().manual_seed(seed)
        page, P = 16, max(8, len(lens) + 2)
        out = {"page": np.int32(page), "P": np.int32(P), "prompt_lens": np.array(lens, np.int32), "n_steps": np.int32(2)}
        prompts = [{"ids": torch.randint(0, cfg.vocab_in, (n, 1), generator=g)} for n in lens]
        for i, pp in enumerate(prompts):
            out[f"glm_r{i}_ids"] = pp["ids"].numpy().astype(np.int32)[:, 0]
        m = GV.GLMVoiceModel.__new__(GV.GLMVoiceModel)
        m.model, m.config, m.device, m.dtype = net, gc, "cpu", torch.bfloat16
        m._num_attention_heads, m._num_key_value_heads = c.heads, c.kv_heads
        m._num_hidden_layers, m._hidden_size = c.layers, c.hidden
        m.stop_token_ids, m.audio_offset = [cfg.vocab_out - 3, cfg.vocab_out - 2, cfg.vocab_out - 1], cfg.vocab_out // 2
        m.default_sampling_config = ns.sampling.SamplingConfig(greedy=True)
        w = _lm_worker(ns, m, c, page, P)
        w.max_batch_size = max(4, len(lens))
        _drive_lm(ns, w, m, prompts, 2, out, "glm")
        del out["glm_kv_final"]                     # (the logits pin the layer; the K/V bytes would double the fixture)
        np.savez_compressed(os.path.join(HERE, f"g22_glm_full_width_{tag}.npz"), **out)
        print("g22", tag, [out[f"glm_f{f}_tokens"].ravel().tolist() for f in range(2)])


def _ref_codec(ns, cfg, W, dtype):
    qc = ns.qwen3_codec
    rc = qc.Qwen3TTSTokenizerV2DecoderConfig(
        latent_dim=cfg.latent_dim, codebook_dim=cfg.codebook_dim, codebook_size=cfg.codebook_size,
        decoder_dim=cfg.decoder_dim, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
        head_dim=cfg.head_dim, num_attention_heads=cfg.num_heads, num_key_value_heads=cfg.num_heads,
        num_hidden_layers=cfg.num_layers, num_quantizers=cfg.num_quantizers, rms_norm_eps=cfg.rms_eps,
        rope_theta=int(cfg.rope_theta), sliding_window=cfg.sliding_window, upsample_rates=list(cfg.upsample_rates),
        upsampling_ratios=list(cfg.upsampling_ratios))
    m = qc.Qwen3TTSTokenizerV2Decoder(rc)
    sd = m.state_dict()
    from oracle import qwen3_codec_ref as CR
    shapes = CR.param_shapes(cfg)
    assert set(shapes) == set(sd), (set(shapes) ^ set(sd))
    for k, v in sd.items():
        assert tuple(v.shape) == shapes[k], (k, v.shape, shapes[k])
    m.load_state_dict({k: v.clone() for k, v in W.items()})
    return m.to(dtype).eval()


def g4_qwen3_codec(ns):
    """Qwen3 12 Hz codec decoder, streaming forward_chunk (qwen3_codec.py:1541-1666): the reference module
    itself, seeded synthetic weights (oracle.qwen3_codec_ref.random_codec_weights), fp32 and bf16."""
    from oracle import qwen3_codec_ref as CR
    out = {}
    for tag, cfg, B, nfr, chunks in (("tiny", CR.tiny_codec_cfg(), 2, 12, (4, 3)), ("full", CR.CodecCfg(), 2, 30, (10,))):
        W = CR.random_codec_weights(cfg, seed=0)
        g = torch.Generator().manual_seed(5)
        codes = torch.randint(0, cfg.codebook_size, (B, cfg.num_quantizers, nfr), generator=g)
        out[f"{tag}_codes"] = codes.numpy().astype(np.int32)
        for dt_name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
            m = _ref_codec(ns, cfg, W, dt)
            for ch in chunks:
                cache = m.init_cache(B, torch.device("cpu"), dt, ch)
                wavs = []
                for t0 in range(0, nfr, ch):
                    wav, cache = m.forward_chunk(codes[:, :, t0:t0 + ch], cache)
                    wavs.append(wav.float().clone())
                wav = torch.cat(wavs, -1)
                out[f"{tag}_{dt_name}_c{ch}"] = wav.numpy().astype(np.float32 if (tag == "tiny" or dt_name == "fp32") else np.float16)
                print(tag, dt_name, ch, tuple(wav.shape), "rms", float(wav.pow(2).mean().sqrt()), "max", float(wav.abs().max()))
    np.savez_compressed(os.path.join(HERE, "g4_qwen3_codec.npz"), **out)
    print("g4 ok")


class FakePlugin:
    """Deterministic stand-in for a model plugin: exercises only the HOST logic of worker + scheduler."""
    model_name = "fake"
    supports_input_streaming = False
    needs_input_masks = True
    needs_input_features = True
    use_repetition_penalty = False
    supports_audio_input = False
    needs_watermarking = False
    detokenize_interval = 4
    detokenize_overlap = 0
    n_codebooks = 3

    def __init__(self, PreprocessOutput):
        self.PO = PreprocessOutput

    def preprocess(self, prompt=None, audio_path=None, **kw):
        n = int(prompt)
        toks = torch.arange(n * 3, dtype=torch.long).view(n, 3)
        return self.PO(input_tokens=toks, input_masks=torch.ones(n, 3, dtype=torch.bool),
                       input_features=torch.zeros(n, 8))

    def postprocess(self, token_ids, **kw):
        b, t, _ = token_ids.shape            # 5 samples per token, value = f(token ids)
        base = (token_ids[:, :, 0].float() % 97) / 100.0 - 0.4
        return base.repeat_interleave(5, dim=1)[:, None, :]


def g6_host_traces(ns):
    """Scripted arrivals through the reference's own prepare_lm_inputs / _select_* / run_detokenize / free_kv_cache."""
    import hashlib
    import logging
    torch.cuda.synchronize = lambda *a, **k: None
    from vox_serve.model.base import PreprocessOutput
    from vox_serve.scheduler.base import Scheduler
    MW, R = ns.ModelWorker, ns.requests.Request
    w = MW.__new__(MW)
    w.model, w.device, w.detokenizer_device, w.page_size, w.max_batch_size = FakePlugin(PreprocessOutput), "cpu", "cpu", 4, 8
    w.empty_pages = queue.Queue()
    for i in range(12):
        w.empty_pages.put(i)
    w.needs_watermarking, w.logger, w.nvtx_enabled = False, logging.getLogger("g6"), False
    s = Scheduler.__new__(Scheduler)
    s.model_worker, s.max_batch_size, s.active_requests = w, 8, []
    arrivals = {0: [("A", 6)], 1: [("B", 3)], 4: [("C", 9)]}
    finish_after = {"A": 9, "B": 5, "C": 6}        # frames until "EOS"
    trace = []
    for step in range(24):
        for rid, n in arrivals.get(step, []):
            s.active_requests.append(R(request_id=rid, prompt=str(n)))
        s.active_requests = [r for r in s.active_requests if not r.done_all]
        detok = s._select_detokenize_requests()
        lm = s._select_lm_requests()
        li = w.prepare_lm_inputs(lm, detok)
        w.run_detokenize(detok)
        rec = {"step": step, "detok": [(r.request_id, list(r.audio_decode_idx)) for r in detok],
               "lm": [r.request_id for r in lm], "pcm": []}
        for r in detok:
            while not r.output_audio.empty():
                b = r.output_audio.get()
                rec["pcm"].append((r.request_id, len(b), hashlib.sha256(b).hexdigest()[:16]))
            if r.done_all:
                w.free_kv_cache(r)
        if li is not None:
            rec.update(qo=li["qo_indptr"], indptr=li["paged_kv_indptr"], indices=li["paged_kv_indices"],
                       last=li["paged_kv_last_page_len"], pos=li["position_ids"].tolist(), is_prefill=li["is_prefill"],
                       n_rows=int(li["input_ids"].shape[0]))
        for r in lm:      # stand-in for the LM: one frame per selected request
            k = len(r.lm_output_tokens)
            row = torch.tensor([[100 * (ord(r.request_id) - 64) + k, k, 7]], dtype=torch.long)
            r.input_tokens, r.input_masks, r.input_features = row, torch.ones(1, 3, dtype=torch.bool), torch.zeros(1, 8)
            r.lm_output_tokens.append(row)
            if k + 1 >= finish_after[r.request_id]:
                r.done_lm_generation, r.finish_reason = True, "stop_id_encountered"
            else:
                r.lm_output_audio_tokens.append(row)
        rec["pages"] = {r.request_id: list(r.kv_pages or []) for r in s.active_requests}
        rec["done"] = [r.request_id for r in s.active_requests if r.done_all]
        trace.append(rec)
        if step > 6 and not [r for r in s.active_requests if not r.done_all]:
            break
    free = []
    while not w.empty_pages.empty():
        free.append(w.empty_pages.get())
    import json
    with open(os.path.join(HERE, "g6_host_traces.json"), "w") as f:
        json.dump({"trace": trace, "free_pages_after": free}, f, indent=0)
    print("g6 ok", len(trace), "steps; pcm chunks", sum(len(t["pcm"]) for t in trace))



# --------------------------------------------------------------------------------------------------
class _FakeTok:
    """Deterministic stand-in for the HF tokenizer used by input_streaming.py: one token per 3 characters."""
    def encode(self, text, add_special_tokens=False):
        return [sum(map(ord, text[i:i + 3])) % 5000 + 10 for i in range(0, len(text), 3)]

    def decode(self, ids, skip_special_tokens=True):
        return "".join(f"<{i}>" for i in ids)


def _scenario(rng, n_req, interval):
    """Random request states: prefill pending / decoding / finished, streaming or not, various backlogs."""
    reqs = []
    for i in range(n_req):
        kind = rng.choice(["prefill", "decode", "decode", "done"])
        ntok = int(rng.integers(0, 4 * interval + 3))
        step_taken = int(rng.integers(0, 3))
        r = {"id": f"r{i}", "done_lm_prefill": kind != "prefill", "done_lm_generation": kind == "done",
             "input_length": int(rng.choice([0, 40, 300, 700, 1100])), "n_tokens": 0 if kind == "prefill" else ntok,
             "next_idx": [] if step_taken == 0 or kind == "prefill" else [int(rng.integers(0, max(1, ntok // 2)))],
             "is_streaming": bool(rng.integers(0, 2)), "is_pressing": bool(rng.integers(0, 2)),
             "is_input_streaming": bool(rng.integers(0, 3) == 0), "prefill_ready": bool(rng.integers(0, 2)),
             "text_complete": bool(rng.integers(0, 2)), "n_pending_text": int(rng.integers(0, 3)),
             "chunk_times": [float(x) for x in np.cumsum(rng.uniform(0.05, 0.3, int(rng.integers(0, 4))))],
             }
        r["chunk_durs"] = [float(rng.choice([0.8, 0.4, 1.6])) for _ in r["chunk_times"]]
        reqs.append(r)
    return reqs


def _build_requests(R, spec):
    out = []
    for s_ in spec:
        r = R(request_id=s_["id"], prompt="x")
        r.done_lm_prefill, r.done_lm_generation = s_["done_lm_prefill"], s_["done_lm_generation"]
        r.input_length = s_["input_length"] or None
        r.lm_output_audio_tokens = [None] * s_["n_tokens"]
        r.next_audio_decode_idx = list(s_["next_idx"])
        r.is_streaming, r.is_pressing = s_["is_streaming"], s_["is_pressing"]
        r.is_input_streaming, r.prefill_ready, r.text_complete = s_["is_input_streaming"], s_["prefill_ready"], s_["text_complete"]
        for t in range(s_["n_pending_text"]):
            r.pending_text_tokens.put(100 + t)
        r.chunk_send_timestamps = [1000.0 + t for t in s_["chunk_times"]]
        r.chunk_durations = list(s_["chunk_durs"])
        out.append(r)
    return out


def g8_scheduler_policies(ns):
    """Selection policies of the reference's Online / Offline / InputStreaming schedulers on scripted request states
    (scheduler/online.py:16-295, offline.py:5-136, input_streaming.py:79-322), incl. the CudaGraphWorker limits."""
    import logging
    import types
    from unittest import mock
    from vox_serve.scheduler.input_streaming import InputStreamingScheduler
    from vox_serve.scheduler.offline import OfflineScheduler
    from vox_serve.scheduler.online import OnlineScheduler
    from vox_serve.worker import CudaGraphWorker
    R = ns.requests.Request
    rng = np.random.default_rng(8)
    cases = []
    for ci in range(60):
        interval, overlap = [(10, 0), (28, 3), (25, 0), (4, 1)][ci % 4]
        maxb = int(rng.choice([2, 4, 8]))
        graph_worker = ci % 3 == 0
        spec = _scenario(rng, int(rng.integers(1, 9)), interval)
        now = 1000.0 + float(rng.uniform(0, 3))
        case = {"interval": interval, "overlap": overlap, "max_batch_size": maxb, "graph_worker": graph_worker,
                "prefill_graph_batch_size": 4, "seq_len_buckets": [256, 512], "now": now, "requests": spec, "out": {}}
        for name, cls in (("online", OnlineScheduler), ("offline", OfflineScheduler), ("input_streaming", InputStreamingScheduler)):
            if graph_worker:
                w = CudaGraphWorker.__new__(CudaGraphWorker)
                w.prefill_graph_batch_size, w.cuda_graph_seq_len_buckets = 4, [256, 512]
                w.model = types.SimpleNamespace(detokenize_interval=interval, detokenize_overlap=overlap)
            else:
                w = types.SimpleNamespace(detokenize_interval=interval, detokenize_overlap=overlap)
            sch = cls.__new__(cls)
            sch.model_worker, sch.max_batch_size, sch.detokenize_max_batch_size = w, maxb, maxb
            sch.logger = logging.getLogger("golden")
            sch.sample_rate, sch.bytes_per_sample, sch.channels = 24000, 2, 1
            res = {}
            sch.active_requests = _build_requests(R, spec)
            if name == "online":
                with mock.patch("time.time", return_value=now):
                    sch._update_pressing_status()
                res["pressing"] = [r.is_pressing for r in sch.active_requests]
            res["lm"] = [r.request_id for r in sch._select_lm_requests()]
            res["waiting_for_text"] = [bool(r.waiting_for_text) for r in sch.active_requests]
            if name != "input_streaming":
                sel = sch._select_detokenize_requests()
                res["detok"] = [[r.request_id, list(r.next_audio_decode_idx), bool(r.done_all)] for r in sel]
                res["done_all"] = [bool(r.done_all) for r in sch.active_requests]
            case["out"][name] = res
        cases.append(case)

    # input-streaming message handling with a deterministic tokenizer
    sch = InputStreamingScheduler.__new__(InputStreamingScheduler)
    sch.logger = logging.getLogger("golden")
    sch.model_worker = types.SimpleNamespace(supports_audio_input=False,
                                             model=types.SimpleNamespace(text_tokenizer=_FakeTok()))
    sch.active_requests = []
    script = [b'a|TEXT_STREAM_START|{"is_streaming": true, "model_kwargs": {"language": "english"}}', b"a|TEXT_UPDATE|Hello the",
              b"a|TEXT_UPDATE|re, this is a longer sentence.", b"b|TEXT_STREAM_START|", b"b|TEXT_COMPLETE|",
              b"a|TEXT_UPDATE| More text", b"c|TEXT_STREAM_START|{}", b"c|TEXT_UPDATE|short", b"c|TEXT_COMPLETE|",
              b"zz|TEXT_UPDATE|nobody", b"a|TEXT_COMPLETE|", b"a|TEXT_UPDATE|late"]
    log = []
    for i, msg in enumerate(script):
        if i == 5:
            sch.active_requests[0].done_lm_prefill = True        # the prefill of "a" has run by now
        req = sch._handle_request_payload(msg)
        if req is not None:
            sch.active_requests.append(req)
        log.append([{"id": r.request_id, "prompt": r.prompt, "buffer": r.input_text_buffer, "prefill_ready": r.prefill_ready,
                     "pending": list(r.pending_text_tokens.queue), "total_text_tokens": r.total_text_tokens,
                     "text_complete": r.text_complete, "done_all": r.done_all, "finish_reason": r.finish_reason,
                     "is_streaming": r.is_streaming, "model_kwargs": r.model_kwargs} for r in sch.active_requests])
    with open(os.path.join(HERE, "g8_scheduler_policies.json"), "w") as f:
        json.dump({"cases": cases, "stream_script": [m.decode() for m in script], "stream_log": log}, f, indent=0)
    print("g8 ok", len(cases), "policy cases;", sum(len(c["out"]["online"]["detok"]) for c in cases), "online detok picks")


# --------------------------------------------------------------------------------------------------
def g9_csm_lm(ns):
    """Tiny CSM (backbone + depth decoder) through the reference modules and the reference worker: prefill of two
    requests (text rows + audio-context rows), then 3 decode frames, greedy (csm.py:55-313, 637-770)."""
    from oracle import csm_ref as CR
    _csm_lm_golden(ns, CR.tiny_csm_cfg(), 7, 0.08, "g9_csm_lm.npz", [(7, 5), (10, 0)], 3, None, 16, 24, 21, 64)


def g23_csm_full_width(ns):
    """ONE backbone layer + ONE depth-decoder layer at the widths of CSM-1B (backbone 2048 / 32 q + 8 kv heads of 64 / FFN 8192, depth
    1024 / 8 + 2 heads of 128 / FFN 8192, 32 codebooks of 2051, llama-3.1 RoPE with the real 8192-token original context; a 1024-entry
    text vocabulary — oracle/csm_wide.py) through the reference's modules and worker (csm.py:235-255, 637-770) at 16 concurrent
    requests (BASELINE config 4's batch), two frames of 31 depth steps: the g21 recipe for the CSM family."""
    from oracle import csm_wide as CW
    specs = [(3, 2), (4, 0), (2, 1), (5, 0), (3, 0), (2, 2), (4, 1), (3, 0), (2, 0), (6, 0), (3, 1), (2, 0), (4, 0), (3, 2), (2, 1), (5, 0)]
    _csm_lm_golden(ns, CW.wide_csm_cfg(), CW.WEIGHT_SEED, CW.WEIGHT_STD, "g23_csm_full_width_b16.npz", specs, 2, 256, 32, 20, 231, 8192)


def _csm_lm_golden(ns, cfg, wseed, std, fname, specs, n_frames, dl_cols, page, P, seed, orig_ctx):
    import transformers
    from unittest import mock
    torch.cuda.synchronize = lambda *a, **k: None
    from vox_serve.model import csm as CS
    from vox_serve.model.base import PreprocessOutput
    from oracle import csm_ref as CR
    FU, MW = ns.flashinfer_utils, ns.ModelWorker
    b, d, C, V = cfg.backbone, cfg.depth, cfg.n_codebooks, cfg.vocab
    W = CR.random_csm_state_dict(cfg, seed=wseed, std=std)
    rs = {"factor": 32.0, "high_freq_factor": 4.0, "low_freq_factor": 1.0, "original_max_position_embeddings": orig_ctx, "rope_type": "llama3"}
    dlc = slice(None) if dl_cols is None else slice(0, dl_cols)
    dcfg = transformers.CsmDepthDecoderConfig(num_codebooks=C, backbone_hidden_size=b.hidden, vocab_size=V, hidden_size=d.hidden,
                                              intermediate_size=d.ffn, num_hidden_layers=d.layers, num_attention_heads=d.heads,
                                              num_key_value_heads=d.kv_heads, head_dim=d.head_dim, rms_norm_eps=d.eps,
                                              rope_theta=d.rope_theta, rope_scaling=dict(rs), max_position_embeddings=33)
    ccfg = transformers.CsmConfig(num_codebooks=C, vocab_size=V, text_vocab_size=cfg.text_vocab, hidden_size=b.hidden,
                                  intermediate_size=b.ffn, num_hidden_layers=b.layers, num_attention_heads=b.heads,
                                  num_key_value_heads=b.kv_heads, head_dim=b.head_dim, rms_norm_eps=b.eps, rope_theta=b.rope_theta,
                                  rope_scaling=dict(rs), max_position_embeddings=cfg.max_pos, depth_decoder_config=dcfg,
                                  tie_codebooks_embeddings=False)
    for c_ in (ccfg, dcfg):           # attributes the reference reads (csm.py:66-70,72-83)
        if getattr(c_, "rope_scaling", None) is None:
            c_.rope_scaling = dict(rs)
        if not hasattr(c_, "rope_theta"):
            c_.rope_theta = b.rope_theta
        if not hasattr(c_, "attention_bias"):
            c_.attention_bias = False
        if not hasattr(c_, "attention_dropout"):
            c_.attention_dropout = 0.0
    with mock.patch.object(transformers.AutoModel, "from_config", return_value=torch.nn.Identity()):
        net = CS.CsmForConditionalGeneration(ccfg).to(torch.bfloat16)
    missing, unexpected = net.load_state_dict({k: vr.to_torch(v) for k, v in W.items()}, strict=False)
    assert not unexpected, unexpected
    assert all("audio_tokens_offsets" in m_ or "codec_model" in m_ for m_ in missing), missing
    m = CS.CSMModel.__new__(CS.CSMModel)
    m.model, m.device, m.dtype, m.model_name = net, "cpu", torch.bfloat16, "tiny-csm"
    m._num_attention_heads, m._num_key_value_heads, m._num_hidden_layers, m._hidden_size = b.heads, b.kv_heads, b.layers, b.hidden
    m._depth_num_attention_heads, m._depth_num_key_value_heads = d.heads, d.kv_heads
    m._depth_num_hidden_layers, m._depth_hidden_size = d.layers, d.hidden
    m.stop_token_id = 0
    m.default_sampling_config = ns.sampling.SamplingConfig(greedy=True)
    cpu = torch.device("cpu")
    import logging
    w = MW.__new__(MW)
    w.model, w.device, w.page_size, w.max_batch_size = m, "cpu", page, max(4, len(specs))
    w.empty_pages = queue.Queue()
    for i in range(P):
        w.empty_pages.put(i)
    w.prefill_wrapper = FU.FlashInferPrefillWrapper(torch.empty(1), b.heads, b.kv_heads, b.heads * b.head_dim, page, device=cpu)
    w.decode_wrapper = FU.FlashInferDecodeWrapper(torch.empty(1), b.heads, b.kv_heads, b.heads * b.head_dim, page, device=cpu)
    w.kv_cache = torch.zeros(b.layers, P, 2, page, b.kv_heads, b.head_dim, dtype=torch.bfloat16)
    w.has_depth_transformer = True
    w.depth_attn_wrapper = FU.FlashInferPrefillWrapper(torch.empty(1), d.heads, d.kv_heads, d.heads * d.head_dim, page, device=cpu)
    w.depth_kv_cache = torch.zeros(d.layers, P, 2, C, d.kv_heads, d.head_dim, dtype=torch.bfloat16)
    w.logger, w.nvtx_enabled = logging.getLogger("golden"), False
    rec = {"logits": [], "hidden": [], "dlogits": []}
    f0, df0 = m.forward, m.depth_forward

    def fwd(**kw):
        lg, hs = f0(**kw)
        rec["logits"].append(lg.clone()); rec["hidden"].append(hs.clone())
        return lg, hs

    def dfwd(**kw):
        lg = df0(**kw)
        rec["dlogits"].append(lg.clone())
        return lg
    m.forward, m.depth_forward = fwd, dfwd
    g = torch.Generator().manual_seed(seed)
    out = {"page": np.int32(page), "P": np.int32(P)}
    if dl_cols is not None:                     # (the wide fixtures carry their shape; g9 keeps its round-1 keys, byte for byte)
        out["n_req"], out["n_frames"] = np.int32(len(specs)), np.int32(n_frames)
    reqs = []
    for r, (nt, na) in enumerate(specs):      # text rows, then audio-context rows (csm.py:473-509)
        n = nt + na
        ids = torch.zeros(n, C + 1, dtype=torch.long)
        masks = torch.zeros(n, C + 1, dtype=torch.bool)
        ids[:nt, -1] = torch.randint(0, cfg.text_vocab, (nt,), generator=g)
        masks[:nt, -1] = True
        if na:
            ids[nt:, :C] = torch.randint(1, V, (na, C), generator=g)
            masks[nt:, :C] = True
        out[f"r{r}_ids"], out[f"r{r}_masks"] = ids.numpy().astype(np.int32), masks.numpy().astype(np.uint8)
        req = ns.requests.Request(request_id=str(r), prompt="x")
        m.preprocess = (lambda ids=ids, masks=masks: (lambda prompt=None, audio_path=None, **kw:
                        PreprocessOutput(input_tokens=ids, input_masks=masks)))()
        li = w.prepare_lm_inputs([req], [])
        n0 = len(rec["logits"])
        w.run_lm_prefill([req], li)
        out[f"r{r}_prefill_logits"] = bits(rec["logits"][n0][-1:, 0])
        out[f"r{r}_prefill_hidden"] = bits(rec["hidden"][n0][-1:])
        out[f"r{r}_frame0"] = req.lm_output_tokens[-1].numpy().astype(np.int32)[0]
        out[f"r{r}_next_pos"] = np.int32(req.next_position_id)
        out[f"r{r}_prefill_dlogits"] = np.stack([bits(x[1::2] if x.shape[0] == 2 else x)[..., dlc] for x in rec["dlogits"]])
        rec["dlogits"].clear()
        reqs.append(req)
    for f in range(n_frames):
        li = w.prepare_lm_inputs(reqs, [])
        out[f"f{f}_pos"] = li["position_ids"].numpy().astype(np.int32)
        out[f"f{f}_in_ids"] = li["input_ids"].numpy().astype(np.int32)
        out[f"f{f}_in_masks"] = li["input_masks"].numpy().astype(np.uint8)
        w.run_lm_decode(reqs, li)
        out[f"f{f}_logits"] = bits(rec["logits"][-1][:, 0])
        out[f"f{f}_hidden"] = bits(rec["hidden"][-1])
        out[f"f{f}_dlogits"] = np.stack([bits(x[1::2] if x.shape[0] == 2 * len(reqs) else x)[..., dlc] for x in rec["dlogits"]])
        rec["dlogits"].clear()
        out[f"f{f}_tokens"] = np.stack([r_.lm_output_tokens[-1].numpy().astype(np.int32)[0] for r_ in reqs])
    if dl_cols is None:
        out["kv_final"] = bits(w.kv_cache)
    np.savez_compressed(os.path.join(HERE, fname), **out)
    print(fname, "ok; frame tokens", out[f"f{n_frames - 1}_tokens"][:, :6])


# --------------------------------------------------------------------------------------------------
def g5_mimi(ns):
    """Mimi decode [2, n_q, 10] through the reference MimiModel (stateless, as CSMModel.postprocess calls it:
    csm.py:772-787, mimi.py:2993-3022), tiny and full-size configuration, fp32."""
    from oracle import mimi_ref as MR
    M = ns.mimi
    out = {}
    for tag, cfg in (("tiny", MR.tiny_mimi_cfg()), ("full", MR.MimiCfg())):
        W = MR.random_mimi_weights(cfg, seed=1)
        sea = dict(M._seanet_kwargs)
        sea.update(dimension=cfg.dim, n_filters=cfg.n_filters, ratios=list(cfg.ratios))
        tr = dict(M._transformer_kwargs)
        tr.update(d_model=cfg.dim, num_heads=cfg.num_heads, num_layers=cfg.num_layers, dim_feedforward=cfg.ffn,
                  input_dimension=cfg.dim, output_dimensions=[cfg.dim])
        enc, dec = M.SEANetEncoder(**sea), M.SEANetDecoder(**sea)
        model = M.MimiModel(enc, dec, M.SplitResidualVectorQuantizer(dimension=cfg.vq_dim, n_q=cfg.n_q, bins=cfg.bins,
                                                                    input_dimension=cfg.dim, output_dimension=cfg.dim),
                            channels=1, sample_rate=24000, frame_rate=24000 / enc.hop_length / 2,
                            encoder_frame_rate=24000 / enc.hop_length, causal=True, resample_method="conv",
                            encoder_transformer=None, decoder_transformer=M.ProjectedTransformer(device="cpu", **tr)).eval()
        missing, unexpected = model.load_state_dict(W, strict=False)
        assert not unexpected, unexpected
        for mod in model.modules():
            if hasattr(mod, "_initialized"):
                mod._initialized.fill_(1)
        model.set_num_codebooks(cfg.n_q)
        g = torch.Generator().manual_seed(17)
        codes = torch.randint(0, cfg.bins, (2, cfg.n_q, 10), generator=g)
        codes[1, :, 7:] = codes[1, :, 6:7]                       # a run of repeated frames (the worker's tail padding)
        wav = model.decode(codes)
        out[f"{tag}_codes"] = codes.numpy().astype(np.int16)
        out[f"{tag}_wav"] = wav.numpy().astype(np.float32 if (tag == "tiny" or dt_name == "fp32") else np.float16)
        print("g5", tag, tuple(wav.shape), "rms", float(wav.pow(2).mean().sqrt()))
    np.savez_compressed(os.path.join(HERE, "g5_mimi.npz"), **out)

# --------------------------------------------------------------------------------------------------
def g10_snac(ns):
    """SNAC decode through the reference SNAC module (tokenizer/snac.py:438-441), tiny and snac_24khz size, fp32, with
    NoiseBlock's torch.randn replaced by the seeded Philox stream of oracle/snac_ref.py (the noise contract); plus
    OrpheusModel.postprocess (model/orpheus.py:479-507) on LM token ids."""
    import importlib
    from oracle import snac_ref as SR
    S = importlib.import_module("vox_serve.tokenizer.snac")
    out = {}
    for tag, cfg, kw in (("tiny", SR.tiny_snac_cfg(), dict(encoder_dim=4, encoder_rates=[2, 2, 2, 2])),
                         ("full", SR.SnacCfg(), dict(encoder_dim=48, encoder_rates=[2, 4, 8, 8]))):
        W = SR.random_snac_weights(cfg, seed=1)
        m = S.SNAC(sampling_rate=24000, latent_dim=cfg.latent_dim, decoder_dim=cfg.decoder_dim, decoder_rates=list(cfg.rates),
                   attn_window_size=None, codebook_size=cfg.codebook_size, codebook_dim=cfg.codebook_dim,
                   vq_strides=list(cfg.vq_strides), noise=True, depthwise=True, **kw).eval()
        missing, unexpected = m.load_state_dict(W, strict=False)
        assert not unexpected and all(k.startswith("encoder.") or "in_proj" in k for k in missing), (missing, unexpected)
        B, T = 2, 16
        g = torch.Generator().manual_seed(3)
        codes = [torch.randint(0, cfg.codebook_size, (B, T // s), generator=g) for s in cfg.vq_strides]
        noise = SR.make_noise(cfg, B, T, seed=77)
        it = iter(noise)
        real_randn = torch.randn

        def fake_randn(shape, **kwargs):
            n = next(it)
            assert tuple(shape) == tuple(n.shape), (shape, n.shape)
            return n
        torch.randn = fake_randn
        try:
            wav = m.decode(codes)
        finally:
            torch.randn = real_randn
        for i, c in enumerate(codes):
            out[f"{tag}_codes{i}"] = c.numpy().astype(np.int16)
        out[f"{tag}_wav"] = wav.numpy().astype(np.float32)
        print("g10", tag, tuple(wav.shape), "rms", float(wav.pow(2).mean().sqrt()), "max", float(wav.abs().max()))
        if tag == "full":
            # OrpheusModel.postprocess on raw LM ids (7 per frame, 4 frames), unbound: the constructor downloads weights
            OM = importlib.import_module("vox_serve.model.orpheus").OrpheusModel
            stub = types.SimpleNamespace(audio_decoder=m, idx_14=torch.tensor([1, 4]), idx_2356=torch.tensor([2, 3, 5, 6]))
            stub._turn_token_into_id = lambda ids: OM._turn_token_into_id(stub, ids)
            tok = 128256 + 10 + torch.randint(0, 7 * 4096, (B, 28), generator=g)
            it = iter(noise)
            torch.randn = fake_randn
            try:
                audio = OM.postprocess(stub, tok)
            finally:
                torch.randn = real_randn
            out["orpheus_tokens"] = tok.numpy().astype(np.int32)
            out["orpheus_audio"] = audio.numpy().astype(np.float32)
            print("g10 orpheus", tuple(audio.shape), "rms", float(audio.pow(2).mean().sqrt()))
    out["noise_seed"] = np.int64(77)
    np.savez_compressed(os.path.join(HERE, "g10_snac.npz"), **out)


def snac_variant_cfgs():
    """Tiny SNAC decoders of the variants the 32 / 44 kHz checkpoints use: dense k7 convs and / or LocalMHA (window 4 here)."""
    import dataclasses
    from oracle import snac_ref as SR
    base = SR.tiny_snac_cfg()
    return {"dense_attn": dataclasses.replace(base, depthwise=False, attn_window_size=4),
            "dw_attn": dataclasses.replace(base, depthwise=True, attn_window_size=4),
            "dense": dataclasses.replace(base, depthwise=False, attn_window_size=None)}


def g20_snac_variants(ns):
    """SNAC.decode of the reference module for the non-depthwise / local-attention variants (tokenizer/snac.py:20-90, 119-176), tiny
    size, fp32, seeded NoiseBlock noise as in g10."""
    import importlib
    from oracle import snac_ref as SR
    S = importlib.import_module("vox_serve.tokenizer.snac")
    out = {}
    for tag, cfg in snac_variant_cfgs().items():
        W = SR.random_snac_weights(cfg, seed=2, final_gain=0.3)      # (dense convs / the attention residual: larger activations)
        m = S.SNAC(sampling_rate=24000, encoder_dim=4, encoder_rates=[2, 2, 2, 2], latent_dim=cfg.latent_dim, decoder_dim=cfg.decoder_dim,
                   decoder_rates=list(cfg.rates), attn_window_size=cfg.attn_window_size, codebook_size=cfg.codebook_size,
                   codebook_dim=cfg.codebook_dim, vq_strides=list(cfg.vq_strides), noise=True, depthwise=cfg.depthwise).eval()
        missing, unexpected = m.load_state_dict(W, strict=False)
        assert not unexpected and all(k.startswith("encoder.") or "in_proj" in k or "inv_freq" in k for k in missing), (missing, unexpected)
        B, T = 2, 16
        g = torch.Generator().manual_seed(5)
        codes = [torch.randint(0, cfg.codebook_size, (B, T // s), generator=g) for s in cfg.vq_strides]
        noise = SR.make_noise(cfg, B, T, seed=78)
        it = iter(noise)
        real_randn = torch.randn

        def fake_randn(shape, **kwargs):
            n = next(it)
            assert tuple(shape) == tuple(n.shape), (shape, n.shape)
            return n
        torch.randn = fake_randn
        try:
            with torch.no_grad():
                wav = m.decode(codes)
        finally:
            torch.randn = real_randn
        for i, c in enumerate(codes):
            out[f"{tag}_codes{i}"] = c.numpy().astype(np.int16)
        out[f"{tag}_wav"] = wav.numpy().astype(np.float32)
        print("g20", tag, tuple(wav.shape), "rms", float(wav.pow(2).mean().sqrt()), "max", float(wav.abs().max()))
    out["noise_seed"] = np.int64(78)
    np.savez_compressed(os.path.join(HERE, "g20_snac_variants.npz"), **out)


def g11_hift(ns):
    """HiFT vocoder through the reference HiFTGenerator.forward_chunk (tokenizer/hifigan.py:641-665), tiny and CosyVoice2 size,
    fp32, with SineGen2's torch.rand / torch.randn_like replaced by the seeded streams of oracle/hift_ref.py (the noise contract);
    plus cosyvoice2.fade_in_out."""
    import importlib
    from oracle import hift_ref as HR
    Hm = importlib.import_module("vox_serve.tokenizer.hifigan")
    out = {}
    for tag, cfg, B, T in (("tiny", HR.tiny_hift_cfg(), 2, 6), ("full", HR.HiftCfg(), 2, 12)):
        W = HR.random_hift_weights(cfg, seed=2)
        m = Hm.HiFTGenerator(in_channels=cfg.in_channels, base_channels=cfg.base_channels, nb_harmonics=cfg.nb_harmonics,
                             sampling_rate=cfg.sampling_rate, nsf_alpha=cfg.nsf_alpha, nsf_sigma=cfg.nsf_sigma,
                             nsf_voiced_threshold=cfg.voiced_threshold, upsample_rates=list(cfg.upsample_rates),
                             upsample_kernel_sizes=list(cfg.upsample_kernel_sizes),
                             istft_params={"n_fft": cfg.n_fft, "hop_len": cfg.hop_len},
                             resblock_kernel_sizes=list(cfg.resblock_kernel_sizes),
                             resblock_dilation_sizes=[list(cfg.resblock_dilations)] * len(cfg.resblock_kernel_sizes),
                             source_resblock_kernel_sizes=list(cfg.source_resblock_kernel_sizes),
                             source_resblock_dilation_sizes=[list(cfg.resblock_dilations)] * len(cfg.source_resblock_kernel_sizes),
                             lrelu_slope=cfg.lrelu_slope, audio_limit=cfg.audio_limit,
                             f0_predictor=Hm.ConvRNNF0Predictor(in_channels=cfg.in_channels, cond_channels=cfg.f0_channels),
                             device=torch.device("cpu")).eval()
        m.load_state_dict(W, strict=True)
        g = torch.Generator().manual_seed(11)
        mel = (0.8 * torch.randn(B, cfg.in_channels, T, generator=g)).to(torch.bfloat16).float()
        ini, nz = HR.make_noise(cfg, B, T, seed=91)
        real_rand, real_randn_like = torch.rand, torch.randn_like
        calls = {"rand": 0, "randn_like": 0}

        def fake_rand(*shape, **kw):
            calls["rand"] += 1
            assert tuple(shape) == tuple(ini.shape), (shape, ini.shape)
            return ini.clone()

        def fake_randn_like(t, **kw):
            calls["randn_like"] += 1
            if tuple(t.shape) == tuple(nz.shape):
                return nz.clone()
            return torch.zeros_like(t)                      # SourceModuleHnNSF2's second draw: its `noise` output is not used
        torch.rand, torch.randn_like = fake_rand, fake_randn_like
        try:
            wav, src = m.forward_chunk(mel)
        finally:
            torch.rand, torch.randn_like = real_rand, real_randn_like
        assert calls == {"rand": 1, "randn_like": 2}, calls
        f0 = m.f0_predictor(mel)
        out[f"{tag}_mel"], out[f"{tag}_wav"], out[f"{tag}_source"] = mel.numpy(), wav.numpy().astype(np.float32), src.numpy().astype(np.float32)
        out[f"{tag}_f0"] = f0.detach().numpy().astype(np.float32)
        print("g11", tag, tuple(wav.shape), "rms", float(wav.pow(2).mean().sqrt()), "max", float(wav.abs().max()),
              "voiced frac", float((f0 > cfg.voiced_threshold).float().mean()), "f0 mean", float(f0.mean()))
    C2 = importlib.import_module("vox_serve.tokenizer.cosyvoice2")
    g = torch.Generator().manual_seed(5)
    a, b = torch.randn(2, 400, generator=g), torch.randn(2, 120, generator=g)
    win = torch.from_numpy(np.hamming(2 * 96)).float()
    out["fade_new"], out["fade_old"], out["fade_out"] = a.numpy(), b.numpy(), C2.fade_in_out(a, b, win).numpy()
    out["noise_seed"] = np.int64(91)
    np.savez_compressed(os.path.join(HERE, "g11_hift.npz"), **out)


def _ref_cosyvoice2_decoder(ns, fc, hc, Wf, Wh):
    """The reference CosyVoice2Decoder's modules (flow + hift) with the given weights, fp32, on a stub object: its constructor
    downloads checkpoints, its methods only need these attributes (tokenizer/cosyvoice2.py:774-860)."""
    import importlib
    Fm = importlib.import_module("vox_serve.tokenizer.cosyvoice_flow")
    Hm = importlib.import_module("vox_serve.tokenizer.hifigan")
    C2 = importlib.import_module("vox_serve.tokenizer.cosyvoice2")
    enc = Fm.UpsampleConformerEncoder(output_size=fc.dim, attention_heads=fc.enc_heads, linear_units=fc.enc_ffn, num_blocks=fc.enc_layers,
                                      input_size=fc.dim)
    est = Fm.CausalConditionalDecoder(in_channels=fc.est_in, out_channels=fc.mel, channels=[fc.est_ch], dropout=0.0,
                                      attention_head_dim=fc.est_head_dim, n_blocks=fc.est_blocks, num_mid_blocks=fc.est_mid,
                                      num_heads=fc.est_heads, act_fn="gelu")
    cfm = Fm.CausalConditionalCFM(in_channels=3 * fc.mel, spk_emb_dim=fc.mel, estimator=est)
    flow = Fm.CausalMaskedDiffWithXvec(input_size=fc.dim, output_size=fc.mel, spk_embed_dim=fc.spk_dim, vocab_size=fc.vocab,
                                       encoder=enc, decoder=cfm)
    flow.load_state_dict(Wf, strict=True)
    flow.eval()
    hift = Hm.HiFTGenerator(in_channels=hc.in_channels, base_channels=hc.base_channels, sampling_rate=hc.sampling_rate,
                            upsample_rates=list(hc.upsample_rates), upsample_kernel_sizes=list(hc.upsample_kernel_sizes),
                            source_resblock_kernel_sizes=list(hc.source_resblock_kernel_sizes),
                            source_resblock_dilation_sizes=[list(hc.resblock_dilations)] * 3,
                            f0_predictor=Hm.ConvRNNF0Predictor(in_channels=hc.in_channels, cond_channels=hc.f0_channels),
                            device=torch.device("cpu")).eval()
    hift.load_state_dict(Wh, strict=True)
    stub = types.SimpleNamespace(flow=flow, hift=hift, device=torch.device("cpu"), shared_prompt_cache_mode=True, mel_cache_len=6,
                                 source_cache_len=6 * 480, speech_window=torch.from_numpy(np.hamming(2 * 6 * 480)),
                                 MAX_CACHE_LEN=128, PREFIX_LEN=16)
    return stub, C2, flow


def g12_flow(ns):
    """CosyVoice2 detokenizer through the reference: CosyVoice2Decoder.init_cache on a synthetic prompt, then decode_chunk in the
    (default) shared-prompt mode — flow.forward_chunk (conformer encoder + 10-step CFM with classifier-free guidance, static prompt
    caches) -> HiFT -> fade / trim — fp32, tiny and CosyVoice2 size, with every torch.randn / rand / randn_like replaced by the seeded
    streams of oracle/flow_ref.py / oracle/hift_ref.py (the noise contract)."""
    from oracle import flow_ref as FR, hift_ref as HR
    import contextlib
    out = {}
    for tag, fc, hc, Np in (("tiny", FR.tiny_flow_cfg(), HR.HiftCfg(base_channels=256, f0_channels=64), 9),
                            ("full", FR.FlowCfg(), HR.HiftCfg(), 40)):
        Wf, Wh = FR.random_flow_weights(fc, seed=3), HR.random_hift_weights(hc, seed=2)
        stub, C2, flow = _ref_cosyvoice2_decoder(ns, fc, hc, Wf, Wh)
        g = torch.Generator().manual_seed(21)
        B, T = 2, 28
        ptok = torch.randint(0, fc.vocab, (1, Np), generator=g)
        pfeat = (0.7 * torch.randn(1, 2 * Np, fc.mel, generator=g)).to(torch.bfloat16).float()
        spk = torch.randn(1, fc.spk_dim, generator=g).to(torch.bfloat16).float()
        tok = torch.randint(0, fc.vocab, (B, T), generator=g)
        seed = 33
        ini, nz = HR.make_noise(hc, B, 2 * T, seed=seed, first_stream=16)
        queue = []
        real = (torch.randn, torch.rand, torch.randn_like)

        def fake_randn(*shape, **kw):
            shape = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
            z = queue.pop(0)
            assert tuple(z.shape) == shape, (z.shape, shape)
            return z.clone()

        def fake_rand(*shape, **kw):
            return ini.clone()

        def fake_randn_like(t, **kw):
            return nz.clone() if tuple(t.shape) == tuple(nz.shape) else torch.zeros_like(t)
        ref_dict = {"prompt_speech_token": ptok, "prompt_speech_token_len": Np, "prompt_feat": pfeat, "prompt_feat_len": 2 * Np,
                    "embedding": spk}
        torch.randn, torch.rand, torch.randn_like = fake_randn, fake_rand, fake_randn_like
        try:
            with contextlib.redirect_stdout(open(os.devnull, "w")), torch.no_grad():
                queue.append(FR.cfm_noise(seed, 0, fc.mel, 2 * (Np + 3)))
                cache = C2.CosyVoice2Decoder.init_cache(stub, ref_dict)
                # the mels of the chunk (decode_chunk does not return them): the same call it makes
                queue.append(FR.cfm_noise(seed, 1, fc.mel, 2 * T))
                ec, dc = cache.flow_encoder_cache, cache.flow_decoder_cache
                mels, _, _ = flow.forward_chunk(token=tok, token_len=torch.full((B,), T), prompt_feat=torch.zeros(1, 0, fc.mel), prompt_feat_len=0,
                                                embedding=spk, encoder_cache=type(ec)(conformer_att_cache=ec.conformer_att_cache.expand(B, -1, -1, -1, -1),
                                                                                      up_conformer_att_cache=ec.up_conformer_att_cache.expand(B, -1, -1, -1, -1)),
                                                decoder_cache=type(dc)(cnn_cache=[[k.expand(B, -1, -1, -1) for k in st] for st in dc.cnn_cache],
                                                                       att_cache=dc.att_cache.expand(B, -1, -1, -1, -1, -1, -1)),
                                                last_chunk=False, return_cache=False)
                queue.append(FR.cfm_noise(seed, 1, fc.mel, 2 * T))
                audio, _ = C2.CosyVoice2Decoder.decode_chunk(stub, tok, T, cache, ref_dict=ref_dict)
        finally:
            torch.randn, torch.rand, torch.randn_like = real
        assert not queue
        out[f"{tag}_prompt_token"], out[f"{tag}_prompt_feat"], out[f"{tag}_spk"] = ptok.numpy().astype(np.int32), pfeat.numpy(), spk.numpy()
        out[f"{tag}_token"] = tok.numpy().astype(np.int32)
        out[f"{tag}_mel"], out[f"{tag}_audio"] = mels.numpy().astype(np.float32), audio.float().numpy().astype(np.float32)
        out[f"{tag}_cache_lens"] = np.array([ec.conformer_att_cache.shape[3], ec.up_conformer_att_cache.shape[3], dc.att_cache.shape[5]], np.int32)
        out[f"{tag}_att_cache_sum"] = np.float64(dc.att_cache.double().sum().item())
        out[f"{tag}_up_cache_last"] = ec.up_conformer_att_cache[0, -1, 0, -1].numpy().astype(np.float32)
        print("g12", tag, "mel", tuple(mels.shape), "rms", float(mels.pow(2).mean().sqrt()), "audio", tuple(audio.shape), "rms",
              float(audio.float().pow(2).mean().sqrt()), "cache lens", out[f"{tag}_cache_lens"])
    out["noise_seed"] = np.int64(33)
    np.savez_compressed(os.path.join(HERE, "g12_flow.npz"), **out)


def g13_glm_decoder(ns):
    """GLM-4-Voice detokenizer through the reference modules (tokenizer/glm.py): GLMFlowModel.inference (block conformer encoder, length
    regulator, 10-step CFM with a non-causal U-Net) and GLMHiFTModel (two x8 stages, SineGen v1), as GLMAudioDecoder.forward chains them
    (:2640-2651), fp32, tiny and GLM-4-Voice size, with the random draws replaced by the seeded streams (oracle/glm_dec_ref.py)."""
    import importlib
    from oracle import glm_dec_ref as GR, hift_ref as HR
    Gm = importlib.import_module("vox_serve.tokenizer.glm")
    out = {}
    for tag, fc, hc in (("tiny", GR.tiny_glm_flow_cfg(), GR.glm_hift_cfg(base_channels=128, f0_channels=64)), ("full", GR.GlmFlowCfg(), GR.glm_hift_cfg())):
        Wf, Wh = GR.random_glm_flow_weights(fc, seed=5), HR.random_hift_weights(hc, seed=6)
        flow = Gm.GLMFlowModel(vocab_size=fc.vocab,
                               encoder=Gm.BlockConformerEncoder(attention_heads=fc.enc_heads, linear_units=fc.enc_ffn, num_blocks=fc.enc_layers,
                                                                block_size=fc.block_size),
                               length_regulator=Gm.InterpolateRegulator(),
                               decoder=Gm.ConditionalCFM(estimator=Gm.ConditionalDecoder(channels=(fc.est_ch, fc.est_ch), attention_head_dim=fc.est_head_dim,
                                                                                         n_blocks=fc.est_blocks, num_mid_blocks=fc.est_mid,
                                                                                         num_heads=fc.est_heads))).eval()
        flow.load_state_dict(Wf, strict=True)
        hift = Gm.GLMHiFTModel(base_channels=hc.base_channels,
                               f0_predictor=Gm.ConvRNNF0Predictor(cond_channels=hc.f0_channels)).eval()
        Wh_old = {k.replace(".parametrizations.weight.original0", ".weight_g").replace(".parametrizations.weight.original1", ".weight_v"): v
                  for k, v in Wh.items()}
        hift.load_state_dict(Wh_old, strict=True)
        g = torch.Generator().manual_seed(8)
        B, T = 2, 25
        tok = torch.randint(0, fc.vocab, (B, T), generator=g)
        Tm = fc.mel_len(T)
        seed = 47
        z = GR.glm_cfm_noise(seed, 0, B, fc.mel, Tm)
        ini, nz = HR.make_noise(hc, B, Tm, seed=seed, first_stream=8)
        real_randn_like = torch.randn_like

        def fake_randn_like(t, **kw):
            if tuple(t.shape) == tuple(z.shape):
                return z.clone()
            if tuple(t.shape) == (B, hc.nb_harmonics + 1, Tm * hc.upsample_scale):
                return nz.transpose(1, 2).clone()
            return torch.zeros_like(t)

        class FakeU:
            def sample(self, sample_shape):
                assert tuple(sample_shape) == (B, hc.nb_harmonics + 1, 1), sample_shape
                return (-np.pi + 2 * np.pi * ini).unsqueeze(-1).float()
        hift.m_source.l_sin_gen._u_dist = FakeU()
        torch.randn_like = fake_randn_like
        try:
            with torch.no_grad():
                mel = flow.inference(token=tok, token_len=torch.tensor([T], dtype=torch.int32), embedding=torch.zeros(B, 192))
                wav, src = hift.inference(mel=mel)
        finally:
            torch.randn_like = real_randn_like
        out[f"{tag}_token"], out[f"{tag}_mel"] = tok.numpy().astype(np.int32), mel.numpy().astype(np.float32)
        out[f"{tag}_wav"], out[f"{tag}_source"] = wav.numpy().astype(np.float32), src.numpy().astype(np.float32)
        print("g13", tag, "mel", tuple(mel.shape), "rms", float(mel.pow(2).mean().sqrt()), "wav", tuple(wav.shape), "rms", float(wav.pow(2).mean().sqrt()))
    out["noise_seed"] = np.int64(47)
    np.savez_compressed(os.path.join(HERE, "g13_glm_decoder.npz"), **out)


class _TplTok:
    """Stand-in for the Qwen text tokenizer inside Qwen3TTSModel.preprocess: the chat-template pieces map to their
    real ids, every other run of text to one token per 3 characters (ids 1000..5999)."""
    SPECIAL = {"<|im_start|>": 151644, "<|im_end|>": 151645, "\n": 198, "assistant": 77091, "user": 872}

    def ids(self, text):
        out, i = [], 0
        while i < len(text):
            for k, v in self.SPECIAL.items():
                if text.startswith(k, i):
                    out.append(v)
                    i += len(k)
                    break
            else:
                j = i
                while j < len(text) and j - i < 3 and not any(text.startswith(k, j) for k in self.SPECIAL):
                    j += 1
                out.append(1000 + sum(map(ord, text[i:j])) % 5000)
                i = j
        return out

    def encode(self, text, return_tensors=None):
        return torch.tensor([self.ids(text)], dtype=torch.long)


def g14_qwen3_preprocess(ns):
    """Qwen3TTSModel.preprocess of the reference (qwen3_tts.py:1373-1803) in every mode: custom voice (language id,
    dialect speaker, instruct), voice design, x-vector-only cloning, ICL cloning, and the input-streaming variants.
    The two prompt-side encoders are replaced by given outputs (speaker embedding / reference codes) here; they have
    their own fixtures (g15, g16)."""
    import logging
    cfg = QR.tiny_cfg()
    W = QR.random_weights(cfg, seed=0, std=0.08)
    m = _ref_qwen3(ns, cfg, W)
    tc = m.config.talker_config
    # special ids inside the tiny vocabulary (1280 codec ids)
    tc.codec_pad_id, tc.codec_bos_id, tc.codec_think_id, tc.codec_nothink_id = 1148, 1149, 1154, 1155
    tc.codec_think_bos_id, tc.codec_think_eos_id = 1156, 1157
    m.config.tts_bos_token_id, m.config.tts_eos_token_id = 151672, 151673
    tc.spk_id = {"vivian": 1101, "dylan": 1102}
    tc.spk_is_dialect = {"vivian": False, "dylan": "beijing_dialect"}
    tc.codec_language_id = {"chinese": 1055, "english": 1050, "german": 1053, "beijing_dialect": 1074}
    out_ids = dict(tts_bos=151672, tts_eos=151673, tts_pad=int(m.config.tts_pad_token_id), codec_pad=1148, codec_bos=1149,
                   codec_think=1154, codec_nothink=1155, codec_think_bos=1156, codec_think_eos=1157)
    tok = _TplTok()
    m.text_tokenizer, m.logger, m.tts_model_size, m.speaker_encoder_sample_rate = tok, logging.getLogger("g14"), "1b7", 24000
    m.audio_decoder_initial_cache = lambda batch_size: None
    H_ = cfg.talker.hidden
    g = torch.Generator().manual_seed(14)
    spk = (torch.randn(H_, generator=g) * 0.5).to(torch.bfloat16)
    ref_codes = torch.randint(0, cfg.depth_vocab, (7, cfg.n_groups), generator=g)
    m._load_audio_to_np = lambda x: (np.zeros(2400, np.float32), 24000)
    m._extract_speaker_embedding = lambda audio, sr: spk
    m._encode_audio_to_codes = lambda audio, sr: ref_codes
    out = {"spk_embedding": bits(spk), "ref_codes": ref_codes.numpy().astype(np.int32),
           "special_ids": np.array(json.dumps(dict(out_ids, spk_id=tc.spk_id, spk_is_dialect=tc.spk_is_dialect,
                                                   codec_language_id=tc.codec_language_id)))}
    cases = [
        ("cv_english", "custom_voice", dict(prompt="Hello there, world.", language="english", speaker="Vivian")),
        ("cv_auto_dialect", "custom_voice", dict(prompt="ni hao", language="auto", speaker="dylan")),
        ("cv_instruct", "custom_voice", dict(prompt="Read this slowly.", language="german", speaker="vivian", instruct="Speak calmly")),
        ("cv_stream", "custom_voice", dict(prompt="Streaming text", language="english", speaker="vivian", is_input_streaming=True)),
        ("vd_instruct", "voice_design", dict(prompt="A designed voice.", language="auto", instruct="A deep male voice")),
        ("vd_stream", "voice_design", dict(prompt="Design stream", language="english", instruct="bright", is_input_streaming=True)),
        ("xvec", "base", dict(prompt="Clone by x-vector.", audio_path="ref.wav", ref_text="ignored", language="english", x_vector_only_mode=True)),
        ("xvec_stream", "base", dict(prompt="Clone stream", audio_path="ref.wav", ref_text="r", language="auto", x_vector_only_mode=True, is_input_streaming=True)),
        ("icl", "base", dict(prompt="Say this in the cloned voice.", audio_path="ref.wav", ref_text="The reference transcript.", language="english")),
        ("icl_auto_instruct", "base", dict(prompt="Second one", audio_path="ref.wav", ref_text="Ref", language="auto", instruct="whisper")),
    ]
    for tag, kind, kw in cases:
        m.tts_model_type = kind
        m.config.tts_model_type = kind
        po = m.preprocess(**kw)
        out[f"{tag}_tokens"] = po.input_tokens.numpy().astype(np.int64)
        out[f"{tag}_masks"] = po.input_masks.numpy()
        out[f"{tag}_features"] = bits(po.input_features)
        tpl = "<|im_start|>assistant\n{prompt}" if kw.get("is_input_streaming") else "<|im_start|>assistant\n{prompt}<|im_end|>\n<|im_start|>assistant\n"
        out[f"{tag}_prompt_ids"] = np.array(tok.ids(tpl.format(prompt=kw["prompt"])), np.int64)
        if kw.get("instruct"):
            out[f"{tag}_instruct_ids"] = np.array(tok.ids("<|im_start|>user\n{}<|im_end|>\n".format(kw["instruct"])), np.int64)
        if kw.get("ref_text") and not kw.get("x_vector_only_mode"):
            out[f"{tag}_ref_text_ids"] = np.array(tok.ids("<|im_start|>assistant\n{}<|im_end|>\n".format(kw["ref_text"])), np.int64)
        print("g14", tag, tuple(po.input_tokens.shape), "feature rows nonzero", int((po.input_features.float().abs().sum(1) > 0).sum()))
    out["cases"] = np.array(json.dumps([(t, k, {a: b for a, b in kw.items()}) for t, k, kw in cases]))
    np.savez_compressed(os.path.join(HERE, "g14_qwen3_preprocess.npz"), **out)


def g15_speaker_encoder(ns):
    """Qwen3TTSSpeakerEncoder + mel_spectrogram of the reference (qwen3_tts.py:21-88, 835-891), run in fp32 over bf16-valued
    weights, tiny and full size (mel 128 -> 2048, the 1.7B base checkpoint's shape).  librosa is absent: its mel filterbank is
    the restatement in oracle/spk_ref.py (the one third-party piece of this path; see that file's header)."""
    from oracle import spk_ref as SR
    Q = ns.qwen3_tts
    out = {}
    for tag, cfg, n, seed in (("tiny", SR.tiny_spk_cfg(), 9000, 3), ("full", SR.SpkCfg(), 60000, 4)):
        Q.librosa_mel_fn = lambda sr, n_fft, n_mels, fmin, fmax: SR.mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
        rc = Q.Qwen3TTSSpeakerEncoderConfig(enc_dim=cfg.enc_dim, sample_rate=cfg.sample_rate, mel_dim=cfg.mel_dim,
                                            enc_channels=cfg.enc_channels, enc_kernel_sizes=cfg.enc_kernel_sizes,
                                            enc_dilations=cfg.enc_dilations, enc_res2net_scale=cfg.enc_res2net_scale,
                                            enc_se_channels=cfg.enc_se_channels, enc_attention_channels=cfg.enc_attention_channels)
        net = Q.Qwen3TTSSpeakerEncoder(rc).float().eval()
        W = SR.random_spk_weights(cfg, seed=seed)
        missing, unexpected = net.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=True)
        audio = SR.test_audio(seed, n)
        mels = Q.mel_spectrogram(torch.from_numpy(audio)[None], n_fft=1024, num_mels=cfg.mel_dim, sampling_rate=24000, hop_size=256,
                                 win_size=1024, fmin=0, fmax=12000).transpose(1, 2)
        emb = net(mels)[0]
        emb_bf16 = net.to(torch.bfloat16)(mels.to(torch.bfloat16))[0]
        out[f"{tag}_mel"], out[f"{tag}_emb"] = mels[0].numpy().astype(np.float32), emb.numpy().astype(np.float32)
        out[f"{tag}_emb_bf16"] = emb_bf16.float().numpy()
        out[f"{tag}_seed"], out[f"{tag}_n"] = np.int64(seed), np.int64(n)
        print("g15", tag, "mel", tuple(mels.shape), "emb rms", float(emb.pow(2).mean().sqrt()),
              "bf16 run rel", float((emb_bf16.float() - emb).pow(2).mean().sqrt() / emb.pow(2).mean().sqrt()))
    np.savez_compressed(os.path.join(HERE, "g15_speaker_encoder.npz"), **out)


def g16_codec_encoder(ns):
    """The speech tokenizer's encoder as the reference wires it (Qwen3TTSTokenizerV2Encoder(MimiModel) behind
    Qwen3TTSTokenizerV2Model.encode, qwen3_codec.py:1669-1773), fp32, tiny and full size: pre-quantisation latents and codes."""
    import transformers
    from transformers import MimiConfig
    from oracle import codec_enc_ref as ER, spk_ref as SR
    C = ns.qwen3_codec
    out = {"transformers_version": np.array(transformers.__version__)}
    for tag, cfg, n, seed in (("tiny", ER.tiny_codec_enc_cfg(), 1000, 5), ("full", ER.CodecEncCfg(), 52000, 6)):
        mc = MimiConfig(audio_channels=1, codebook_dim=cfg.codebook_dim, codebook_size=cfg.codebook_size, compress=cfg.compress,
                        dilation_growth_rate=2, head_dim=cfg.head_dim, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                        kernel_size=cfg.kernel_size, last_kernel_size=cfg.last_kernel_size, num_filters=cfg.num_filters,
                        num_hidden_layers=cfg.num_layers, num_attention_heads=cfg.num_heads, num_key_value_heads=cfg.num_heads,
                        num_quantizers=cfg.num_quantizers, num_residual_layers=1, pad_mode="constant",
                        residual_kernel_size=cfg.residual_kernel_size, rope_theta=cfg.rope_theta, sampling_rate=24000,
                        sliding_window=cfg.sliding_window, trim_right_ratio=1.0, upsample_groups=cfg.hidden_size,
                        upsampling_ratios=list(reversed(cfg.ratios)), use_cache=False, use_conv_shortcut=False,
                        vector_quantization_hidden_dimension=cfg.codebook_dim, num_semantic_quantizers=cfg.num_semantic_quantizers,
                        norm_eps=cfg.norm_eps)
        enc = C.Qwen3TTSTokenizerV2Encoder(mc).float().eval()
        W = ER.random_codec_enc_weights(cfg, seed=seed)
        missing, unexpected = enc.load_state_dict(W, strict=False)
        assert not unexpected, unexpected
        assert all(("initialized" in k) or ("output_proj" in k) or ("inv_freq" in k) for k in missing), missing
        for m_ in enc.modules():                      # the codebook property caches embed_sum / cluster_usage
            if hasattr(m_, "_embed"):
                m_._embed = None
        stub = types.SimpleNamespace(encoder=enc, encoder_valid_num_quantizers=cfg.valid_quantizers, encode_downsample_rate=cfg.hop)
        wav = torch.from_numpy(SR.test_audio(seed, n))
        codes = C.Qwen3TTSTokenizerV2Model.encode(stub, wav[None], torch.ones(1, n, dtype=torch.long))[0]
        emb = enc.encoder(wav.view(1, 1, -1))
        lat = enc.downsample(enc.encoder_transformer(emb.transpose(1, 2))[0].transpose(1, 2))[0].transpose(0, 1)
        out[f"{tag}_codes"], out[f"{tag}_latents"] = codes.numpy().astype(np.int32), lat.numpy().astype(np.float32)
        out[f"{tag}_seanet"] = emb[0].transpose(0, 1).numpy().astype(np.float32)[:: max(1, emb.shape[-1] // 16)]
        out[f"{tag}_seed"], out[f"{tag}_n"] = np.int64(seed), np.int64(n)
        print("g16", tag, "codes", tuple(codes.shape), "latent rms", float(lat.pow(2).mean().sqrt()), "distinct codes per layer",
              [int(codes[:, q].unique().numel()) for q in range(codes.shape[1])][:6])
    np.savez_compressed(os.path.join(HERE, "g16_codec_encoder.npz"), **out)


def g17_flow_evolving(ns):
    """CosyVoice2Decoder.decode_chunk in the per-request mode (shared_prompt_cache_mode=False, i.e. use_detokenizer_cache=True:
    cosyvoice2.py:1010-1083) through the reference, tiny size, one request, three consecutive 28-token chunks: the caches grow and are
    cut back to the sliding window (the third chunk runs against truncated caches), the fade-in blends against the previous chunk's tail.
    Pins the oracle restatement (oracle/flow_ref.py::decode_chunk_evolving); tests/test_gpu_flow.py holds the HIP path to both."""
    from oracle import flow_ref as FR, hift_ref as HR
    import contextlib
    fc, hc, Np = FR.tiny_flow_cfg(), HR.HiftCfg(base_channels=256, f0_channels=64), 9
    Wf, Wh = FR.random_flow_weights(fc, seed=3), HR.random_hift_weights(hc, seed=2)
    stub, C2, flow = _ref_cosyvoice2_decoder(ns, fc, hc, Wf, Wh)
    stub.shared_prompt_cache_mode = False
    g = torch.Generator().manual_seed(23)
    T, n_chunks, seed = 28, 3, 35
    ptok = torch.randint(0, fc.vocab, (1, Np), generator=g)
    pfeat = (0.7 * torch.randn(1, 2 * Np, fc.mel, generator=g)).to(torch.bfloat16).float()
    spk = torch.randn(1, fc.spk_dim, generator=g).to(torch.bfloat16).float()
    toks = torch.randint(0, fc.vocab, (n_chunks, 1, T), generator=g)
    queue, cur = [], {}
    real = (torch.randn, torch.rand, torch.randn_like)

    def fake_randn(*shape, **kw):
        shape = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
        z = queue.pop(0)
        assert tuple(z.shape) == shape, (z.shape, shape)
        return z.clone()

    def fake_rand(*shape, **kw):
        return cur["ini"].clone()

    def fake_randn_like(t, **kw):
        return cur["nz"].clone() if tuple(t.shape) == tuple(cur["nz"].shape) else torch.zeros_like(t)
    ref_dict = {"prompt_speech_token": ptok, "prompt_speech_token_len": Np, "prompt_feat": pfeat, "prompt_feat_len": 2 * Np, "embedding": spk}
    out = {"prompt_token": ptok.numpy().astype(np.int32), "prompt_feat": pfeat.numpy(), "spk": spk.numpy(),
           "tokens": toks.numpy().astype(np.int32), "noise_seed": np.int64(seed)}
    torch.randn, torch.rand, torch.randn_like = fake_randn, fake_rand, fake_randn_like
    try:
        with contextlib.redirect_stdout(open(os.devnull, "w")), torch.no_grad():
            queue.append(FR.cfm_noise(seed, 0, fc.mel, 2 * (Np + 3)))
            cache = C2.CosyVoice2Decoder.init_cache(stub, ref_dict)
            lens = []
            for k in range(n_chunks):
                cur["ini"], cur["nz"] = HR.make_noise(hc, 1, 2 * T, seed=seed, first_stream=16 + 2 * k)
                queue.append(FR.cfm_noise(seed, 1 + k, fc.mel, 2 * T))
                audio, cache = C2.CosyVoice2Decoder.decode_chunk(stub, toks[k], T, cache, ref_dict=ref_dict)
                out[f"audio_{k}"] = audio.float().numpy().astype(np.float32)
                ec, dc = cache.flow_encoder_cache, cache.flow_decoder_cache
                lens.append([ec.conformer_att_cache.shape[3], ec.up_conformer_att_cache.shape[3], dc.att_cache.shape[5]])
                out[f"att_cache_sum_{k}"] = np.float64(dc.att_cache.double().sum().item())
                out[f"speech_cache_{k}"] = cache.hift_cache.speech_cache.float().numpy().astype(np.float32)
    finally:
        torch.randn, torch.rand, torch.randn_like = real
    assert not queue
    out["cache_lens"] = np.array(lens, np.int32)
    print("g17 cache lens per chunk", lens, "audio rms", [float(np.sqrt((out[f"audio_{k}"] ** 2).mean())) for k in range(n_chunks)])
    np.savez_compressed(os.path.join(HERE, "g17_flow_evolving.npz"), **out)


ALL = {"g1": g1_sampler, "g2": g2_wrappers, "g3": g3_qwen3_lm, "g18": g18_qwen3_lm_b12, "g21": g21_qwen3_full_width, "g22": g22_glm_full_width, "g19": g19_sampler_mc, "g4": g4_qwen3_codec, "g6": g6_host_traces,
       "g7": g7_single_stack_lms, "g8": g8_scheduler_policies, "g9": g9_csm_lm, "g23": g23_csm_full_width, "g5": g5_mimi, "g10": g10_snac, "g11": g11_hift, "g12": g12_flow, "g13": g13_glm_decoder, "g14": g14_qwen3_preprocess, "g15": g15_speaker_encoder, "g16": g16_codec_encoder, "g17": g17_flow_evolving, "g20": g20_snac_variants}

if __name__ == "__main__":
    ns = H.boot()
    torch.manual_seed(0)
    for k in (sys.argv[1:] or list(ALL)):
        with torch.no_grad():
            ALL[k](ns)
