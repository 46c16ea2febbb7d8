"""Parity of the single-stack speech-LM engine (libvoxhip vox_lm_* through the C ABI: GLM-4-Voice, CosyVoice2) against
the CPU oracle (oracle/lm_ref.py), which is itself pinned to the reference modules by tests/golden/g7.

Bar: <= 8 rows per call -> BIT-EXACT logits, sampled ids, repetition caches and KV contents, free-running, under
greedy, seeded top-k, seeded top-p-only (the full-vocabulary sampler) and repetition penalty with a persisted
cache.  Longer prefills run the bf16-MFMA linears: statistical bf16 bar, then the oracle adopts the GPU state.
"""
import numpy as np
import pytest
import torch

from oracle import lm_ref as LR
from oracle import voxref as vr
from tests.conftest import bf16_close
from tests.oracle_tape import Tape, Weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda")


def build(dev, family, cfg, St, B, page, max_pages, rep_window=None, max_seq_len=512):
    """Reference-named state dict St (torch bf16 on the device) -> product plugin packers -> LMEngine."""
    from vox_serve_amd.engine import LMEngine
    c = cfg.stack
    if family == "glm":
        from vox_serve_amd.model.glm_voice import GLMVoiceConfig, pack_glm_weights
        pc = GLMVoiceConfig(ffn_hidden_size=c.ffn, hidden_size=c.hidden, multi_query_group_num=c.kv_heads,
                            num_attention_heads=c.heads, num_layers=c.layers, padded_vocab_size=cfg.vocab_out,
                            vocab_size=cfg.vocab_out)
        layers, norm, emb, head = pack_glm_weights(St, pc)
        head_b = None
    else:
        from vox_serve_amd.model.cosyvoice2 import CosyVoice2Config, pack_cosyvoice2_weights
        pc = CosyVoice2Config(llm_input_size=c.hidden, llm_output_size=c.hidden, speech_token_size=cfg.vocab_out - 3,
                              hidden_size=c.hidden, intermediate_size=c.ffn, num_attention_heads=c.heads,
                              num_key_value_heads=c.kv_heads, num_hidden_layers=c.layers)
        layers, norm, emb, head, head_b = pack_cosyvoice2_weights(St, pc)
    ecfg = pc.lm_cfg(max_pos=cfg.max_pos)
    return LMEngine(ecfg, layers, norm, emb, head, head_b, max_batch=B, page_size=page, max_pages=max_pages,
                    max_seq_len=max_seq_len, max_prefill_rows=128, rep_window=rep_window, device=dev)


def run_parity(dev, family, cfg, S, prompt_lens, n_steps, page=16, max_pages=48, sampler_kw=None, penalty=1.0, window=None,
               policy=None, tape=None):
    """S: numpy state dict or tests.oracle_tape.Weights.  tape: None = live oracle (tests/oracle_tape.py)."""
    tape = tape or Tape()
    S = S if isinstance(S, Weights) else Weights(S)
    rng = np.random.default_rng(11)
    B = len(prompt_lens)
    W = (LR.from_glm_state_dict if family == "glm" else LR.from_cosyvoice2_state_dict)(cfg, S.numpy()) if tape.oracle else None
    ref = LR.LMRef(cfg, W, page_size=page, max_pages=max_pages, policy=policy, dry=not tape.oracle)
    use_rep = penalty != 1.0
    eng = build(dev, family, cfg, S.torch(dev), B, page, max_pages, rep_window=window if use_rep else None) if tape.gpu else None
    seed, step = 4321, [0]
    sampler = (lambda lg: vr.sample(lg, seed=seed, offset=step[0], **sampler_kw)) if sampler_kw else None
    H, V = cfg.stack.hidden, cfg.vocab_out
    Wn = (window if window and window > 0 else 1)
    if eng:
        sc = eng.sampling_cfg(greedy=not sampler_kw, repetition_penalty=penalty, **(sampler_kw or {}))
        state_ids = torch.zeros(B, 1, dtype=torch.int32, device=dev)
        state_rep = torch.zeros(B, Wn, 1, V, dtype=torch.uint8, device=dev)
    reqs = []
    for r, n in enumerate(prompt_lens):
        ids = rng.integers(0, cfg.vocab_in if family == "glm" else 640, n).astype(np.int32)
        masks = feats = None
        if cfg.input_mode == 1:
            masks = np.ones(n, np.uint8)
            masks[n // 2] = 0                                   # one row falls back to the embedding table
            ids[n // 2] = rng.integers(0, cfg.vocab_in)
            feats = vr.f2bf((0.08 * rng.standard_normal((n, H))).astype(np.float32))
        req = LR.LMRequest(rep_cache=np.zeros((Wn, 1, V), np.uint8) if use_rep else None)
        lg = ref.prefill(req, ids, masks, feats)
        tok, pen = ref.sample(lg, [req], sampler, penalty, window)
        if eng:
            eng.row_ids[:n, 0] = torch.from_numpy(ids).to(dev)
            if masks is not None:
                eng.row_masks[:n] = torch.from_numpy(masks).to(dev)
                eng.row_feats[:n] = vr.to_torch(feats).to(dev)
            eng.upload_plan(pos=np.arange(n), kvlen=np.arange(1, n + 1), page=[req.kv_pages[t // page] for t in range(n)],
                            slot=[t % page for t in range(n)], q_req=np.zeros(n), last_rows=[n - 1],
                            indptr=[0, len(req.kv_pages)], indices=req.kv_pages)
            eng.rng_offset.fill_(step[0])
            if use_rep:
                eng.rep_cache[0].zero_()
            eng.prefill(n, 1, n, sc, seed=seed, feedback=True)
            torch.cuda.synchronize()
        # bit-exact at every prompt length (MFMA prefills incl.)
        tape.check(f"prefill logits r{r}", lambda: vr.from_torch(eng.out_logits[:1]), lambda: pen)
        tape.check(f"prefill token r{r}", lambda: eng.out_ids[:1].cpu().numpy(), lambda: tok)
        tape.check(f"prefill feedback r{r}", lambda: eng.input_ids[:1, :1].cpu().numpy(), lambda: req.input_ids[:1, :1])
        if use_rep:
            tape.check(f"rep cache r{r}", lambda: eng.rep_cache[0].cpu().numpy(), lambda: req.rep_cache)
        if eng:
            if cfg.input_mode == 1:
                assert int(eng.input_masks[0]) == 0
            state_ids[r] = eng.input_ids[0]
            if use_rep:
                state_rep[r] = eng.rep_cache[0]
        reqs.append(req)
    step[0] = 1
    if eng:
        eng.input_ids[:B] = state_ids
        eng.input_masks[:B] = 0
        if use_rep:
            eng.rep_cache[:B] = state_rep
        eng.rng_offset.fill_(step[0])
    for f in range(n_steps):
        lg = ref.decode(reqs)
        tok, pen = ref.sample(lg, reqs, sampler, penalty, window)
        if eng:
            indptr, indices = [0], []
            for q in reqs:
                indptr.append(indptr[-1] + len(q.kv_pages))
                indices += q.kv_pages
            eng.upload_plan(pos=[q.next_position_id - 1 for q in reqs], kvlen=[q.kv_token_len for q in reqs],
                            page=[q.kv_pages[-1] for q in reqs], slot=[q.kv_last_page_len - 1 for q in reqs],
                            indptr=indptr, indices=indices)
            eng.frame(B, max(q.kv_token_len for q in reqs), sc, seed=seed, feedback=True, use_graph=True)
            torch.cuda.synchronize()
        tape.check(f"logits step {f}", lambda: vr.from_torch(eng.out_logits[:B]), lambda: pen)
        tape.check(f"tokens step {f}", lambda: eng.out_ids[:B].cpu().numpy(), lambda: tok)
        if use_rep:
            tape.check(f"rep {f}", lambda: eng.rep_cache[:B].cpu().numpy(), lambda: np.stack([q.rep_cache for q in reqs]))
        step[0] += 1
    used = sorted({p for q in reqs for p in q.kv_pages})
    tape.check("kv", lambda: vr.from_torch(eng.kv)[:, used], lambda: np.stack([l[used] for l in ref.kv]))
    if eng:
        eng.close()
    tape.done(kind=family, prompt_lens=list(prompt_lens), n_steps=n_steps)
    return [q.tokens for q in reqs]


def test_glm_tiny_greedy(dev):
    cfg = LR.tiny_glm_cfg()
    run_parity(dev, "glm", cfg, LR.random_glm_state_dict(cfg, seed=3, std=0.08), [5, 8, 3], 20)


def test_glm_tiny_top_p_only(dev):
    """GLM-4-Voice defaults: top_p 0.8, temperature 0.8, no top-k (glm_voice.py:358-366) -> bucket sampler."""
    cfg = LR.tiny_glm_cfg()
    toks = run_parity(dev, "glm", cfg, LR.random_glm_state_dict(cfg, seed=3, std=0.08), [6, 7], 24,
                      sampler_kw=dict(top_k=0, top_p=0.8, temperature=0.8))
    assert len({t for q in toks for t in q}) > 8          # it really samples


def test_glm_tiny_long_prefill_then_exact_decode(dev):
    cfg = LR.tiny_glm_cfg()
    run_parity(dev, "glm", cfg, LR.random_glm_state_dict(cfg, seed=3, std=0.08), [37, 21], 12)


def test_cosyvoice2_tiny_greedy_and_topk(dev):
    cfg = LR.tiny_cosyvoice2_cfg()
    S = LR.random_cosyvoice2_state_dict(cfg, seed=4, std=0.08)
    run_parity(dev, "cosy", cfg, S, [7, 4, 8], 20)
    run_parity(dev, "cosy", cfg, S, [7, 4], 20, sampler_kw=dict(top_k=25, temperature=1.0))


def test_cosyvoice2_tiny_repetition_window(dev):
    """Persisted per-request repetition cache, sliding window 2 and global window, penalty 2.0 (sampling.py:121-178)."""
    cfg = LR.tiny_cosyvoice2_cfg()
    S = LR.random_cosyvoice2_state_dict(cfg, seed=4, std=0.08)
    run_parity(dev, "cosy", cfg, S, [8, 6], 16, penalty=2.0, window=2)
    run_parity(dev, "cosy", cfg, S, [8, 6], 16, penalty=2.0, window=-1, sampler_kw=dict(top_k=0, top_p=0.9, temperature=0.7))


def test_against_reference_goldens(dev, golden):
    """GPU engine vs the tokens the reference modules produced (g7): prompts of 11/19 and 9/14 tokens take the MFMA
    prefill, so logits are compared at bf16 tolerance and greedy tokens may flip only on near-ties."""
    g = golden("g7_single_stack_lms")
    page, P = int(g["page"]), int(g["P"])
    for fam, tag in (("glm", "glm"), ("cosy", "cosy")):
        if fam == "glm":
            cfg = LR.tiny_glm_cfg()
            S = LR.random_glm_state_dict(cfg, seed=3, std=0.08)
        else:
            cfg = LR.tiny_cosyvoice2_cfg()
            S = LR.random_cosyvoice2_state_dict(cfg, seed=4, std=0.08)
        eng = build(dev, fam, cfg, {k: vr.to_torch(v).to(dev) for k, v in S.items()}, 2, page, P)
        sc = eng.sampling_cfg(greedy=True)
        pages, lens, free, mism = [], [], list(range(P)), 0
        for r in range(2):
            ids = g[f"{fam}_r{r}_ids"]
            n = len(ids)
            pg = [free.pop(0) for _ in range((n + page - 1) // page)]
            eng.row_ids[:n, 0] = torch.from_numpy(ids).to(dev)
            if fam == "cosy":
                eng.row_masks[:n] = 1
                eng.row_feats[:n] = vr.to_torch(g[f"cosy_r{r}_feats"]).to(dev)
            eng.upload_plan(pos=np.arange(n), kvlen=np.arange(1, n + 1), page=[pg[t // page] for t in range(n)],
                            slot=[t % page for t in range(n)], q_req=np.zeros(n), last_rows=[n - 1], indptr=[0, len(pg)],
                            indices=pg)
            eng.prefill(n, 1, n, sc, feedback=False)
            torch.cuda.synchronize()
            assert bf16_close(vr.from_torch(eng.out_logits[:1]), g[f"{tag}_r{r}_prefill_logits"], ulps=4, atol=6e-2).all()
            mism += int(eng.out_ids[0].item() != g[f"{tag}_r{r}_tok0"][0])
            pages.append(pg)
            lens.append(n)
        toks = np.array([g[f"{tag}_r{r}_tok0"][0] for r in range(2)], np.int32)
        for f in range(4):
            eng.input_ids[:2, 0] = torch.from_numpy(toks).to(dev)          # teacher forcing with the reference's tokens
            eng.input_masks[:2] = 0
            lens = [n + 1 for n in lens]
            for r in range(2):
                if lens[r] > len(pages[r]) * page:
                    pages[r].append(free.pop(0))
            eng.upload_plan(pos=g[f"{tag}_f{f}_pos"], kvlen=lens, page=[pages[r][(lens[r] - 1) // page] for r in range(2)],
                            slot=[(lens[r] - 1) % page for r in range(2)], indptr=[0, len(pages[0]), len(pages[0]) + len(pages[1])],
                            indices=pages[0] + pages[1])
            eng.frame(2, max(lens), sc, feedback=False)
            torch.cuda.synchronize()
            assert bf16_close(vr.from_torch(eng.out_logits[:2]), g[f"{tag}_f{f}_logits"], ulps=4, atol=6e-2).all(), (tag, f)
            mism += int((eng.out_ids[:2].cpu().numpy() != g[f"{tag}_f{f}_tokens"][:, 0]).sum())
            toks = g[f"{tag}_f{f}_tokens"][:, 0].astype(np.int32)
        assert mism <= 1, (tag, mism)
        eng.close()


@pytest.mark.parametrize("tag", ["b1", "b8"])
def test_glm_full_width_layer_against_reference_fixture(dev, golden, tag):
    """g22 (round 5): the HIP engine against logits the REFERENCE's GLM-4-Voice modules produced for ONE full-width layer (K = 4096 /
    13696, 32 q / 2 kv heads, QKV bias, half-rotary interleaved RoPE) at 1 and 8 requests — prefills of 2..9 rows and two
    teacher-forced decode steps; relative RMS of the logits <= 1e-2 (bf16 rounding noise), greedy ids equal except near-ties."""
    from oracle import lm_wide as LW
    g = golden(f"g22_glm_full_width_{tag}")
    cfg = LW.wide_glm_cfg()
    S = LR.random_glm_state_dict(cfg, seed=LW.WEIGHT_SEED, std=LW.WEIGHT_STD, device=dev)
    page, P = int(g["page"]), int(g["P"])
    lens = g["prompt_lens"].tolist()
    B = len(lens)
    eng = build(dev, "glm", cfg, S, B, page, P)
    sc = eng.sampling_cfg(greedy=True)
    rel = lambda a, b: float(np.sqrt(((vr.bf2f(a).astype(np.float64) - vr.bf2f(b)) ** 2).mean() / (vr.bf2f(b).astype(np.float64) ** 2).mean()))
    pages, free, mism, worst = [], list(range(P)), 0, 0.0
    for r, n in enumerate(lens):
        ids = g[f"glm_r{r}_ids"]
        pg = [free.pop(0) for _ in range((n + page - 1) // page)]
        eng.row_ids[:n, 0] = torch.from_numpy(ids).to(dev)
        eng.upload_plan(pos=np.arange(n), kvlen=np.arange(1, n + 1), page=[pg[t // page] for t in range(n)], slot=[t % page for t in range(n)],
                        q_req=np.zeros(n), last_rows=[n - 1], indptr=[0, len(pg)], indices=pg)
        eng.prefill(n, 1, n, sc, feedback=False)
        torch.cuda.synchronize()
        worst = max(worst, rel(vr.from_torch(eng.out_logits[:1]), g[f"glm_r{r}_prefill_logits"]))
        mism += int(eng.out_ids[0].item() != g[f"glm_r{r}_tok0"][0])
        pages.append(pg)
    toks = np.array([g[f"glm_r{r}_tok0"][0] for r in range(B)], np.int32)
    for f in range(int(g["n_steps"])):
        eng.input_ids[:B, 0] = torch.from_numpy(toks).to(dev)          # teacher forcing with the reference's tokens
        eng.input_masks[:B] = 0
        lens = [n + 1 for n in lens]
        for r in range(B):
            if lens[r] > len(pages[r]) * page:
                pages[r].append(free.pop(0))
        indptr = np.cumsum([0] + [len(p_) for p_ in pages])
        eng.upload_plan(pos=g[f"glm_f{f}_pos"], kvlen=lens, page=[pages[r][(lens[r] - 1) // page] for r in range(B)],
                        slot=[(lens[r] - 1) % page for r in range(B)], indptr=indptr, indices=sum(pages, []))
        eng.frame(B, max(lens), sc, feedback=False)
        torch.cuda.synchronize()
        worst = max(worst, rel(vr.from_torch(eng.out_logits[:B]), g[f"glm_f{f}_logits"]))
        mism += int((eng.out_ids[:B].cpu().numpy() != g[f"glm_f{f}_tokens"][:, 0]).sum())
        toks = g[f"glm_f{f}_tokens"][:, 0].astype(np.int32)
    eng.close()
    assert worst <= 1.0e-2, worst
    assert mism <= (1 if tag == "b1" else 3), mism


# ---- heavy cases: the oracle side is recorded ahead of time (tests/oracle_tape.py, tests/golden/make_oracle_tapes.py) ----------
TAPED = {}


def taped(case):
    def deco(fn):
        TAPED[case] = fn
        return fn
    return deco


def glm_full_width_cfg():
    return LR.glm_cfg(layers=2, max_pos=512)


# GLM-4-Voice-9B layer shapes, 2 layers (the two 168960 x 4096 tables are 1.4 G values), built once per session and side
_GLM_FULL = Weights(lambda device=None: LR.random_glm_state_dict(glm_full_width_cfg(), seed=1, std=0.02, device=device))
_GLM_LENS = [4, 6, 3, 5, 7, 2, 8, 5]


@taped("glm_full_width_two_requests_top_p")
def case_glm_full_width_two_layers(tape, dev):
    run_parity(dev, "glm", glm_full_width_cfg(), _GLM_FULL, [4, 6], 3, page=128, max_pages=8,
               sampler_kw=dict(top_k=0, top_p=0.8, temperature=0.8), tape=tape)


@taped("glm_full_width_b8_exact_rows_2")
def case_glm_full_width_b8(tape, dev):
    run_parity(dev, "glm", glm_full_width_cfg(), _GLM_FULL, _GLM_LENS, 2, page=128, max_pages=16, tape=tape)


@taped("glm_full_width_b8_exact_rows_8")
def case_glm_full_width_b8_er8(tape, dev):
    from oracle.policy import Policy
    run_parity(dev, "glm", glm_full_width_cfg(), _GLM_FULL, _GLM_LENS, 2, page=128, max_pages=16, policy=Policy(exact_rows=8), tape=tape)


def glm_full_depth_cfg():
    return LR.glm_cfg(layers=40, max_pos=512)


# the WHOLE GLM-4-Voice-9B stack: 40 layers at full width + both 168960 x 4096 tables (9.4 G parameters: 19 GB of bf16 on either side)
_GLM_DEPTH = Weights(lambda device=None: LR.random_glm_state_dict(glm_full_depth_cfg(), seed=1, std=0.02, device=device))


@taped("glm_full_depth_b8")
def case_glm_full_depth_b8(tape, dev):
    run_parity(dev, "glm", glm_full_depth_cfg(), _GLM_DEPTH, [1, 2, 1, 2, 1, 2, 1, 2], 1, page=128, max_pages=16, tape=tape)


@taped("cosyvoice2_full_size_top_k")
def case_cosyvoice2_full_size(tape, dev):
    cfg = LR.cosyvoice2_cfg(max_pos=512)
    S = Weights(lambda device=None: LR.random_cosyvoice2_state_dict(cfg, seed=2, std=0.02, device=device))
    run_parity(dev, "cosy", cfg, S, [5, 3], 6, page=128, max_pages=8, sampler_kw=dict(top_k=25, temperature=1.0), tape=tape)


@pytest.mark.slow
def test_glm_full_width_two_layers(dev):
    """GLM-4-Voice-9B layer shapes (4096 hidden, 32/2 heads, FFN 13696, vocab 168960), 2 of the 40 layers: bit-exact
    decode under the model's default top-p-only sampling over the full 168960-entry vocabulary."""
    case_glm_full_width_two_layers(Tape.open("glm_full_width_two_requests_top_p"), dev)


@pytest.mark.slow
def test_glm_full_width_b8_both_settings(dev):
    """BASELINE config 4's per-GPU share — 8 concurrent requests — at GLM-4-Voice-9B layer shapes (2 of 40 layers, full
    168960-entry vocabulary).  (a) default (`exact_rows 2`): every call with more than 2 rows runs on the matrix cores
    (K = 4096: normalise-once + full-K GEMM; down_proj K = 13696: 4-wave GEMM); (b) `exact_rows 8`: 8 rows stay on the wave64
    VALU kernels.  Both are bit-exact against the oracle under the same policy, prefills included."""
    from vox_serve_amd import _native as N
    case_glm_full_width_b8(Tape.open("glm_full_width_b8_exact_rows_2"), dev)
    N.set_exact_rows(8)
    try:
        case_glm_full_width_b8_er8(Tape.open("glm_full_width_b8_exact_rows_8"), dev)
    finally:
        N.set_exact_rows(2)


@pytest.mark.slow
def test_glm_full_depth_b8(dev):
    """Round-4 review, parity tail: GLM-4-Voice-9B at its FULL depth — all 40 layers, 4096 wide, FFN 13696, the 168960-entry tables —
    at BASELINE config 4's per-GPU share of 8 requests: eight one- and two-token prefills and one 8-row decode frame through the
    40-layer stack (normalise-once + full-K GEMMs at K = 4096, the streamed K = 13696 down projection, 16-head-group attention),
    logits / greedy ids / fed-back ids / the K/V cache of all 40 layers bit-exact against the oracle's tape (the 2-of-40-layer
    cases above cover sampling modes and longer prompts; this one covers depth: error that only accumulates over 40 layers)."""
    case_glm_full_depth_b8(Tape.open("glm_full_depth_b8"), dev)
    _GLM_DEPTH.drop_torch()          # (19 GB of weights: not kept for the rest of the session)
    torch.cuda.empty_cache()


@pytest.mark.slow
def test_cosyvoice2_full_size(dev):
    """CosyVoice2-0.5B at full size (24 layers x 896, 14/2 heads of 64, FFN 4864, 6564 speech ids), top-k 25."""
    case_cosyvoice2_full_size(Tape.open("cosyvoice2_full_size_top_k"), dev)
