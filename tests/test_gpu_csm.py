"""Parity of the CSM-1B frame engine (libvoxhip vox_csm_* through the C ABI) against the CPU oracle
(oracle/csm_ref.py, pinned to the reference CSM modules by tests/golden/g9): ragged prefill, then free-running batched
decode — masked 33-column embedding sum, llama-3.1 RoPE backbone, 31-step (here 5-step) depth loop with per-codebook
heads.  Bar: <= 8 rows per call -> BIT-EXACT logits, hidden states, depth logits and sampled ids (greedy and seeded
top-k); longer prefills take the MFMA path (bf16 bar) and the oracle then adopts the GPU's state.
"""
import numpy as np
import pytest
import torch

from oracle import csm_ref as CR
from oracle import qwen3_ref as QR
from oracle import voxref as vr
from tests.conftest import bf16_close
from tests.oracle_tape import Tape, Weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda")


def to_engine_cfg(c: CR.CSMCfg):
    from vox_serve_amd.engine import CSMCfg, StackCfg
    conv = lambda s: StackCfg(s.hidden, s.layers, s.heads, s.kv_heads, s.head_dim, s.ffn, s.eps, s.rope_theta, s.rope_scale,
                              None, False, s.rope_llama31, False, False)
    return CSMCfg(conv(c.backbone), conv(c.depth), c.vocab, c.text_vocab, c.n_codebooks, c.max_pos)


def make_prompt(rng, cfg, n_text, n_audio):
    C, n = cfg.n_codebooks, n_text + n_audio
    ids, masks = np.zeros((n, C + 1), np.int32), np.zeros((n, C + 1), np.uint8)
    ids[:n_text, -1] = rng.integers(0, cfg.text_vocab, n_text)
    masks[:n_text, -1] = 1
    if n_audio:
        ids[n_text:, :C] = rng.integers(1, cfg.vocab, (n_audio, C))
        masks[n_text:, :C] = 1
    return ids, masks


def run_parity(dev, cfg, W, prompts, n_frames, page=16, max_pages=64, sampler_kw=None, tape=None):
    """W: numpy state dict or tests.oracle_tape.Weights.  tape: None = live oracle (tests/oracle_tape.py)."""
    tape = tape or Tape()
    W = W if isinstance(W, Weights) else Weights(W)
    rng = np.random.default_rng(5)
    B, C, C1 = len(prompts), cfg.n_codebooks, cfg.n_codebooks + 1
    ref = CR.CSMRef(cfg, W.numpy() if tape.oracle else None, page_size=page, max_pages=max_pages, max_batch=B, dry=not tape.oracle)
    eng = None
    if tape.gpu:
        from vox_serve_amd.engine import CSMEngine
        eng = CSMEngine(to_engine_cfg(cfg), W.torch(dev), max_batch=B, page_size=page, max_pages=max_pages, max_seq_len=512,
                        max_prefill_rows=128, keep_depth_logits=True, device=dev)
    seed, frame_no = 99, [0]
    sampler = (lambda lg, i: vr.sample(lg, seed=seed, offset=frame_no[0] * C + i, **sampler_kw)) if sampler_kw else None
    if eng:
        sc = eng.sampling_cfg(greedy=False, **sampler_kw) if sampler_kw else eng.sampling_cfg(greedy=True)
        st_ids = torch.zeros(B, C1, dtype=torch.int32, device=dev)
        st_masks = torch.zeros(B, C1, dtype=torch.uint8, device=dev)
    reqs = []
    for r, (nt, na) in enumerate(prompts):
        ids, masks = make_prompt(rng, cfg, nt, na)
        n = nt + na
        req = QR.RefRequest()
        lg, hid = ref.prefill(req, ids, masks)
        out, _, _, dl = ref.frame([req], lg, hid, sampler)
        if eng:
            eng.row_ids[:n] = torch.from_numpy(ids).to(dev)
            eng.row_masks[:n] = torch.from_numpy(masks).to(dev)
            eng.upload_plan(pos=np.arange(n), kvlen=np.arange(1, n + 1), page=[req.kv_pages[t // page] for t in range(n)],
                            slot=[t % page for t in range(n)], q_req=np.zeros(n), last_rows=[n - 1],
                            indptr=[0, len(req.kv_pages)], indices=req.kv_pages)
            eng.rng_offset.fill_(frame_no[0])
            eng.prefill(n, 1, n, sc, seed=seed, feedback=True)
            torch.cuda.synchronize()
        tape.check(f"prefill hidden r{r}", lambda: vr.from_torch(eng.out_hidden[:1]), lambda: hid)       # bit-exact at every length
        tape.check(f"prefill logits r{r}", lambda: vr.from_torch(eng.out_logits[:1]), lambda: lg)
        tape.check(f"prefill depth r{r}", lambda: vr.from_torch(eng.out_depth_logits[:, 0]), lambda: np.stack(dl)[:, 0])
        tape.check(f"prefill tokens r{r}", lambda: eng.out_ids[:1].cpu().numpy(), lambda: out)
        tape.check(f"feedback ids r{r}", lambda: eng.input_ids[:1].cpu().numpy(), lambda: req.input_ids)
        tape.check(f"feedback masks r{r}", lambda: eng.input_masks[:1].cpu().numpy(), lambda: req.input_mask)
        if eng:
            st_ids[r], st_masks[r] = eng.input_ids[0], eng.input_masks[0]
        reqs.append(req)
    frame_no[0] = 1
    if eng:
        eng.input_ids[:B], eng.input_masks[:B] = st_ids, st_masks
        eng.rng_offset.fill_(frame_no[0])
    for f in range(n_frames):
        lg, hid = ref.decode(reqs)
        out, _, _, dl = ref.frame(reqs, lg, hid, sampler)
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
        tape.check(f"hidden f{f}", lambda: vr.from_torch(eng.out_hidden[:B]), lambda: hid)
        tape.check(f"logits f{f}", lambda: vr.from_torch(eng.out_logits[:B]), lambda: lg)
        tape.check(f"depth logits f{f}", lambda: vr.from_torch(eng.out_depth_logits[:, :B]), lambda: np.stack(dl))
        tape.check(f"tokens f{f}", lambda: eng.out_ids[:B].cpu().numpy(), lambda: out)
        frame_no[0] += 1
    used = sorted({p for q in reqs for p in q.kv_pages})
    tape.check("kv", lambda: vr.from_torch(eng.kv)[:, used], lambda: np.stack([l[used] for l in ref.kv]))
    if eng:
        eng.close()
    tape.done(kind="csm", prompts=[list(p_) for p_ in prompts], n_frames=n_frames)


def test_csm_tiny_greedy(dev):
    cfg = CR.tiny_csm_cfg()
    run_parity(dev, cfg, CR.random_csm_state_dict(cfg, 7, 0.08), [(5, 3), (8, 0), (2, 2)], 24)


def test_csm_tiny_topk(dev):
    """CSM defaults: top_k 50, temperature 0.9 (csm.py:355-363), seeded Philox contract"""
    cfg = CR.tiny_csm_cfg()
    run_parity(dev, cfg, CR.random_csm_state_dict(cfg, 8, 0.08), [(4, 4), (6, 1)], 20, sampler_kw=dict(top_k=50, temperature=0.9))


def test_csm_tiny_long_prefill_then_exact_decode(dev):
    cfg = CR.tiny_csm_cfg()
    run_parity(dev, cfg, CR.random_csm_state_dict(cfg, 9, 0.08), [(30, 11), (12, 9)], 10)


def test_csm_against_reference_goldens(dev, golden):
    """GPU engine vs the tokens the reference CSM modules produced (g9); prompts of 12 / 10 rows take the MFMA prefill."""
    from vox_serve_amd.engine import CSMEngine
    g = golden("g9_csm_lm")
    cfg = CR.tiny_csm_cfg()
    W = CR.random_csm_state_dict(cfg, seed=7, std=0.08)
    page, P, C1 = int(g["page"]), int(g["P"]), cfg.n_codebooks + 1
    eng = CSMEngine(to_engine_cfg(cfg), {k: vr.to_torch(v).to(dev) for k, v in W.items()}, max_batch=2, page_size=page,
                    max_pages=P, max_seq_len=512, max_prefill_rows=64, device=dev)
    sc = eng.sampling_cfg(greedy=True)
    free, pages, lens, mism = list(range(P)), [], [], 0
    for r in range(2):
        ids, masks = g[f"r{r}_ids"], g[f"r{r}_masks"]
        n = len(ids)
        pg = [free.pop(0) for _ in range((n + page - 1) // page)]
        eng.row_ids[:n], eng.row_masks[:n] = torch.from_numpy(ids).to(dev), torch.from_numpy(masks).to(dev)
        eng.upload_plan(pos=np.arange(n), kvlen=np.arange(1, n + 1), page=[pg[t // page] for t in range(n)],
                        slot=[t % page for t in range(n)], q_req=np.zeros(n), last_rows=[n - 1], indptr=[0, len(pg)], indices=pg)
        eng.prefill(n, 1, n, sc, feedback=False)
        torch.cuda.synchronize()
        assert bf16_close(vr.from_torch(eng.out_logits[:1]), g[f"r{r}_prefill_logits"], ulps=4, atol=6e-2).all()
        mism += int((eng.out_ids[0].cpu().numpy() != g[f"r{r}_frame0"]).sum())
        pages.append(pg)
        lens.append(n)
    for f in range(3):
        eng.input_ids[:2] = torch.from_numpy(g[f"f{f}_in_ids"]).to(dev)          # teacher forcing
        eng.input_masks[:2] = torch.from_numpy(g[f"f{f}_in_masks"]).to(dev)
        lens = [n + 1 for n in lens]
        for r in range(2):
            if lens[r] > len(pages[r]) * page:
                pages[r].append(free.pop(0))
        eng.upload_plan(pos=g[f"f{f}_pos"], kvlen=lens, page=[pages[r][(lens[r] - 1) // page] for r in range(2)],
                        slot=[(lens[r] - 1) % page for r in range(2)], indptr=[0, len(pages[0]), len(pages[0]) + len(pages[1])],
                        indices=pages[0] + pages[1])
        eng.frame(2, max(lens), sc, feedback=False)
        torch.cuda.synchronize()
        assert bf16_close(vr.from_torch(eng.out_logits[:2]), g[f"f{f}_logits"], ulps=4, atol=6e-2).all(), f
        mism += int((eng.out_ids[:2].cpu().numpy() != g[f"f{f}_tokens"]).sum())
    assert mism <= 8, mism
    eng.close()


def test_csm_full_width_layers_against_reference_fixture(dev, golden):
    """g23 (round 5): the HIP engine against numbers the REFERENCE's CSM modules produced for ONE backbone layer + ONE depth-decoder
    layer at the widths of CSM-1B, 16 requests (BASELINE config 4's batch), two teacher-forced frames: backbone logits and the first
    depth step's logits within relative RMS 1e-2 (bf16 rounding noise), codebook-0 ids equal except near-ties."""
    from oracle import csm_wide as CW
    from vox_serve_amd.engine import CSMEngine
    g = golden("g23_csm_full_width_b16")
    cfg = CW.wide_csm_cfg()
    W = CR.random_csm_state_dict(cfg, seed=CW.WEIGHT_SEED, std=CW.WEIGHT_STD, device=dev)
    page, P, B, NF = int(g["page"]), int(g["P"]), int(g["n_req"]), int(g["n_frames"])
    eng = CSMEngine(to_engine_cfg(cfg), W, max_batch=B, page_size=page, max_pages=P, max_seq_len=512, max_prefill_rows=64,
                    keep_depth_logits=True, device=dev)
    sc = eng.sampling_cfg(greedy=True)
    f64 = lambda a: vr.bf2f(a).astype(np.float64)
    rel = lambda a, b: float(np.sqrt(((f64(a) - f64(b)) ** 2).mean() / (f64(b) ** 2).mean()))
    free, pages, lens, c0_mism, worst, worst_d = list(range(P)), [], [], 0, 0.0, 0.0
    for r in range(B):
        ids, masks = g[f"r{r}_ids"], g[f"r{r}_masks"]
        n = len(ids)
        pg = [free.pop(0) for _ in range((n + page - 1) // page)]
        eng.row_ids[:n], eng.row_masks[:n] = torch.from_numpy(ids).to(dev), torch.from_numpy(masks).to(dev)
        eng.upload_plan(pos=np.arange(n), kvlen=np.arange(1, n + 1), page=[pg[t // page] for t in range(n)],
                        slot=[t % page for t in range(n)], q_req=np.zeros(n), last_rows=[n - 1], indptr=[0, len(pg)], indices=pg)
        eng.prefill(n, 1, n, sc, feedback=False)
        torch.cuda.synchronize()
        worst = max(worst, rel(vr.from_torch(eng.out_logits[:1]), g[f"r{r}_prefill_logits"]))
        c0_mism += int(eng.out_ids[0, 0].item() != g[f"r{r}_frame0"][0])
        pages.append(pg)
        lens.append(n)
    for f in range(NF):
        eng.input_ids[:B] = torch.from_numpy(g[f"f{f}_in_ids"]).to(dev)          # teacher forcing
        eng.input_masks[:B] = torch.from_numpy(g[f"f{f}_in_masks"]).to(dev)
        lens = [n + 1 for n in lens]
        for r in range(B):
            if lens[r] > len(pages[r]) * page:
                pages[r].append(free.pop(0))
        eng.upload_plan(pos=g[f"f{f}_pos"], kvlen=lens, page=[pages[r][(lens[r] - 1) // page] for r in range(B)],
                        slot=[(lens[r] - 1) % page for r in range(B)], indptr=np.cumsum([0] + [len(p_) for p_ in pages]), indices=sum(pages, []))
        eng.frame(B, max(lens), sc, feedback=False)
        torch.cuda.synchronize()
        worst = max(worst, rel(vr.from_torch(eng.out_logits[:B]), g[f"f{f}_logits"]))
        got = eng.out_ids[:B].cpu().numpy()
        ok = got[:, 0] == g[f"f{f}_tokens"][:, 0]
        c0_mism += int((~ok).sum())
        d1 = vr.from_torch(eng.out_depth_logits[0, :B])[ok][:, :256]            # depth step 1: a function of the backbone output and codebook 0
        worst_d = max(worst_d, rel(d1, g[f"f{f}_dlogits"][0][ok]))
    eng.close()
    assert worst <= 1.0e-2 and worst_d <= 1.0e-2, (worst, worst_d)
    assert c0_mism <= 3, c0_mism


# ---- heavy cases: the oracle side is recorded ahead of time (tests/oracle_tape.py, tests/golden/make_oracle_tapes.py) ----------
TAPED = {}


def taped(case):
    def deco(fn):
        TAPED[case] = fn
        return fn
    return deco


def csm_full_width_cfg():
    cfg = CR.CSMCfg(max_pos=512)
    cfg.backbone.layers, cfg.depth.layers, cfg.text_vocab = 2, 2, 4096
    return cfg


@taped("csm_full_width_two_requests_top_k")
def case_csm_full_width_two_layers(tape, dev):
    cfg = csm_full_width_cfg()
    W = Weights(lambda device=None: CR.random_csm_state_dict(cfg, 3, 0.02, device=device))
    run_parity(dev, cfg, W, [(4, 2), (3, 0)], 3, page=128, max_pages=8, sampler_kw=dict(top_k=50, temperature=0.9), tape=tape)


@taped("csm_full_width_b16")
def case_csm_full_width_b16(tape, dev):
    cfg = csm_full_width_cfg()
    W = Weights(lambda device=None: CR.random_csm_state_dict(cfg, 5, 0.02, device=device))
    run_parity(dev, cfg, W, [(2 + i % 5, i % 3) for i in range(16)], 2, page=128, max_pages=32, tape=tape)


def csm_full_depth_cfg():
    cfg = CR.CSMCfg(max_pos=512)          # all 16 backbone + 4 depth layers of CSM-1B at full width
    cfg.text_vocab = 4096
    return cfg


@taped("csm_full_depth_b16")
def case_csm_full_depth_b16(tape, dev):
    cfg = csm_full_depth_cfg()
    W = Weights(lambda device=None: CR.random_csm_state_dict(cfg, 9, 0.02, device=device))
    run_parity(dev, cfg, W, [(2 + i % 4, i % 3) for i in range(16)], 1, page=128, max_pages=32, tape=tape)


@pytest.mark.slow
def test_csm_full_depth_b16(dev):
    """The whole CSM-1B stack — 16 backbone layers + the 4-layer depth decoder over all 31 depth steps — at BASELINE config 3's
    batch size (16 requests): prefills and a free-running frame bit-exact against the oracle's recorded run (the 2 + 2-layer cases
    above pin the layer shapes; this one the depth of the real model: 16 x 4 + 31 x 4 x 4 dependent linears per frame)."""
    case_csm_full_depth_b16(Tape.open("csm_full_depth_b16"), dev)


@pytest.mark.slow
def test_csm_full_width_two_layers(dev):
    """CSM-1B layer shapes (2048 hidden, 32/8 heads of 64, FFN 8192; depth 1024, 8/2 heads of 128, FFN 8192, vocab 2051,
    32 codebooks -> 31 depth steps), 2 backbone + 2 depth layers: bit-exact decode."""
    case_csm_full_width_two_layers(Tape.open("csm_full_width_two_requests_top_k"), dev)


@pytest.mark.slow
def test_csm_full_width_b16_mfma_batch(dev):
    """BASELINE config 3's batch size: 16 concurrent requests at CSM-1B layer shapes (2 backbone + 2 depth layers, all 32
    codebooks -> 31 depth steps): every linear of the frame runs on the matrix cores (32-row depth step 1 included);
    prefills and two free-running frames bit-exact against the oracle."""
    case_csm_full_width_b16(Tape.open("csm_full_width_b16"), dev)
