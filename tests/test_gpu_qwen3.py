"""End-to-end parity of the Qwen3-TTS frame engine (libvoxhip vox_qwen3_* through the C ABI) against the CPU
oracle: ragged prefill, then free-running batched decode under greedy AND seeded top-k sampling.

Bar: BIT-EXACT token ids, logits, hidden states and KV cache contents, free-running over many frames, at every batch
size.  Calls of <= 2 rows (`exact_rows`, settable 1..8) run the wave64 VALU kernels (canonical order of oracle/voxref.c); calls with more rows
(longer prompts, batches > 8) run on the matrix cores, whose accumulation arithmetic the oracle restates bit for bit
(voxref.c: vr_mfma_step8, measured on the MI355X with tools/mfma_probe) together with each GEMM kernel's K split
(oracle/policy.py).
"""
from tests.conftest import bf16_close
import dataclasses

import numpy as np
import pytest
import torch

from oracle import qwen3_ref as QR
from oracle import voxref as vr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda")


def to_engine_cfg(c: QR.Qwen3Cfg):
    from vox_serve_amd.engine import Qwen3Cfg, StackCfg
    conv = lambda s: StackCfg(s.hidden, s.layers, s.heads, s.kv_heads, s.head_dim, s.ffn, s.eps, s.rope_theta)
    return Qwen3Cfg(conv(c.talker), conv(c.depth), c.vocab, c.text_vocab, c.text_hidden, c.depth_vocab, c.n_groups,
                    c.eos_id, c.tts_pad_id, c.max_pos)


def make_prompt(rng, cfg, n):
    ids = np.zeros((n, cfg.n_groups + 1), np.int32)
    ids[:, -1] = rng.integers(0, cfg.text_vocab, n)
    ids[:, 0] = rng.integers(0, cfg.vocab - 1024, n)
    masks = np.zeros(n, np.uint8)
    masks[n // 2:] = 1
    feats = vr.f2bf((0.05 * rng.standard_normal((n, cfg.talker.hidden))).astype(np.float32))
    feats[: n // 3] = 0
    return ids, masks, feats


def run_parity(dev, cfg, W, prompt_lens, n_frames, page, sampler_kw=None, max_pages=64, policy=None):
    from vox_serve_amd.engine import Qwen3Engine
    rng = np.random.default_rng(3)
    B = len(prompt_lens)
    ref = QR.Qwen3Ref(cfg, W, page_size=page, max_pages=max_pages, max_batch=B, policy=policy)
    Wt = {k: vr.to_torch(v).to(dev) for k, v in W.items()}
    eng = Qwen3Engine(to_engine_cfg(cfg), Wt, max_batch=B, page_size=page, max_pages=max_pages, max_seq_len=512,
                      max_prefill_rows=128, keep_depth_logits=True, device=dev)
    seed = 1234
    if sampler_kw:
        sc = eng.sampling_cfg(greedy=False, **sampler_kw)
        frame_no = [0]

        def sampler(logits, i):
            return vr.sample(logits, seed=seed, offset=frame_no[0] * cfg.n_groups + i, **sampler_kw)
    else:
        sc, sampler, frame_no = eng.sampling_cfg(greedy=True), None, [0]

    G1 = cfg.n_groups + 1
    reqs = []
    state_ids = torch.zeros(B, G1, dtype=torch.int32, device=dev)
    state_feat = torch.zeros(B, cfg.talker.hidden, dtype=torch.bfloat16, device=dev)
    for r, n in enumerate(prompt_lens):
        ids, masks, feats = make_prompt(rng, cfg, n)
        req = QR.RefRequest()
        lg, hid = ref.prefill(req, ids, masks, feats)
        out, masked, _, dl = ref.frame([req], lg, hid, sampler)
        # engine: stage rows + plan, prefill, compare
        eng.row_ids[:n] = torch.from_numpy(ids).to(dev)
        eng.row_masks[:n] = torch.from_numpy(masks).to(dev)
        eng.row_feats[:n] = vr.to_torch(feats).to(dev)
        eng.upload_plan(pos=np.arange(n), kvlen=np.arange(1, n + 1), page=[req.kv_pages[t // page] for t in range(n)],
                        slot=[t % page for t in range(n)], q_req=np.zeros(n), last_rows=[n - 1],
                        indptr=[0, len(req.kv_pages)], indices=req.kv_pages)
        eng.rng_offset.fill_(frame_no[0])
        eng.prefill(n, 1, n, sc, seed=seed, feedback=True)
        torch.cuda.synchronize()
        assert np.array_equal(vr.from_torch(eng.out_hidden[:1]), hid), f"prefill hidden r{r}"
        assert np.array_equal(vr.from_torch(eng.out_logits[:1]), masked), f"prefill logits r{r}"
        assert np.array_equal(vr.from_torch(eng.out_depth_logits[:, 0]), np.stack(dl)[:, 0]), f"prefill depth r{r}"
        assert np.array_equal(eng.out_ids[:1].cpu().numpy(), out), f"prefill tokens r{r}"
        state_ids[r] = eng.input_ids[0]
        state_feat[r] = eng.input_features[0]
        reqs.append(req)
    frame_no[0] = 1
    # batched free-running decode from the fed-back state
    eng.input_ids[:B] = state_ids
    eng.input_masks[:B] = 1
    eng.input_features[:B] = state_feat
    eng.rng_offset.fill_(frame_no[0])
    for f in range(n_frames):
        lg, hid = ref.decode(reqs)
        out, masked, _, dl = ref.frame(reqs, lg, hid, sampler)
        indptr, indices = [0], []
        for q in reqs:
            indptr.append(indptr[-1] + len(q.kv_pages))
            indices += q.kv_pages
        eng.upload_plan(pos=[q.next_position_id - 1 for q in reqs], kvlen=[q.kv_token_len for q in reqs],
                        page=[q.kv_pages[-1] for q in reqs], slot=[q.kv_last_page_len - 1 for q in reqs],
                        indptr=indptr, indices=indices)
        eng.frame(B, max(q.kv_token_len for q in reqs), sc, seed=seed, feedback=True, use_graph=True)
        torch.cuda.synchronize()
        assert np.array_equal(vr.from_torch(eng.out_hidden[:B]), hid), f"hidden f{f}"
        assert np.array_equal(vr.from_torch(eng.out_logits[:B]), masked), f"logits f{f}"
        assert np.array_equal(vr.from_torch(eng.out_depth_logits[:, :B]), np.stack(dl)), f"depth logits f{f}"
        assert np.array_equal(eng.out_ids[:B].cpu().numpy(), out), f"tokens f{f}"
        assert np.array_equal(vr.from_torch(eng.input_features[:B]),
                              np.concatenate([q.input_features for q in reqs])), f"features f{f}"
        frame_no[0] += 1
    kv_ref = np.stack(ref.kv)
    assert np.array_equal(vr.from_torch(eng.kv), kv_ref), "KV cache"
    eng.close()


def test_tiny_greedy_b1(dev):
    cfg = QR.tiny_cfg()
    run_parity(dev, cfg, QR.random_weights(cfg, 0, 0.08), [20], 6, page=16)


def test_tiny_greedy_b3_ragged_pages(dev):
    cfg = QR.tiny_cfg()
    run_parity(dev, cfg, QR.random_weights(cfg, 1, 0.08), [13, 33, 7], 40, page=16)   # crosses page + chunk edges


def test_tiny_topk_sampling_b2(dev):
    cfg = QR.tiny_cfg()
    run_parity(dev, cfg, QR.random_weights(cfg, 2, 0.08), [11, 18], 8, page=16,
               sampler_kw=dict(top_k=50, top_p=1.0, temperature=0.9))


def test_tiny_short_prompts_bit_exact_through_prefill(dev):
    cfg = QR.tiny_cfg()
    run_parity(dev, cfg, QR.random_weights(cfg, 5, 0.08), [8, 3, 5, 1], 12, page=16)


def test_tiny_b8_batch(dev):
    cfg = QR.tiny_cfg()
    run_parity(dev, cfg, QR.random_weights(cfg, 4, 0.08), [9, 5, 12, 6, 8, 10, 7, 11], 3, page=16, max_pages=96)


def test_tiny_b12_mfma_batch(dev):
    """12 rows: every linear of the frame runs on the matrix cores; free-running for 40 frames, bit-exact."""
    cfg = QR.tiny_cfg()
    run_parity(dev, cfg, QR.random_weights(cfg, 6, 0.08), [9, 5, 12, 6, 8, 10, 7, 11, 13, 4, 15, 6], 40, page=16, max_pages=128)


def test_tiny_b32_mfma_batch_100_frames(dev):
    """BASELINE config 2's batch size on the tiny config: 32 requests (prompts of 3..40 tokens, MFMA prefills included),
    100 free-running frames with no re-synchronisation — every token of every codebook equals the oracle's."""
    cfg = QR.tiny_cfg()
    lens = [3 + (7 * i) % 38 for i in range(32)]
    run_parity(dev, cfg, QR.random_weights(cfg, 8, 0.08), lens, 100, page=16, max_pages=32 * 10)


def test_tiny_b20_topk_sampling_mfma(dev):
    cfg = QR.tiny_cfg()
    run_parity(dev, cfg, QR.random_weights(cfg, 9, 0.08), [5 + i for i in range(20)], 10, page=16, max_pages=160,
               sampler_kw=dict(top_k=50, top_p=1.0, temperature=0.9))


def test_exact_rows_8_setting_bit_exact_under_its_own_policy(dev):
    """`vox_ctx_set_exact_rows(8)` (bench.py --exact-rows 8): calls of up to 8 rows stay on the wave64 VALU kernels instead of
    the matrix cores.  The oracle follows the same policy — tiny config, 8 requests, 20 free-running frames, bit-exact
    (the default setting, 2, runs in every other test of this file)."""
    from oracle.policy import Policy
    from vox_serve_amd import _native as N
    cfg = QR.tiny_cfg()
    N.set_exact_rows(8)
    try:
        run_parity(dev, cfg, QR.random_weights(cfg, 4, 0.08), [9, 5, 12, 6, 8, 10, 7, 11], 20, page=16, max_pages=96,
                   policy=Policy(exact_rows=8))
    finally:
        N.set_exact_rows(2)


_FULL = {}


def full_size_cfg_and_weights():
    """Qwen3-TTS-1.7B shapes with random weights, built once per session (1.7 G normals take a minute of host time)."""
    if not _FULL:
        cfg = QR.Qwen3Cfg(text_vocab=4096, tts_pad_id=4095, max_pos=1024)   # text table shrunk (gathered, not streamed)
        _FULL["cfg"], _FULL["W"] = cfg, QR.random_weights(cfg, 0, 0.02)
    return _FULL["cfg"], _FULL["W"]


def test_full_size_qwen3_1p7b_one_frame(dev):
    """Qwen3-TTS-1.7B shapes (28+5 layers, random weights): 12-token prefill (matrix-core path through every layer) + 1 decode frame,
    greedy."""
    cfg, W = full_size_cfg_and_weights()
    run_parity(dev, cfg, W, [12], 1, page=128, max_pages=8)


def test_full_size_b32_rows_independent_and_close_to_the_exact_path(dev):
    """BASELINE config 2 shapes (Qwen3-TTS-1.7B, 32 concurrent requests) through the batched path (full-K MFMA GEMMs on
    fragment-major weights, 64-row depth step 1, chunked talker attention): size-independent properties —
      (1) determinism: the same batch twice gives identical ids and logits;
      (2) row independence: a request's outputs do not depend on which rows its neighbours occupy (batch permuted);
      (3) the batched logits sit within bf16 rounding of the 1-row fixed-order path (the one pinned bit-exactly to the oracle)."""
    from vox_serve_amd.engine import Qwen3Cfg, Qwen3Engine
    from vox_serve_amd.synth import synth_qwen3_weights
    cfg = Qwen3Cfg()
    W = synth_qwen3_weights(cfg, dev, seed=0)
    B, ps, kv0, frames = 32, 128, 40, 3
    g = torch.Generator(device="cpu").manual_seed(5)
    kv_req = (torch.randn(B, cfg.talker.layers, 2, kv0, cfg.talker.kv_heads, cfg.talker.head_dim, generator=g) * 0.5).to(torch.bfloat16)
    ids0 = torch.randint(0, cfg.vocab, (B, cfg.n_groups + 1), generator=g, dtype=torch.int32)
    ids0[:, -1] = cfg.tts_pad_id
    feats0 = (torch.randn(B, cfg.talker.hidden, generator=g) * 0.05).to(torch.bfloat16)

    def run(order, max_batch):
        """order[r] = request placed in row r"""
        n = len(order)
        eng = Qwen3Engine(cfg, W, max_batch=max_batch, page_size=ps, max_pages=max(8, 2 * n), max_seq_len=512, max_prefill_rows=16)
        eng.keep_hidden = False
        for r, q in enumerate(order):
            eng.kv[:, r, :, :kv0] = kv_req[q].to(dev)
            eng.input_ids[r] = ids0[q].to(dev)
            eng.input_features[r] = feats0[q].to(dev)
        eng.input_masks[:n] = 0
        sc = eng.sampling_cfg(greedy=True)
        out_ids, out_logits = [], []
        for f in range(frames):
            L = kv0 + 1 + f
            eng.upload_plan(pos=[L] * n, kvlen=[L] * n, page=list(range(n)), slot=[L - 1] * n, indptr=list(range(n + 1)), indices=list(range(n)))
            eng.frame(n, L, sc, feedback=True)
            torch.cuda.synchronize()
            out_ids.append(eng.out_ids[:n].cpu().clone())
            out_logits.append(eng.out_logits[:n].float().cpu().clone())
        del eng
        return torch.stack(out_ids), torch.stack(out_logits)

    ident = list(range(B))
    ids_a, lg_a = run(ident, B)
    ids_b, lg_b = run(ident, B)
    assert torch.equal(ids_a, ids_b) and torch.equal(lg_a, lg_b)                      # (1)
    perm = torch.randperm(B, generator=g).tolist()
    ids_p, lg_p = run(perm, B)
    inv = [perm.index(q) for q in range(B)]                                            # row of request q in the permuted batch
    assert torch.equal(ids_p[:, inv], ids_a) and torch.equal(lg_p[:, inv], lg_a)      # (2)
    for q in (0, 13, 31):                                                              # (3) frame 0: same inputs on both paths
        ids_1, lg_1 = run([q], 1)
        a, e = lg_a[0, q].double(), lg_1[0, 0].double()
        assert float(((a - e) ** 2).mean().sqrt() / (e ** 2).mean().sqrt()) <= 2e-2


def test_full_size_b32_free_running_bit_exact_vs_oracle(dev):
    """BASELINE config 2 at the 1.7B layer shapes, 32 concurrent requests: every linear of the frame runs on the matrix cores (full-K
    GEMMs on fragment-major weights, 64-row depth step 1, 4-wave GEMM for codec_head and the text projection).  One whole frame from an
    injected 40-token KV state: hidden states, masked logits, all 15 depth logits, all 16 codebook ids, the fed-back next-frame features
    and the K/V cache equal the oracle's bit for bit.  The stacks are cut to 6 of the 28 talker layers
    and 2 of the 5 depth layers (the 15-step depth loop is kept): kernel selection depends on the layer shapes and the row count, not on the
    number of layers, and the oracle's restatement of the matrix cores' arithmetic costs minutes of host time per full-depth frame on
    the GPU box's 16-CPU share.  The full 28 + 5 stack is pinned by the one-request test above (12-row MFMA prefill through all
    layers) and, at 32 requests, by the property test above; the 100-frame free run at 32 requests is the tiny-config test."""
    from vox_serve_amd.engine import Qwen3Engine
    cfg = QR.Qwen3Cfg(text_vocab=4096, tts_pad_id=4095, max_pos=1024)
    cfg.talker, cfg.depth = dataclasses.replace(cfg.talker, layers=6), dataclasses.replace(cfg.depth, layers=2)
    W = QR.random_weights(cfg, 0, 0.02)
    B, ps, kv0, frames = 32, 128, 40, 1
    rng = np.random.default_rng(11)
    t = cfg.talker
    ref = QR.Qwen3Ref(cfg, W, page_size=ps, max_pages=B, max_batch=B)
    eng = Qwen3Engine(to_engine_cfg(cfg), {k: vr.to_torch(v).to(dev) for k, v in W.items()}, max_batch=B, page_size=ps,
                      max_pages=B, max_seq_len=512, max_prefill_rows=32, keep_depth_logits=True, device=dev)
    kv0_bits = vr.f2bf((0.5 * rng.standard_normal((t.layers, B, 2, kv0, t.kv_heads, t.head_dim))).astype(np.float32))
    for l in range(t.layers):
        ref.kv[l][:, :, :kv0] = kv0_bits[l]
    eng.kv[:, :, :, :kv0] = vr.to_torch(kv0_bits).to(dev)
    ids0 = np.zeros((B, cfg.n_groups + 1), np.int32)
    ids0[:, 0] = rng.integers(0, cfg.vocab - 1024, B)
    ids0[:, -1] = cfg.tts_pad_id
    feats0 = vr.f2bf((0.05 * rng.standard_normal((B, t.hidden))).astype(np.float32))
    reqs = []
    for b in range(B):
        ref.free_pages.remove(b)
        reqs.append(QR.RefRequest(kv_pages=[b], kv_token_len=kv0, kv_last_page_len=kv0, next_position_id=kv0 + 1,
                                  input_ids=ids0[b:b + 1].copy(), input_mask=True, input_features=feats0[b:b + 1].copy()))
    eng.input_ids[:B] = torch.from_numpy(ids0).to(dev)
    eng.input_masks[:B] = 1
    eng.input_features[:B] = vr.to_torch(feats0).to(dev)
    sc = eng.sampling_cfg(greedy=True)
    for f in range(frames):
        lg, hid = ref.decode(reqs)
        out, masked, _, dl = ref.frame(reqs, lg, hid, None)
        eng.upload_plan(pos=[q.next_position_id - 1 for q in reqs], kvlen=[q.kv_token_len for q in reqs],
                        page=[q.kv_pages[-1] for q in reqs], slot=[q.kv_last_page_len - 1 for q in reqs],
                        indptr=list(range(B + 1)), indices=list(range(B)))
        eng.frame(B, max(q.kv_token_len for q in reqs), sc, feedback=True)
        torch.cuda.synchronize()
        assert np.array_equal(vr.from_torch(eng.out_hidden[:B]), hid), f"hidden f{f}"
        assert np.array_equal(vr.from_torch(eng.out_logits[:B]), masked), f"logits f{f}"
        assert np.array_equal(vr.from_torch(eng.out_depth_logits[:, :B]), np.stack(dl)), f"depth logits f{f}"
        assert np.array_equal(eng.out_ids[:B].cpu().numpy(), out), f"tokens f{f}"
        assert np.array_equal(vr.from_torch(eng.input_features[:B]), np.concatenate([q.input_features for q in reqs])), f"features f{f}"
    assert np.array_equal(vr.from_torch(eng.kv), np.stack(ref.kv)), "KV cache"
    eng.close()
