"""GPU parity of every libvoxhip op (called through the C ABI via the drop-in Python wrappers) against the
CPU oracle.  Bar: BIT-EXACT bf16 outputs and integer ids — the HIP kernels and oracle/voxref.c follow the
same fixed-order numeric contract.  Also replays the reference-captured goldens on the GPU path.
"""
import numpy as np
import pytest
import torch

from oracle import voxref as vr
from tests.conftest import bf16_close
from oracle.policy import EPI_SILU, EPI_SILU_MUL, EPI_STORE, Call, route
from tests.oracle_tape import Tape

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from vox_serve_amd import _native as N
    N.ctx()
    return torch.device("cuda")


def T(bits, dev):
    return vr.to_torch(bits).to(dev)


def Bits(t):
    return vr.from_torch(t)


def rnd(rng, *shape, s=1.0):
    return vr.f2bf(rng.standard_normal(shape).astype(np.float32) * np.float32(s))


@pytest.mark.parametrize("B,N_,K", [(1, 64, 256), (1, 2048, 2048), (2, 1030, 1024), (3, 512, 6144), (4, 4096, 2048),
                                    (5, 96, 512), (8, 3072, 1024), (9, 128, 2048), (19, 256, 768), (1, 7, 64),
                                    (16, 2048, 2048), (32, 4096, 2048), (32, 2048, 6144), (75, 1000, 1024), (33, 64, 96),
                                    (8, 256, 13696), (7, 64, 13696), (6, 100, 13696), (1, 128, 13696), (2, 256, 13696), (3, 4096, 13696), (12, 2051, 1024), (16, 1000, 2048), (31, 2051, 1024),
                                    # <= 16 rows off the full-K shapes: k_linear_mfma_small / _stream (CosyVoice2's 896 / 4864, a ragged K, 1..3 groups)
                                    (8, 896, 4864), (8, 1152, 896), (16, 896, 896), (9, 6564, 896), (3, 40, 1056), (5, 48, 2080), (4, 64, 3104), (11, 32, 4128), (16, 64, 6176), (7, 128, 96)])
def test_linear_bit_exact(dev, B, N_, K):
    from vox_serve_amd import _native as N
    rng = np.random.default_rng(B * 1000 + N_ + K)
    W, x, bias, res = rnd(rng, N_, K, s=0.05), rnd(rng, B, K), rnd(rng, N_, s=0.1), rnd(rng, B, N_)
    for use_bias, use_res, act in [(False, False, 0), (True, False, 0), (True, True, 0), (True, False, 1)]:
        y = torch.empty(B, N_, dtype=torch.bfloat16, device=dev)
        Wt, xt, bt, rt = T(W, dev), T(x, dev), T(bias, dev), T(res, dev)
        N.check(N.lib().vox_linear(N.ctx(), N.stream(), N.ptr(Wt), N.ptr(bt) if use_bias else None, N.ptr(xt),
                                   N.ptr(rt) if use_res else None, N.ptr(y), B, N_, K, act))
        # bit-exact at every row count: the oracle sums in the order of the kernel the call routes to (oracle/policy.py)
        order = route(Call(B=B, N=N_, K=K, epi=EPI_SILU if act else EPI_STORE))[0]
        ref = vr.linear(W, x, bias if use_bias else None, None, order=order, act=act)
        if use_res:
            ref = vr.add(res, ref)
        assert np.array_equal(Bits(y), ref), (use_bias, use_res, act, order)


@pytest.mark.parametrize("B,N_,K", [(1, 6144, 2048), (2, 768, 256), (4, 3072, 1024), (8, 520, 512), (11, 64, 128),
                                    (32, 6144, 2048), (75, 3072, 1024), (8, 4864, 896), (16, 100, 992)])
def test_linear_silu_mul_bit_exact(dev, B, N_, K):
    from vox_serve_amd import _native as N
    rng = np.random.default_rng(N_ + K + B)
    Wg, Wu, x = rnd(rng, N_, K, s=0.05), rnd(rng, N_, K, s=0.05), rnd(rng, B, K)
    h = torch.empty(B, N_, dtype=torch.bfloat16, device=dev)
    Wgt, Wut, xt = T(Wg, dev), T(Wu, dev), T(x, dev)        # keep alive: ptr() does not own the tensor
    N.check(N.lib().vox_linear_silu_mul(N.ctx(), N.stream(), N.ptr(Wgt), N.ptr(Wut), N.ptr(xt), N.ptr(h), B, N_, K))
    ref = vr.linear_silu_mul(Wg, Wu, x, order=route(Call(B=B, N=N_, K=K, epi=EPI_SILU_MUL))[0])
    assert np.array_equal(Bits(h), ref)


@pytest.mark.parametrize("rows,H", [(1, 2048), (5, 128), (33, 1024), (64, 64), (3, 4096)])
def test_rmsnorm_bit_exact(dev, rows, H):
    from vox_serve_amd.flashinfer_utils import rms_norm
    rng = np.random.default_rng(rows + H)
    x, w = rnd(rng, rows, H, s=2.0), vr.f2bf(1 + 0.1 * rng.standard_normal(H).astype(np.float32))
    assert np.array_equal(Bits(rms_norm(T(x, dev), T(w, dev), 1e-6)), vr.rmsnorm(x, w, 1e-6))


def test_rmsnorm_reference_golden(dev, golden):
    from vox_serve_amd.flashinfer_utils import rms_norm
    g = golden("g2_wrappers")
    assert bf16_close(Bits(rms_norm(T(g["rms_x"], dev), T(g["rms_w"], dev), 1e-6)), g["rms_y"], ulps=1).all()


@pytest.mark.parametrize("tag,kw", [
    ("neox", dict(rope_theta=1e6)),
    ("glm", dict(rope_theta=1e4, rotary_dim=32, interleave=True)),
    ("l31", dict(rope_theta=5e5, rope_scale=32.0, low_freq_factor=1.0, high_freq_factor=4.0, old_context_len=8192)),
])
def test_rope_bit_exact_and_golden(dev, golden, tag, kw):
    from vox_serve_amd.flashinfer_utils import apply_rope_pos_ids
    g = golden("g2_wrappers")
    q, k = apply_rope_pos_ids(T(g["rope_q"], dev), T(g["rope_k"], dev), torch.from_numpy(g["rope_pos"]).to(dev), **kw)
    rot = kw.get("rotary_dim", 64)
    l31 = (1.0, 4.0, 8192) if "low_freq_factor" in kw else None
    cs = vr.rope_table(8192, rot, kw["rope_theta"], kw.get("rope_scale", 1.0), l31)
    qo = vr.rope(g["rope_q"], g["rope_pos"], cs, rot, kw.get("interleave", False))
    ko = vr.rope(g["rope_k"], g["rope_pos"], cs, rot, kw.get("interleave", False))
    assert np.array_equal(Bits(q), qo) and np.array_equal(Bits(k), ko)          # vs oracle: bit-exact
    assert bf16_close(Bits(q), g[f"rope_{tag}_q"], ulps=1, atol=1e-3).all()      # vs reference capture


def test_reference_wrappers_golden_on_gpu(dev, golden):
    """The reference's own plan()/set_kv_cache()/run() sequence, replayed through the drop-in wrappers."""
    from vox_serve_amd.flashinfer_utils import FlashInferDecodeWrapper, FlashInferPrefillWrapper
    g = golden("g2_wrappers")
    page, Hq, Hkv, D = int(g["page"]), 4, 2, 16
    w = FlashInferPrefillWrapper(None, Hq, Hkv, Hq * D, page, device=dev)
    i32 = lambda a: torch.from_numpy(a.astype(np.int32))
    w.plan(i32(g["pf_qo"]), i32(g["pf_indptr"]), i32(g["pf_indices"]), i32(g["pf_last"]), torch.bfloat16)
    assert np.array_equal(w.token_to_page.cpu().numpy(), g["pf_token_to_page"])
    assert np.array_equal(w.token_to_cache.cpu().numpy(), g["pf_token_to_cache"])
    kv = T(g["pf_kv_in"], dev)
    w.set_kv_cache(kv, T(g["pf_k"], dev), T(g["pf_v"], dev))
    assert np.array_equal(Bits(kv), g["pf_kv_out"])
    out = w.run(T(g["pf_q"], dev), kv)
    assert bf16_close(Bits(out), g["pf_out"], ulps=1).all()
    d = FlashInferDecodeWrapper(None, Hq, Hkv, Hq * D, page, device=dev)
    d.plan(i32(g["pf_indptr"]), i32(g["pf_indices"]), i32(g["pf_last"]), torch.bfloat16)
    assert np.array_equal(d.kv_cache_locations.cpu().numpy(), g["dc_loc"])
    d.set_kv_cache(kv, T(g["dc_k"], dev), T(g["dc_v"], dev))
    assert np.array_equal(Bits(kv), g["dc_kv_out"])
    out = d.run(T(g["dc_q"], dev), kv)
    assert bf16_close(Bits(out), g["dc_out"], ulps=1).all()


@pytest.mark.parametrize("Hq,Hkv,D,page,lens", [
    (16, 8, 128, 128, [1, 31, 32, 33, 200, 517]),
    (32, 8, 64, 16, [5, 64, 100]),
    (32, 2, 128, 128, [77, 300]),
    (14, 2, 64, 8, [9, 130]),
    (4, 2, 16, 4, [3, 40]),
    (16, 8, 128, 128, [200, 33, 256, 97, 129, 64, 255, 161] * 4),      # 32 rows x 8 chunks x 8 kv heads: paired-chunk blocks
    (16, 8, 128, 128, [2000, 1, 1300, 257]),                            # the long-context merge (k_attn_merge_row: up to 63 chunks)
    (14, 2, 64, 16, [700, 290]),
])
def test_paged_decode_attention_bit_exact(dev, Hq, Hkv, D, page, lens):
    from vox_serve_amd.flashinfer_utils import FlashInferDecodeWrapper
    rng = np.random.default_rng(Hq * D + page)
    B = len(lens)
    npages = [(n + page - 1) // page for n in lens]
    P = sum(npages) + 3
    perm = rng.permutation(P)
    indptr = np.concatenate([[0], np.cumsum(npages)]).astype(np.int32)
    indices = perm[: indptr[-1]].astype(np.int32)
    last = np.array([n % page or page for n in lens], np.int32)
    kv = rnd(rng, P, 2, page, Hkv, D)
    q = rnd(rng, B, Hq, D)
    w = FlashInferDecodeWrapper(None, Hq, Hkv, Hq * D, page, device=dev)
    w.plan(torch.from_numpy(indptr), torch.from_numpy(indices), torch.from_numpy(last), torch.bfloat16)
    out = w.run(T(q, dev), T(kv, dev))
    ref = vr.paged_attention(q, kv, np.arange(B), np.array(lens), indptr, indices)
    assert np.array_equal(Bits(out), ref)


def test_paged_prefill_attention_bit_exact(dev):
    from vox_serve_amd.flashinfer_utils import FlashInferPrefillWrapper
    rng = np.random.default_rng(5)
    Hq, Hkv, D, page = 16, 8, 128, 16
    q_lens, kv_lens = [40, 1, 75], [40, 90, 75]
    npages = [(n + page - 1) // page for n in kv_lens]
    P = sum(npages) + 2
    indptr = np.concatenate([[0], np.cumsum(npages)]).astype(np.int32)
    indices = rng.permutation(P)[: indptr[-1]].astype(np.int32)
    last = np.array([n % page or page for n in kv_lens], np.int32)
    qo = np.concatenate([[0], np.cumsum(q_lens)]).astype(np.int32)
    kv = rnd(rng, P, 2, page, Hkv, D)
    Tn = int(qo[-1])
    q, k, v = rnd(rng, Tn, Hq, D), rnd(rng, Tn, Hkv, D), rnd(rng, Tn, Hkv, D)
    w = FlashInferPrefillWrapper(None, Hq, Hkv, Hq * D, page, device=dev)
    w.plan(torch.from_numpy(qo), torch.from_numpy(indptr), torch.from_numpy(indices), torch.from_numpy(last), torch.bfloat16)
    kvt = T(kv, dev)
    w.set_kv_cache(kvt, T(k, dev), T(v, dev))
    out = w.run(T(q, dev), kvt)
    tp, tc = w.token_to_page.cpu().numpy().astype(np.int32), w.token_to_cache.cpu().numpy().astype(np.int32)
    kv_ref = kv.copy()
    vr.kv_append(kv_ref, k, v, tp, tc)
    assert np.array_equal(Bits(kvt), kv_ref)
    q_req = np.repeat(np.arange(3), q_lens)
    q_kvlen = np.concatenate([np.arange(n - m + 1, n + 1) for m, n in zip(q_lens, kv_lens)])
    ref = vr.paged_attention(q, kv_ref, q_req, q_kvlen, indptr, indices)
    assert np.array_equal(Bits(out), ref)


# ---------------------------------------------------------------- sampler -------------------------
@pytest.mark.parametrize("name", ["a", "b", "c", "d", "e", "tie"])
def test_greedy_goldens(dev, golden, name):
    from vox_serve_amd.sampling import Sampler, SamplingConfig
    g = golden("g1_sampler")
    ids = Sampler.run_sampling(T(g[f"greedy_{name}_logits"], dev), SamplingConfig(greedy=True))
    assert np.array_equal(ids.cpu().numpy().astype(np.int32), g[f"greedy_{name}_ids"])


def test_module_level_sampler_helpers(dev, golden):
    """sampling.py:21-80 of the reference: greedy_sampling / top_k_sampling / top_p_sampling / top_k_top_p_sampling / min_p_sampling as
    module-level functions with the reference's arguments.  greedy replays the reference's own fixture (first maximum, any leading
    shape); the stochastic ones equal the oracle's draw for the same (seed, offset) — the contract Sampler.run_sampling is tested on."""
    from vox_serve_amd import sampling as S
    g = golden("g1_sampler")
    for name in ("a", "tie"):
        lg = T(g[f"greedy_{name}_logits"], dev)
        ids = S.greedy_sampling(lg)
        assert ids.dtype == torch.int64 and np.array_equal(ids.cpu().numpy().astype(np.int32), g[f"greedy_{name}_ids"])
        assert torch.equal(S.greedy_sampling(lg[None]), ids[None])            # argmax(dim=-1) keeps the leading dims
    rng = np.random.default_rng(11)
    logits = vr.f2bf(rng.standard_normal((4, 2048)).astype(np.float32) * 3)
    lt = T(logits, dev)
    cases = [(lambda: S.top_k_sampling(lt, 50, 0.9), dict(top_k=50, top_p=1.0, min_p=0.0, temperature=0.9)),
             (lambda: S.top_p_sampling(lt, 0.8, 0.7), dict(top_k=0, top_p=0.8, min_p=0.0, temperature=0.7)),
             (lambda: S.top_k_top_p_sampling(lt, 25, 0.8, 1.0), dict(top_k=25, top_p=0.8, min_p=0.0, temperature=1.0)),
             (lambda: S.min_p_sampling(lt, 0.1, 1.0), dict(top_k=0, top_p=1.0, min_p=0.1, temperature=1.0))]
    S.Sampler.manual_seed(123)
    for off, (fn, kw) in enumerate(cases):
        got = fn()
        assert got.dtype == torch.int32 and got.shape == (4,)
        assert np.array_equal(got.cpu().numpy(), vr.sample(logits, seed=123, offset=off, **kw)), kw
    with pytest.raises(NotImplementedError):
        S.top_k_top_p_sampling(lt, 25, 0.8, 1.0, filter_apply_order="joint")


def test_repetition_penalty_and_update_goldens(dev, golden):
    from vox_serve_amd.sampling import Sampler
    g = golden("g1_sampler")
    out = Sampler.apply_repetition_penalty(T(g["pen_logits"], dev), torch.from_numpy(g["pen_cache"]).to(dev).bool(), 1.3)
    assert np.array_equal(Bits(out), g["pen_out"])
    for tag, window in (("glob", -1), ("win", 3)):
        c = torch.from_numpy(g[f"upd_{tag}_in"]).to(dev).bool()
        Sampler.update_repetition_penalty_cache(c, torch.from_numpy(g[f"upd_{tag}_ids"]).to(dev)[:, None], window)
        assert np.array_equal(c.cpu().numpy().astype(np.uint8), g[f"upd_{tag}_out"])


def test_multi_codebook_repetition_penalty_and_update(dev, golden):
    """logits [B, C, V] / output_ids [B, C] with C > 1 (sampling.py:122-178): the native kernels equal the reference's outputs (g19)
    and the oracle on a second, larger random case; mismatched codebook counts are rejected."""
    from oracle import sampler_mc_ref as MR
    from vox_serve_amd.sampling import Sampler
    g = golden("g19_sampler_mc")
    out = Sampler.apply_repetition_penalty(T(g["pen_logits"], dev), torch.from_numpy(g["pen_cache"]).to(dev).bool(), 1.3)
    assert np.array_equal(Bits(out), g["pen_out"])
    for tag, window in (("glob", -1), ("win", 3)):
        c = torch.from_numpy(g[f"upd_{tag}_in"]).to(dev).bool()
        Sampler.update_repetition_penalty_cache(c, torch.from_numpy(g[f"upd_{tag}_ids"]).to(dev), window)
        assert np.array_equal(c.cpu().numpy().astype(np.uint8), g[f"upd_{tag}_out"]), tag
    rng = np.random.default_rng(5)
    B, W, C, V = 5, 3, 8, 2051
    lg = vr.f2bf(rng.standard_normal((B, C, V)).astype(np.float32) * 2)
    cache = (rng.random((B, W, C, V)) < 0.05).astype(np.uint8)
    out = Sampler.apply_repetition_penalty(T(lg, dev), torch.from_numpy(cache).to(dev).bool(), 1.1)
    assert np.array_equal(Bits(out), MR.rep_penalty_mc(lg, cache, 1.1))
    ids = rng.integers(0, V, (B, C)).astype(np.int64)
    for window in (-1, W):
        ct, want = torch.from_numpy(cache).to(dev).bool(), cache.copy()
        Sampler.update_repetition_penalty_cache(ct, torch.from_numpy(ids).to(dev), window)
        MR.rep_update_mc(want, ids, window)
        assert np.array_equal(ct.cpu().numpy().astype(np.uint8), want), window
    with pytest.raises(ValueError):
        Sampler.apply_repetition_penalty(T(lg[:, :3], dev), torch.from_numpy(cache).to(dev).bool(), 1.1)


@pytest.mark.parametrize("B,V,k,p,mp,T_", [(4, 3072, 50, 1.0, 0.0, 0.9), (8, 2048, 50, 1.0, 0.0, 0.9),
                                          (3, 2051, 25, 0.8, 0.0, 1.0), (2, 6564, 25, 1.0, 0.1, 0.7),
                                          (5, 512, 200, 0.95, 0.0, 1.3), (2, 300, 256, 1.0, 0.0, 1.0)])
def test_stochastic_sampler_bit_exact(dev, B, V, k, p, mp, T_):
    from vox_serve_amd import _native as N
    rng = np.random.default_rng(V + k)
    logits = vr.f2bf(rng.standard_normal((B, V)).astype(np.float32) * 3)
    logits[0, : V // 2] = logits[0, 0]            # heavy ties: exercises ordered tie selection
    lt = T(logits, dev)
    out = torch.empty(B, dtype=torch.int32, device=dev)
    for off in range(12):
        cfg = N.SamplingCfg(0, k, p, mp, T_, 1.0)
        N.check(N.lib().vox_sample(N.ctx(), N.stream(), N.ptr(lt), B, V, cfg, 77, off, N.ptr(out)))
        ref = vr.sample(logits, top_k=k, top_p=p, min_p=mp, temperature=T_, seed=77, offset=off)
        assert np.array_equal(out.cpu().numpy(), ref), off


@pytest.mark.parametrize("B,V,k,p,mp,T_", [(2, 168960, 0, 0.8, 0.0, 0.8),      # GLM-4-Voice defaults (glm_voice.py:358-366)
                                          (4, 156940, 0, 0.8, 0.0, 0.6),      # Orpheus defaults (orpheus.py:260-268)
                                          (3, 3072, 0, 1.0, 0.1, 1.0),        # Zonos min-p (zonos.py:591-599)
                                          (2, 65536, 0, 0.9, 0.05, 0.7), (2, 4096, 0, 0.0, 0.0, 1.0),
                                          (2, 2048, 0, 1.0, 0.0, 1.0),        # full multinomial
                                          (2, 168960, 50, 1.0, 0.0, 0.9), (2, 156940, 25, 0.8, 0.0, 1.0)])
def test_full_vocab_sampler_bit_exact(dev, B, V, k, p, mp, T_):
    """top-p-only / min-p-only ("bucket" contract) and top-k over vocabularies too large for LDS."""
    from vox_serve_amd import _native as N
    rng = np.random.default_rng(V + k + int(p * 100))
    logits = vr.f2bf(rng.standard_normal((B, V)).astype(np.float32) * 3)
    logits[0, : V // 2] = logits[0, 0]            # one huge bucket on row 0
    logits[-1, ::3] = vr.f2bf(np.float32(4.0))    # the top bucket holds a third of the row
    lt = T(logits, dev)
    out = torch.empty(B, dtype=torch.int32, device=dev)
    cfg = N.SamplingCfg(0, k, p, mp, T_, 1.0)
    for off in range(10):
        N.check(N.lib().vox_sample(N.ctx(), N.stream(), N.ptr(lt), B, V, cfg, 1234, off, N.ptr(out)))
        ref = vr.sample(logits, top_k=k, top_p=p, min_p=mp, temperature=T_, seed=1234, offset=off)
        assert np.array_equal(out.cpu().numpy(), ref), (off, out.cpu().numpy(), ref)
    # the histogram scratch must be left all-zero: a repeated call agrees
    N.check(N.lib().vox_sample(N.ctx(), N.stream(), N.ptr(lt), B, V, cfg, 1234, 9, N.ptr(out)))
    assert np.array_equal(out.cpu().numpy(), ref)


def test_stack_forward_abi_bit_exact(dev):
    """vox_stack_create / vox_stack_forward (the decoder-stack entry point of the C ABI) vs the oracle's RefStack:
    a 7-token ragged prefill, then decode rows with the decode hints (per-row page table), llama-3.1 RoPE, QKV bias."""
    import ctypes
    from oracle import qwen3_ref as QR
    from vox_serve_amd import _native as N
    from vox_serve_amd.engine import StackCfg, _stack_config, rope_table
    L, ctx = N.lib(), N.ctx()
    oc = QR.StackCfg(256, 2, 4, 2, 64, 512, eps=1e-5, rope_theta=5e5, rope_scale=32.0, rope_llama31=(1.0, 4.0, 64), qk_norm=False, qkv_bias=True)
    rng = np.random.default_rng(12)
    w = lambda *s_: vr.f2bf(rng.standard_normal(s_, dtype=np.float32) * np.float32(0.08))
    W = {"m.norm.weight": vr.f2bf(np.ones(256, np.float32))}
    for i in range(2):
        p = f"m.layers.{i}."
        for n_, rows in (("q", 4), ("k", 2), ("v", 2)):
            W[p + f"self_attn.{n_}_proj.weight"], W[p + f"self_attn.{n_}_proj.bias"] = w(rows * 64, 256), w(rows * 64)
        W[p + "self_attn.o_proj.weight"] = w(256, 256)
        W[p + "mlp.gate_proj.weight"], W[p + "mlp.up_proj.weight"], W[p + "mlp.down_proj.weight"] = w(512, 256), w(512, 256), w(256, 512)
        W[p + "input_layernorm.weight"] = vr.f2bf(1 + 0.1 * rng.standard_normal(256).astype(np.float32))
        W[p + "post_attention_layernorm.weight"] = vr.f2bf(1 + 0.1 * rng.standard_normal(256).astype(np.float32))
    ref = QR.RefStack(oc, W, "m", 128)
    page, P = 8, 6
    kv_ref = [np.zeros((P, 2, page, 2, 64), np.uint16) for _ in range(2)]
    ec = StackCfg(256, 2, 4, 2, 64, 512, 1e-5, 5e5, 32.0, None, False, (1.0, 4.0, 64), False, True)
    keep, arr = [], (N.LayerWeights * 2)()
    for i in range(2):
        p = f"m.layers.{i}."
        ts = dict(wqkv=np.concatenate([W[p + f"self_attn.{n_}_proj.weight"] for n_ in "qkv"]),
                  bqkv=np.concatenate([W[p + f"self_attn.{n_}_proj.bias"] for n_ in "qkv"]), wo=W[p + "self_attn.o_proj.weight"],
                  wgate=W[p + "mlp.gate_proj.weight"], wup=W[p + "mlp.up_proj.weight"], wdown=W[p + "mlp.down_proj.weight"],
                  ln1=W[p + "input_layernorm.weight"], ln2=W[p + "post_attention_layernorm.weight"])
        for k, v in ts.items():
            t = T(v, dev)
            keep.append(t)
            setattr(arr[i], k, t.data_ptr())
    fn, rope = T(W["m.norm.weight"], dev), rope_table(128, ec, dev)
    sc = _stack_config(ec, page, 16, 64)
    h = ctypes.c_void_p()
    N.check(L.vox_stack_create(ctx, ctypes.byref(sc), arr, fn.data_ptr(), rope.data_ptr(), 128, ctypes.byref(h)))
    kv = torch.zeros(2, P, 2, page, 2, 64, dtype=torch.bfloat16, device=dev)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)
    # prefill: request 0 = 4 tokens on pages [3,...], request 1 = 3 tokens on page [1]
    x0 = w(7, 256)
    pos, q_req, kvl = [0, 1, 2, 3, 0, 1, 2], [0, 0, 0, 0, 1, 1, 1], [1, 2, 3, 4, 1, 2, 3]
    pg, sl, indptr, indices = [3, 3, 3, 3, 1, 1, 1], [0, 1, 2, 3, 0, 1, 2], [0, 1, 2], [3, 1]
    want = ref.forward(x0.copy(), np.array(pos, np.int32), kv_ref, np.array(q_req, np.int32), np.array(kvl, np.int32),
                       np.array(indptr, np.int32), np.array(indices, np.int32), np.array(pg, np.int32), np.array(sl, np.int32))
    x, y = T(x0, dev), torch.empty(7, 256, dtype=torch.bfloat16, device=dev)
    tens = [i32(v) for v in (pos, q_req, kvl, pg, sl, indptr, indices)]
    rows = N.Rows(*[t.data_ptr() for t in tens], 7, 4, None, 0, 0, -1, 0)
    N.check(L.vox_stack_forward(h, N.stream(), x.data_ptr(), y.data_ptr(), kv.data_ptr(), kv[0].numel(), ctypes.byref(rows)))
    torch.cuda.synchronize()
    assert np.array_equal(Bits(y), want)
    # decode: one new token per request, per-row page table hint -> fused (norm + RoPE + append in-kernel) path
    x1 = w(2, 256)
    pos, q_req, kvl, pg, sl = [5, 4], [0, 1], [5, 4], [3, 1], [4, 3]      # positions follow quirk Q1 (n + 1)
    want = ref.forward(x1.copy(), np.array(pos, np.int32), kv_ref, np.array(q_req, np.int32), np.array(kvl, np.int32),
                       np.array(indptr, np.int32), np.array(indices, np.int32), np.array(pg, np.int32), np.array(sl, np.int32))
    x, y = T(x1, dev), torch.empty(2, 256, dtype=torch.bfloat16, device=dev)
    tens = [i32(v) for v in (pos, q_req, kvl, pg, sl, indptr, indices)]
    ptab = i32([[3, 0], [1, 0]])
    rows = N.Rows(*[t.data_ptr() for t in tens], 2, 5, ptab.data_ptr(), 2, 0, -1, 0)
    N.check(L.vox_stack_forward(h, N.stream(), x.data_ptr(), y.data_ptr(), kv.data_ptr(), kv[0].numel(), ctypes.byref(rows)))
    torch.cuda.synchronize()
    assert np.array_equal(Bits(y), want)
    for l in range(2):
        assert np.array_equal(vr.from_torch(kv[l])[[1, 3]], kv_ref[l][[1, 3]]), l
    L.vox_stack_destroy(h)


def stack_rows_case(tape, dev, n_rows, dims=(1024, 8, 4, 3072), policy=None):
    """9..32 rows take the one-launch full-K MFMA GEMM (norm prologue, SiLU*up / residual epilogues, fragment-major
    weights and activation hand-offs), 33+ the split-K pair: two decoder layers at depth-transformer widths vs the oracle's
    RefStack, through the three attention routes that feed o_proj (single-chunk prefill rows; decode rows at a 41-token
    context = chunked + merge; decode rows on a <= 16-token stack = one-wave short attention).  MFMA accumulation order =>
    bit-exact against the oracle, which follows the same kernel routing (oracle/policy.py).
    tape: live oracle, or the oracle side recorded / replayed (tests/oracle_tape.py) for the larger row counts."""
    import ctypes
    from oracle import qwen3_ref as QR
    (H, heads, kvh, F), D, NL = dims, 128, 2
    oc = QR.StackCfg(H, NL, heads, kvh, D, F, eps=1e-6, rope_theta=1e6, qk_norm=True, qkv_bias=False)
    rng = np.random.default_rng(n_rows)
    w = lambda *s_, sd=0.03: vr.f2bf(rng.standard_normal(s_, dtype=np.float32) * np.float32(sd))
    g = lambda n: vr.f2bf(1 + 0.1 * rng.standard_normal(n).astype(np.float32))
    W = {"m.norm.weight": g(H)}
    for l in range(NL):
        p = f"m.layers.{l}."
        W.update({p + "self_attn.q_proj.weight": w(heads * D, H), p + "self_attn.k_proj.weight": w(kvh * D, H),
                  p + "self_attn.v_proj.weight": w(kvh * D, H), p + "self_attn.o_proj.weight": w(H, heads * D),
                  p + "self_attn.q_norm.weight": g(D), p + "self_attn.k_norm.weight": g(D),
                  p + "mlp.gate_proj.weight": w(F, H), p + "mlp.up_proj.weight": w(F, H), p + "mlp.down_proj.weight": w(H, F),
                  p + "input_layernorm.weight": g(H), p + "post_attention_layernorm.weight": g(H)})
    ref = QR.RefStack(oc, W, "m", 512, policy)
    page, ppr = 8, (64 if n_rows in (3, 12) else 6)        # pages of 8 slots per request (long-context checks: 64)
    A = lambda v: np.array(v, np.int32)
    P = n_rows * ppr
    if tape.gpu:
        from vox_serve_amd import _native as N
        from vox_serve_amd.engine import StackCfg, _stack_config, rope_table
        L, ctx = N.lib(), N.ctx()
        ec = StackCfg(H, NL, heads, kvh, D, F, 1e-6, 1e6, 1.0, None, False, None, True, False)
        arr, keep = (N.LayerWeights * NL)(), []
        for l in range(NL):
            p = f"m.layers.{l}."
            ts = dict(wqkv=np.concatenate([W[p + f"self_attn.{n_}_proj.weight"] for n_ in "qkv"]), wo=W[p + "self_attn.o_proj.weight"],
                      wgate=W[p + "mlp.gate_proj.weight"], wup=W[p + "mlp.up_proj.weight"], wdown=W[p + "mlp.down_proj.weight"],
                      ln1=W[p + "input_layernorm.weight"], ln2=W[p + "post_attention_layernorm.weight"],
                      qnorm=W[p + "self_attn.q_norm.weight"], knorm=W[p + "self_attn.k_norm.weight"])
            for k, v in ts.items():
                t = T(v, dev)
                keep.append(t)
                setattr(arr[l], k, t.data_ptr())
        fn, rope = T(W["m.norm.weight"], dev), rope_table(512, ec, dev)
        i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)

    def check(max_kvlen, kv_tokens, hints):
        """every row = the newest token of its own request, which already holds kv_tokens tokens of random K/V"""
        kv_ref = [np.zeros((P, 2, page, kvh, D), np.uint16) for _ in range(NL)]
        for l in range(NL):
            for r in range(n_rows):
                for t in range(kv_tokens):
                    kv_ref[l][r * ppr + t // page, :, t % page] = w(2, kvh, D, sd=0.5)
        kv0 = [k.copy() for k in kv_ref] if tape.gpu else None
        x0 = w(n_rows, H, sd=1.0)
        n_new = kv_tokens + 1
        pos, q_req, kvl = [n_new] * n_rows, list(range(n_rows)), [n_new] * n_rows
        pg, sl = [r * ppr + kv_tokens // page for r in range(n_rows)], [kv_tokens % page] * n_rows
        npg = kv_tokens // page + 1
        indptr, indices = [r * npg for r in range(n_rows + 1)], [r * ppr + j for r in range(n_rows) for j in range(npg)]
        want = ref.forward(x0.copy(), A(pos), kv_ref, A(q_req), A(kvl), A(indptr), A(indices), A(pg), A(sl)) if tape.oracle else None
        tag = f"kv{kv_tokens} bucket{max_kvlen}"
        if tape.gpu:
            sc = _stack_config(ec, page, 144, max_kvlen)
            h = ctypes.c_void_p()
            N.check(L.vox_stack_create(ctx, ctypes.byref(sc), arr, fn.data_ptr(), rope.data_ptr(), 512, ctypes.byref(h)))
            kv = torch.stack([T(k, dev) for k in kv0])
            x, y = T(x0, dev), torch.empty(n_rows, H, dtype=torch.bfloat16, device=dev)
            tens = [i32(v) for v in (pos, q_req, kvl, pg, sl, indptr, indices)]
            ptab = i32([[r * ppr + j for j in range(ppr)] for r in range(n_rows)])
            rows = N.Rows(*[t.data_ptr() for t in tens], n_rows, n_new, ptab.data_ptr() if hints else None, ppr if hints else 0, 0, -1, 0)
            N.check(L.vox_stack_forward(h, N.stream(), x.data_ptr(), y.data_ptr(), kv.data_ptr(), kv[0].numel(), ctypes.byref(rows)))
            torch.cuda.synchronize()
        tape.check(f"hidden {tag}", lambda: Bits(y), lambda: want)
        # the new token's K/V landed where the oracle put them
        tape.check(f"new K/V {tag}", lambda: np.stack([vr.from_torch(kv[l])[pg, :, sl] for l in range(NL)]),
                   lambda: np.stack([kv_ref[l][pg, :, sl] for l in range(NL)]))
        if tape.gpu:
            L.vox_stack_destroy(h)

    check(64, 0, False)      # prefill-style rows: head_prepare + single-chunk attention
    check(64, 40, True)      # decode rows, 41-token context: fused chunked attention + merge
    check(16, 2, True)       # decode rows on a short-context stack: one-wave attention (heads == 2 * kv heads)
    if ppr == 64:            # 131 / 450-token contexts (5 / 15 chunks of 32 tokens, ragged last chunk, 57 pages per request)
        check(256, 130, True)
        check(512, 449, True)
    tape.done(kind="stack_forward", n_rows=n_rows, dims=list(dims))


# the larger row counts: the oracle's side is recorded ahead of time (tests/oracle_tape.py, tests/golden/make_oracle_tapes.py)
TAPED = {}
for _n in (24, 32, 48, 64, 75, 100, 128, 140):
    TAPED[f"ops_stack_rows_{_n}"] = (lambda tape, dev, n=_n: stack_rows_case(tape, dev, n))
for _n in (12, 32):
    TAPED[f"ops_stack_rows_k4096_{_n}"] = (lambda tape, dev, n=_n: stack_rows_case(tape, dev, n, dims=(4096, 32, 16, 4096)))


@pytest.mark.parametrize("n_rows", [12, 16, 24, 32, 48, 64, 75, 100, 128, 140])
def test_stack_forward_batched_rows_paths(dev, n_rows):
    name = f"ops_stack_rows_{n_rows}"
    if name in TAPED:
        TAPED[name](Tape.open(name), dev)
    else:
        stack_rows_case(Tape(), dev, n_rows)


@pytest.mark.parametrize("n_rows", [3, 4, 8])
def test_stack_forward_small_batches_both_settings(dev, n_rows):
    """3..8 rows: on the matrix cores by default (exact_rows 2), on the wave64 VALU kernels with vox_ctx_set_exact_rows(8);
    both bit-exact against the oracle under the matching policy."""
    from oracle.policy import Policy
    from vox_serve_amd import _native as N
    stack_rows_case(Tape(), dev, n_rows)
    N.set_exact_rows(8)
    try:
        stack_rows_case(Tape(), dev, n_rows, policy=Policy(exact_rows=8))
    finally:
        N.set_exact_rows(2)


@pytest.mark.parametrize("n_rows", [12, 32])
def test_stack_forward_batched_rows_k4096(dev, n_rows):
    """The same three attention routes at GLM-4-Voice width (hidden 4096 = 16 k-steps per wave): copy-prologue linears on the
    full-K kernel directly, norm-prologue ones through the normalise-once-into-scratch route."""
    name = f"ops_stack_rows_k4096_{n_rows}"
    TAPED[name](Tape.open(name), dev)
