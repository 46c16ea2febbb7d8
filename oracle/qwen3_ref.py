"""CPU oracle model for the Qwen3-TTS speech-LM decode path, composed from oracle/voxref.c primitives.

TEST INFRASTRUCTURE ONLY (see oracle/voxref.c header).  Follows, step by step:
  Qwen3TTSModel.forward            /root/reference/vox_serve/model/qwen3_tts.py:1805-1861
  Qwen3TTSDecoderLayer / Attention / MLP                           qwen3_tts.py:573-704
  Qwen3TTSModel.sampling           qwen3_tts.py:1863-1962   (greedy / seeded sampler contract)
  run_lm_depth + depth_forward + depth_sampling   worker/base.py:546-614, qwen3_tts.py:923-944,1964-2004
  ModelWorker.prepare_lm_inputs page/position bookkeeping          worker/base.py:210-360 (incl. quirk Q1)
Weights use the reference's state_dict names so the same dict loads into the reference nn.Module.
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from . import voxref as vr
from .policy import EPI_SILU, EPI_SILU_MUL, NORM_ROWS1024, PRO_RMSNORM, Call, Policy


@dataclass
class StackCfg:
    hidden: int
    layers: int
    heads: int
    kv_heads: int
    head_dim: int
    ffn: int
    eps: float = 1e-6
    rope_theta: float = 1e6
    rope_scale: float = 1.0
    rope_dim: Optional[int] = None
    rope_interleave: bool = False
    rope_llama31: Optional[tuple] = None      # (low, high, old_ctx)
    qk_norm: bool = True
    qkv_bias: bool = False


@dataclass
class Qwen3Cfg:
    talker: StackCfg = field(default_factory=lambda: StackCfg(2048, 28, 16, 8, 128, 6144))
    depth: StackCfg = field(default_factory=lambda: StackCfg(1024, 5, 16, 8, 128, 3072))
    vocab: int = 3072
    text_vocab: int = 151936
    text_hidden: int = 2048
    depth_vocab: int = 2048
    n_groups: int = 16
    eos_id: int = 2150
    tts_pad_id: int = 151671
    max_pos: int = 4096

    @property
    def suppress_ids(self):
        return [i for i in range(self.vocab - 1024, self.vocab) if i != self.eos_id]


def tiny_cfg() -> Qwen3Cfg:
    """Small config with the same structure (all dims multiples of 8) for fast parity tests."""
    return Qwen3Cfg(talker=StackCfg(256, 2, 4, 2, 64, 512), depth=StackCfg(128, 2, 4, 2, 64, 256),
                    vocab=1280, text_vocab=512, text_hidden=256, depth_vocab=256, n_groups=4, eos_id=300,
                    tts_pad_id=7, max_pos=512)


def random_weights(cfg: Qwen3Cfg, seed: int = 0, std: float = 0.02, device=None) -> Dict[str, np.ndarray]:
    """N(0, std^2) bf16 weights, norm weights 1 (SURVEY §8d synthetic recipe); reference state_dict names.  `device`: the same
    bits as torch bf16 tensors on that device (large tensors are generated there, vr.random_bf16), else numpy bit arrays."""
    rng = np.random.default_rng(seed)

    def w(*shape, s=std):
        return vr.random_bf16(rng, shape, s, device)

    def ones(n):
        o = vr.f2bf(np.ones(n, np.float32))
        return o if device is None else vr.to_torch(o).to(device)

    W: Dict[str, np.ndarray] = {}

    def stack(prefix, c: StackCfg):
        for i in range(c.layers):
            p = f"{prefix}.layers.{i}."
            W[p + "self_attn.q_proj.weight"] = w(c.heads * c.head_dim, c.hidden)
            W[p + "self_attn.k_proj.weight"] = w(c.kv_heads * c.head_dim, c.hidden)
            W[p + "self_attn.v_proj.weight"] = w(c.kv_heads * c.head_dim, c.hidden)
            W[p + "self_attn.o_proj.weight"] = w(c.hidden, c.heads * c.head_dim)
            W[p + "self_attn.q_norm.weight"] = ones(c.head_dim)
            W[p + "self_attn.k_norm.weight"] = ones(c.head_dim)
            W[p + "mlp.gate_proj.weight"] = w(c.ffn, c.hidden)
            W[p + "mlp.up_proj.weight"] = w(c.ffn, c.hidden)
            W[p + "mlp.down_proj.weight"] = w(c.hidden, c.ffn)
            W[p + "input_layernorm.weight"] = ones(c.hidden)
            W[p + "post_attention_layernorm.weight"] = ones(c.hidden)
        W[prefix + ".norm.weight"] = ones(c.hidden)

    H = cfg.talker.hidden
    stack("talker.model", cfg.talker)
    W["talker.model.codec_embedding.weight"] = w(cfg.vocab, H)
    W["talker.model.text_embedding.weight"] = w(cfg.text_vocab, cfg.text_hidden)
    W["talker.text_projection.linear_fc1.weight"] = w(cfg.text_hidden, cfg.text_hidden)
    W["talker.text_projection.linear_fc1.bias"] = w(cfg.text_hidden)
    W["talker.text_projection.linear_fc2.weight"] = w(H, cfg.text_hidden)
    W["talker.text_projection.linear_fc2.bias"] = w(H)
    W["talker.codec_head.weight"] = w(cfg.vocab, H)
    stack("talker.code_predictor.model", cfg.depth)
    for j in range(cfg.n_groups - 1):
        W[f"talker.code_predictor.model.codec_embedding.{j}.weight"] = w(cfg.depth_vocab, H)
        W[f"talker.code_predictor.lm_head.{j}.weight"] = w(cfg.depth_vocab, cfg.depth.hidden)
    W["talker.code_predictor.small_to_mtp_projection.weight"] = w(cfg.depth.hidden, H)
    W["talker.code_predictor.small_to_mtp_projection.bias"] = w(cfg.depth.hidden)
    return W


class RefStack:
    """Decoder stack (qwen3_tts.py:611-739) over a paged KV cache [L][P,2,page,Hkv,D].

    `policy` (oracle/policy.py) selects the summation order of every linear from the call's row count and shape: the
    canonical order up to `exact_rows` rows, the matrix-core orders above."""

    def __init__(self, c: StackCfg, W, prefix, max_pos, policy: Optional[Policy] = None):
        self.c, self.W, self.p = c, W, prefix
        self.policy = policy or Policy()
        self.cs = vr.rope_table(max_pos, c.rope_dim or c.head_dim, c.rope_theta, c.rope_scale, c.rope_llama31)

    def forward(self, x, pos, kv, q_req, q_kvlen, indptr, indices, page, slot, final_norm=True, fixed_order=False):
        c, W, P = self.c, self.W, self.policy
        N = x.shape[0]
        nq, nkv = c.heads * c.head_dim, c.kv_heads * c.head_dim
        st = dict(B=N, fixed_order=fixed_order, splitk_ws=True)
        xn = None            # norm(x) written by the producing split-K linear's fused reduce (block order), if any
        for i in range(c.layers):
            p = f"{self.p}.layers.{i}."
            a = Call(N=nq + 2 * nkv, K=c.hidden, pro=PRO_RMSNORM, norm_scratch=True, x_prenormed=xn is not None, **st)
            oa, na = P.route(a)
            h = xn if (xn is not None and P.rows_gemm(a)) else vr.rmsnorm(x, W[p + "input_layernorm.weight"], c.eps, order=na)
            q = vr.linear(W[p + "self_attn.q_proj.weight"], h, W.get(p + "self_attn.q_proj.bias"), order=oa)
            k = vr.linear(W[p + "self_attn.k_proj.weight"], h, W.get(p + "self_attn.k_proj.bias"), order=oa)
            v = vr.linear(W[p + "self_attn.v_proj.weight"], h, W.get(p + "self_attn.v_proj.bias"), order=oa)
            if c.qk_norm:
                q = vr.rmsnorm(q.reshape(-1, c.head_dim), W[p + "self_attn.q_norm.weight"], c.eps)
                k = vr.rmsnorm(k.reshape(-1, c.head_dim), W[p + "self_attn.k_norm.weight"], c.eps)
            q = vr.rope(q.reshape(N, c.heads, c.head_dim), pos, self.cs, c.rope_dim, c.rope_interleave)
            k = vr.rope(k.reshape(N, c.kv_heads, c.head_dim), pos, self.cs, c.rope_dim, c.rope_interleave)
            vr.kv_append(kv[i], k, v.reshape(N, c.kv_heads, c.head_dim), page, slot)
            at = vr.paged_attention(q, kv[i], q_req, q_kvlen, indptr, indices)
            o = Call(N=c.hidden, K=nq, **st)
            x = vr.linear(W[p + "self_attn.o_proj.weight"], at.reshape(N, -1), residual=x, order=P.route(o)[0])
            xn = vr.rmsnorm(x, W[p + "post_attention_layernorm.weight"], c.eps, order=NORM_ROWS1024) if P.rows_gemm(o) else None
            g = Call(N=c.ffn, K=c.hidden, pro=PRO_RMSNORM, epi=EPI_SILU_MUL, norm_scratch=True, x_prenormed=xn is not None, **st)
            og, ng = P.route(g)
            h = xn if (xn is not None and P.rows_gemm(g)) else vr.rmsnorm(x, W[p + "post_attention_layernorm.weight"], c.eps, order=ng)
            gg = vr.linear_silu_mul(W[p + "mlp.gate_proj.weight"], W[p + "mlp.up_proj.weight"], h, order=og)
            d = Call(N=c.hidden, K=c.ffn, **st)
            x = vr.linear(W[p + "mlp.down_proj.weight"], gg, residual=x, order=P.route(d)[0])
            xn = (vr.rmsnorm(x, W[f"{self.p}.layers.{i + 1}.input_layernorm.weight"], c.eps, order=NORM_ROWS1024)
                  if i + 1 < c.layers and P.rows_gemm(d) else None)
        return vr.rmsnorm(x, W[self.p + ".norm.weight"], c.eps) if final_norm else x

    def norm_head(self, x, head_w, head_b=None, x_out=False, x_rows=False):
        """final norm fused into the head linear's prologue -> (normed rows, logits)."""
        call = Call(B=x.shape[0], N=head_w.shape[0], K=head_w.shape[1], pro=PRO_RMSNORM, x_out=x_out, x_rows=x_rows)
        od, nd = self.policy.route(call)
        h = vr.rmsnorm(x, self.W[self.p + ".norm.weight"], self.c.eps, order=nd)
        return h, vr.linear(head_w, h, head_b, order=od)


DRY = "dry-oracle"      # what a dry oracle hands back in place of a tensor (so that `frame(reqs, logits, hidden)` still knows
                        # that the step was already run by prefill / decode)


@dataclass
class RefRequest:
    """Mirror of the KV / position bookkeeping fields of vox_serve.requests.Request (requests.py:12-77)."""
    kv_pages: List[int] = field(default_factory=list)
    kv_token_len: int = 0
    kv_last_page_len: int = 0
    next_position_id: int = 0
    input_ids: np.ndarray = None       # [1, n_groups+1] int32
    input_mask: bool = True
    input_features: np.ndarray = None  # [1,H] bf16 bits
    frames: List[np.ndarray] = field(default_factory=list)


class Qwen3Ref:
    def __init__(self, cfg: Qwen3Cfg, W, page_size=128, max_pages=64, max_batch=8, policy: Optional[Policy] = None, dry=False):
        """dry: page / position bookkeeping only, no arithmetic — prefill / decode / frame return None where they would return
        tensors.  Used when a GPU test replays an oracle run recorded earlier (tests/oracle_tape.py)."""
        self.cfg, self.W, self.page_size, self.dry = cfg, W, page_size, dry
        self.policy = policy or Policy()
        t, d = cfg.talker, cfg.depth
        self.free_pages = list(range(max_pages))
        if dry:
            self.kv = self.dkv = None
            return
        self.talker = RefStack(t, W, "talker.model", cfg.max_pos, self.policy)
        self.depth = RefStack(d, W, "talker.code_predictor.model", 64, self.policy)
        self.kv = [np.zeros((max_pages, 2, page_size, t.kv_heads, t.head_dim), np.uint16) for _ in range(t.layers)]
        # depth KV: one page of n_groups slots per batch row (worker/base.py:196-205)
        self.dkv = [np.zeros((max_batch, 2, cfg.n_groups, d.kv_heads, d.head_dim), np.uint16)
                    for _ in range(d.layers)]
        self.free_pages = list(range(max_pages))

    # ---- embedding mix (qwen3_tts.py:1836-1852) ----
    def embed(self, input_ids, masks, feats):
        W = self.W
        te = vr.gather(W["talker.model.text_embedding.weight"], input_ids[:, -1])
        n, th = te.shape[0], self.cfg.text_hidden
        o1 = self.policy.route(Call(B=n, N=th, K=th, epi=EPI_SILU))[0]
        o2 = self.policy.route(Call(B=n, N=self.cfg.talker.hidden, K=th))[0]
        h = vr.linear(W["talker.text_projection.linear_fc1.weight"], te, W["talker.text_projection.linear_fc1.bias"], order=o1, act=1)
        text = vr.linear(W["talker.text_projection.linear_fc2.weight"], h, W["talker.text_projection.linear_fc2.bias"], order=o2)
        codec = vr.gather(W["talker.model.codec_embedding.weight"], input_ids[:, 0])
        return vr.qwen3_mix(text, codec, masks, feats)

    # ---- prefill of one request (worker/base.py:253-300) ----
    def prefill(self, req: RefRequest, input_ids, masks, feats):
        n = input_ids.shape[0]
        ps = self.page_size
        npg = (n + ps - 1) // ps
        req.kv_pages = [self.free_pages.pop(0) for _ in range(npg)]
        req.kv_token_len = n
        req.kv_last_page_len = n % ps or ps
        req.next_position_id = n + 1                       # quirk Q1 (worker/base.py:299)
        if self.dry:
            return DRY, DRY
        pos = np.arange(n, dtype=np.int32)
        x = self.embed(input_ids, masks, feats)
        indptr, indices = np.array([0, npg], np.int32), np.array(req.kv_pages, np.int32)
        page = np.array([req.kv_pages[t // ps] for t in range(n)], np.int32)
        slot = np.array([t % ps for t in range(n)], np.int32)
        xs = self.talker.forward(x, pos, self.kv, np.zeros(n, np.int32), np.arange(1, n + 1, dtype=np.int32),
                                 indptr, indices, page, slot, final_norm=False)
        hid, logits = self.talker.norm_head(xs[-1:], self.W["talker.codec_head.weight"], x_out=True, x_rows=True)
        return logits, hid

    # ---- one decode step for a batch (worker/base.py:302-325 + qwen3_tts.py:1805-1861) ----
    def decode(self, reqs: List[RefRequest]):
        ps = self.page_size
        B = len(reqs)
        indptr, indices, page, slot, pos, kvlen = [0], [], [], [], [], []
        for r in reqs:
            r.kv_token_len += 1
            r.kv_last_page_len += 1
            if r.kv_last_page_len > ps:
                r.kv_pages.append(self.free_pages.pop(0))
                r.kv_last_page_len = 1
            indptr.append(indptr[-1] + len(r.kv_pages))
            indices.extend(r.kv_pages)
            page.append(r.kv_pages[-1])
            slot.append(r.kv_last_page_len - 1)
            pos.append(r.next_position_id)
            kvlen.append(r.kv_token_len)
            r.next_position_id += 1
        if self.dry:
            return DRY, DRY
        ids = np.concatenate([r.input_ids for r in reqs], 0)
        masks = np.array([r.input_mask for r in reqs], np.uint8)
        feats = np.concatenate([r.input_features for r in reqs], 0)
        x = self.embed(ids, masks, feats)
        xs = self.talker.forward(x, np.array(pos, np.int32), self.kv, np.arange(B, dtype=np.int32),
                                 np.array(kvlen, np.int32), np.array(indptr, np.int32),
                                 np.array(indices, np.int32), np.array(page, np.int32), np.array(slot, np.int32), final_norm=False)
        hid, logits = self.talker.norm_head(xs, self.W["talker.codec_head.weight"], x_out=True)
        return logits, hid

    # ---- sampling of codebook 0 (qwen3_tts.py:1863-1925) ----
    def sample0(self, logits, sampler=None):
        logits = vr.suppress(logits, self.cfg.suppress_ids)
        # repetition penalty is an identity for Qwen3 (quirk Q3: cache never persisted) -> op kept, mask empty
        return (vr.argmax(logits) if sampler is None else sampler(logits, 0)), logits

    # ---- depth loop (worker/base.py:546-614; qwen3_tts.py:923-944, 1981-2004) ----
    def depth_loop(self, hid, c0, sampler=None):
        cfg, W = self.cfg, self.W
        B = hid.shape[0]
        for l in self.dkv:
            l[:] = 0
        out = np.zeros((B, cfg.n_groups + 1), np.int32)
        out[:, 0] = c0
        out[:, -1] = cfg.tts_pad_id
        c0e = vr.gather(W["talker.model.codec_embedding.weight"], c0)
        feats = np.zeros((B, cfg.talker.hidden), np.uint16)          # req.input_features = zeros (qwen3_tts.py:1944)
        indptr = np.arange(B + 1, dtype=np.int32)
        indices = np.arange(B, dtype=np.int32)
        all_logits = []
        for i in range(1, cfg.n_groups):
            if i == 1:
                x = np.stack([hid, c0e], 1).reshape(2 * B, -1)
                pos = np.tile(np.array([0, 1], np.int32), B)
                q_req = np.repeat(np.arange(B, dtype=np.int32), 2)
                q_kvlen = np.tile(np.array([1, 2], np.int32), B)
                slot = pos.copy()
            else:
                pos = np.full(B, i, np.int32)
                q_req = np.arange(B, dtype=np.int32)
                q_kvlen = np.full(B, i + 1, np.int32)
                slot = pos.copy()
            # the engine pins the canonical kernels for the depth stack of a batch of <= exact_rows requests (step 1 has 2B
            # rows); steps >= 2 gather small_to_mtp_projection(embedding[id]) from a table built with the canonical kernel
            er = self.policy.exact_rows
            pinned = er is None or B <= er
            tabulated = i >= 2 and cfg.n_groups > 2
            op = vr.ORD_CANON if tabulated else self.policy.route(
                Call(B=x.shape[0], N=cfg.depth.hidden, K=cfg.talker.hidden, fixed_order=pinned, splitk_ws=True))[0]
            x = vr.linear(W["talker.code_predictor.small_to_mtp_projection.weight"], x,
                          W["talker.code_predictor.small_to_mtp_projection.bias"], order=op)
            xs = self.depth.forward(x, pos, self.dkv, q_req, q_kvlen, indptr, indices, q_req.copy(), slot, final_norm=False,
                                    fixed_order=pinned)
            if i == 1:
                xs = xs[1::2]
            h, logits = self.depth.norm_head(xs, W[f"talker.code_predictor.lm_head.{i - 1}.weight"], x_rows=(i == 1))
            all_logits.append(logits)
            ids = vr.argmax(logits) if sampler is None else sampler(logits, i)
            out[:, i] = ids
            x = vr.gather(W[f"talker.code_predictor.model.codec_embedding.{i - 1}.weight"], ids)
            feats = vr.add(feats, x)                                  # req.input_features[:] += ci_embed
        return out, feats, all_logits

    # ---- one full frame for a batch of already-prefilled requests ----
    def frame(self, reqs: List[RefRequest], first_logits=None, first_hidden=None, sampler=None):
        if first_logits is None:
            logits, hid = self.decode(reqs)
        else:
            logits, hid = first_logits, first_hidden
        if self.dry:
            return None, None, None, None
        c0, masked = self.sample0(logits, sampler)
        out, feats, dl = self.depth_loop(hid, c0, sampler)
        for b, r in enumerate(reqs):
            ids = np.zeros((1, self.cfg.n_groups + 1), np.int32)
            ids[0, 0] = c0[b]
            ids[0, -1] = self.cfg.tts_pad_id
            r.input_ids, r.input_mask, r.input_features = ids, True, feats[b:b + 1].copy()
            r.frames.append(out[b].copy())
        return out, masked, hid, dl


def prompt_features(W: Dict[str, np.ndarray], cfg: Qwen3Cfg, n_rows: int, speaker_row, spk_bits, icl_row, ref_codes,
                    codec_pad_id: int) -> np.ndarray:
    """input_features (bf16 bits [n_rows, H]) of a voice-clone prompt, as Qwen3TTSModel.preprocess builds them
    (/root/reference/vox_serve/model/qwen3_tts.py:1657-1672 speaker position, :1733-1744 reference-code rows): the
    speaker row holds bf16(speaker_embedding - codec_embedding[codec_pad]); each reference frame's row is the in-place
    bf16 `+=` of code_predictor.codec_embedding[cb-1][code] over codebooks 1..G-1, i.e. rounded to bf16 after every add."""
    H = cfg.talker.hidden
    out = np.zeros((n_rows, H), np.uint16)
    if speaker_row is not None:
        pad = vr.bf2f(W["talker.model.codec_embedding.weight"][codec_pad_id])
        out[speaker_row] = vr.f2bf(vr.bf2f(spk_bits) - pad)
    if icl_row is not None:
        T = ref_codes.shape[0]
        acc = np.zeros((T, H), np.float32)
        for cb in range(1, cfg.n_groups):
            tab = W[f"talker.code_predictor.model.codec_embedding.{cb - 1}.weight"]
            acc = vr.bf2f(vr.f2bf(acc + vr.bf2f(tab[ref_codes[:, cb]])))
        out[icl_row:icl_row + T] = vr.f2bf(acc)
    return out
