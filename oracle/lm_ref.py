"""CPU oracle for the single-stack speech LMs (GLM-4-Voice, CosyVoice2), composed from oracle/voxref.c.

TEST INFRASTRUCTURE ONLY (see oracle/voxref.c header).  Follows, step by step:
  GLMVoiceModel.forward / GLMVoiceForCausalLM        /root/reference/vox_serve/model/glm_voice.py:85-305,517-536
  CosyVoice2Model.forward / CosyVoice2ForCausalLM    /root/reference/vox_serve/model/cosyvoice2.py:106-316,1008-1033
  {GLMVoice,CosyVoice2}Model.sampling                glm_voice.py:538-592, cosyvoice2.py:1035-1091
  ModelWorker.prepare_lm_inputs bookkeeping          worker/base.py:210-360 (first decode position n+1, quirk Q1)
Both families reduce to: embedding gather (optionally overridden per row by input_features), a llama-style
decoder stack, final RMSNorm, an output head (optionally biased), sampler with an optional per-request
repetition cache.  Family differences are configuration (StackCfg) plus the state_dict layout, converted by
`from_glm_state_dict` / `from_cosyvoice2_state_dict` into the names RefStack reads.
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from . import voxref as vr
from .policy import PRO_RMSNORM, Call, Policy
from .qwen3_ref import RefRequest, RefStack, StackCfg


@dataclass
class LMCfg:
    stack: StackCfg
    vocab_in: int                 # rows of the input embedding table
    vocab_out: int                # logits width
    input_mode: int = 0           # 1: x = mask ? input_features : embedding[clamp(id)]  (cosyvoice2.py:1020-1024)
    max_pos: int = 4096
    head_bias: bool = False


def glm_cfg(hidden=4096, layers=40, heads=32, kv_heads=2, ffn=13696, vocab=168960, max_pos=4096) -> LMCfg:
    """glm_voice.py:22-55,104-160: fused QKV with bias, half-rotary interleaved RoPE theta=1e4, eps 3.90625e-08."""
    d = hidden // heads
    return LMCfg(StackCfg(hidden, layers, heads, kv_heads, d, ffn, eps=3.90625e-08, rope_theta=1e4, rope_dim=d // 2,
                          rope_interleave=True, qk_norm=False, qkv_bias=True), vocab, vocab, 0, max_pos, False)


def cosyvoice2_cfg(hidden=896, layers=24, heads=14, kv_heads=2, ffn=4864, speech_vocab=6564, max_pos=4096) -> LMCfg:
    """cosyvoice2.py:27-38,122-168,285-316: Qwen2-0.5B body, QKV bias, NeoX RoPE theta=1e6, biased llm_decoder."""
    return LMCfg(StackCfg(hidden, layers, heads, kv_heads, hidden // heads, ffn, eps=1e-6, rope_theta=1e6,
                          qk_norm=False, qkv_bias=True), speech_vocab, speech_vocab, 1, max_pos, True)


def tiny_glm_cfg() -> LMCfg:
    return glm_cfg(hidden=256, layers=2, heads=4, kv_heads=2, ffn=512, vocab=1536, max_pos=512)


def tiny_cosyvoice2_cfg() -> LMCfg:
    return cosyvoice2_cfg(hidden=192, layers=2, heads=3, kv_heads=1, ffn=384, speech_vocab=516, max_pos=512)


# ---- synthetic weights under the reference's state_dict names (so the same dict loads into its modules) ----
def _ones(n, device):
    o = vr.f2bf(np.ones(n, np.float32))
    return o if device is None else vr.to_torch(o).to(device)


def random_glm_state_dict(cfg: LMCfg, seed=0, std=0.02, device=None) -> Dict[str, np.ndarray]:
    """`device`: the same bits as torch bf16 tensors on that device (vr.random_bf16), else numpy bit arrays."""
    rng = np.random.default_rng(seed)
    c = cfg.stack
    w = lambda *s: vr.random_bf16(rng, s, std, device)
    ones = lambda n: _ones(n, device)
    qkv = (c.heads + 2 * c.kv_heads) * c.head_dim
    W = {"transformer.embedding.word_embeddings.weight": w(cfg.vocab_in, c.hidden),
         "transformer.encoder.final_layernorm.weight": ones(c.hidden),
         "transformer.output_layer.weight": w(cfg.vocab_out, c.hidden)}
    for i in range(c.layers):
        p = f"transformer.encoder.layers.{i}."
        W[p + "self_attention.query_key_value.weight"] = w(qkv, c.hidden)
        W[p + "self_attention.query_key_value.bias"] = w(qkv)
        W[p + "self_attention.dense.weight"] = w(c.hidden, c.hidden)
        W[p + "mlp.dense_h_to_4h.weight"] = w(2 * c.ffn, c.hidden)
        W[p + "mlp.dense_4h_to_h.weight"] = w(c.hidden, c.ffn)
        W[p + "input_layernorm.weight"] = ones(c.hidden)
        W[p + "post_attention_layernorm.weight"] = ones(c.hidden)
    return W


def random_cosyvoice2_state_dict(cfg: LMCfg, seed=0, std=0.02, text_vocab=640, device=None) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    c = cfg.stack
    w = lambda *s: vr.random_bf16(rng, s, std, device)
    ones = lambda n: _ones(n, device)
    W = {"llm.model.model.embed_tokens.weight": w(text_vocab, c.hidden), "llm.model.model.norm.weight": ones(c.hidden),
         "llm.model.lm_head.weight": w(text_vocab, c.hidden),        # unused by the path (cosyvoice2.py:247-248)
         "llm_embedding.weight": w(2, c.hidden), "llm_decoder.weight": w(cfg.vocab_out, c.hidden),
         "llm_decoder.bias": w(cfg.vocab_out), "speech_embedding.weight": w(cfg.vocab_in, c.hidden)}
    for i in range(c.layers):
        p = f"llm.model.model.layers.{i}."
        for n, rows in (("q", c.heads), ("k", c.kv_heads), ("v", c.kv_heads)):
            W[p + f"self_attn.{n}_proj.weight"] = w(rows * c.head_dim, c.hidden)
            W[p + f"self_attn.{n}_proj.bias"] = w(rows * c.head_dim)
        W[p + "self_attn.o_proj.weight"] = w(c.hidden, c.heads * c.head_dim)
        W[p + "mlp.gate_proj.weight"] = w(c.ffn, c.hidden)
        W[p + "mlp.up_proj.weight"] = w(c.ffn, c.hidden)
        W[p + "mlp.down_proj.weight"] = w(c.hidden, c.ffn)
        W[p + "input_layernorm.weight"] = ones(c.hidden)
        W[p + "post_attention_layernorm.weight"] = ones(c.hidden)
    return W


# ---- state_dict -> the names RefStack reads (prefix "model") + embedding / head ----
def from_glm_state_dict(cfg: LMCfg, S) -> Dict[str, np.ndarray]:
    c = cfg.stack
    nq, nk = c.heads * c.head_dim, c.kv_heads * c.head_dim
    W = {"embedding": S["transformer.embedding.word_embeddings.weight"], "head_w": S["transformer.output_layer.weight"],
         "model.norm.weight": S["transformer.encoder.final_layernorm.weight"]}
    for i in range(c.layers):
        s, d = f"transformer.encoder.layers.{i}.", f"model.layers.{i}."
        qkv, b = S[s + "self_attention.query_key_value.weight"], S[s + "self_attention.query_key_value.bias"]
        for n, (lo, hi) in (("q", (0, nq)), ("k", (nq, nq + nk)), ("v", (nq + nk, nq + 2 * nk))):   # glm_voice.py:137-144
            W[d + f"self_attn.{n}_proj.weight"] = np.ascontiguousarray(qkv[lo:hi])
            W[d + f"self_attn.{n}_proj.bias"] = np.ascontiguousarray(b[lo:hi])
        W[d + "self_attn.o_proj.weight"] = S[s + "self_attention.dense.weight"]
        h4 = S[s + "mlp.dense_h_to_4h.weight"]                      # chunk(2): silu(first half) * second half (glm_voice.py:95-97)
        W[d + "mlp.gate_proj.weight"] = np.ascontiguousarray(h4[:c.ffn])
        W[d + "mlp.up_proj.weight"] = np.ascontiguousarray(h4[c.ffn:])
        W[d + "mlp.down_proj.weight"] = S[s + "mlp.dense_4h_to_h.weight"]
        W[d + "input_layernorm.weight"] = S[s + "input_layernorm.weight"]
        W[d + "post_attention_layernorm.weight"] = S[s + "post_attention_layernorm.weight"]
    return W


def from_cosyvoice2_state_dict(cfg: LMCfg, S) -> Dict[str, np.ndarray]:
    W = {"embedding": S["speech_embedding.weight"], "head_w": S["llm_decoder.weight"], "head_b": S["llm_decoder.bias"],
         "model.norm.weight": S["llm.model.model.norm.weight"]}
    for k, v in S.items():
        if k.startswith("llm.model.model.layers."):
            W["model." + k[len("llm.model.model."):]] = v
    return W


@dataclass
class LMRequest(RefRequest):
    rep_cache: Optional[np.ndarray] = None       # [W,1,V] uint8, persisted across steps (glm_voice.py:585-588)
    tokens: List[int] = field(default_factory=list)


class LMRef:
    def __init__(self, cfg: LMCfg, W, page_size=128, max_pages=64, policy: Optional[Policy] = None, dry=False):
        """dry: page / position bookkeeping only (see Qwen3Ref)."""
        self.cfg, self.W, self.page_size, self.dry = cfg, W, page_size, dry
        self.policy = policy or Policy()
        c = cfg.stack
        self.free_pages = list(range(max_pages))
        if dry:
            self.kv = None
            return
        self.stack = RefStack(c, W, "model", cfg.max_pos, self.policy)
        self.kv = [np.zeros((max_pages, 2, page_size, c.kv_heads, c.head_dim), np.uint16) for _ in range(c.layers)]
        self.free_pages = list(range(max_pages))

    def embed(self, ids, masks=None, feats=None):
        ids = np.clip(np.asarray(ids, np.int32), 0, self.cfg.vocab_in - 1)      # cosyvoice2.py:1020-1022
        x = vr.gather(self.W["embedding"], ids)
        if self.cfg.input_mode == 1 and masks is not None:
            x = np.where(np.asarray(masks, bool)[:, None], feats, x)            # cosyvoice2.py:1024
        return np.ascontiguousarray(x)

    def _pinned(self, n):
        er = self.policy.exact_rows
        return er is None or n <= er

    def head(self, xs, x_rows=False):
        """final norm fused into the head linear (engine.hip lm_run): the canonical kernels for <= exact_rows requests, else the
        matrix-core kernel the shape routes to (K = 4096 without row indirection: normalised once, then the full-K GEMM)."""
        W, c = self.W, self.cfg.stack
        call = Call(B=xs.shape[0], N=W["head_w"].shape[0], K=c.hidden, pro=PRO_RMSNORM, x_rows=x_rows,
                    fixed_order=self._pinned(xs.shape[0]), norm_scratch=True)
        od, nd = self.policy.route(call)
        h = vr.rmsnorm(xs, W["model.norm.weight"], c.eps, order=nd)
        return vr.linear(W["head_w"], h, W.get("head_b"), order=od)

    def prefill(self, req: LMRequest, ids, masks=None, feats=None):
        n, ps = len(ids), self.page_size
        npg = (n + ps - 1) // ps
        req.kv_pages = [self.free_pages.pop(0) for _ in range(npg)]
        req.kv_token_len, req.kv_last_page_len = n, n % ps or ps
        req.next_position_id = n + 1                                           # quirk Q1 (worker/base.py:299)
        if self.dry:
            return None
        page = np.array([req.kv_pages[t // ps] for t in range(n)], np.int32)
        slot = np.array([t % ps for t in range(n)], np.int32)
        xs = self.stack.forward(self.embed(ids, masks, feats), np.arange(n, dtype=np.int32), self.kv, np.zeros(n, np.int32),
                                np.arange(1, n + 1, dtype=np.int32), np.array([0, npg], np.int32),
                                np.array(req.kv_pages, np.int32), page, slot, final_norm=False, fixed_order=self._pinned(n))
        return self.head(xs[-1:], x_rows=True)

    def decode(self, reqs: List[LMRequest]):
        ps, B = self.page_size, len(reqs)
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
            return None
        ids = np.array([r.input_ids[0, 0] for r in reqs], np.int32)
        masks = np.array([r.input_mask for r in reqs], np.uint8)
        feats = None
        if self.cfg.input_mode == 1:
            feats = np.concatenate([r.input_features for r in reqs], 0)
        xs = self.stack.forward(self.embed(ids, masks, feats), np.array(pos, np.int32), self.kv, np.arange(B, dtype=np.int32),
                                np.array(kvlen, np.int32), np.array(indptr, np.int32), np.array(indices, np.int32),
                                np.array(page, np.int32), np.array(slot, np.int32), final_norm=False, fixed_order=self._pinned(B))
        return self.head(xs)

    def sample(self, logits, reqs: List[LMRequest], sampler=None, penalty=1.0, window=None):
        """{GLMVoice,CosyVoice2}Model.sampling: penalty -> draw -> cache update -> next inputs."""
        if self.dry:
            return None, None
        use_rep = reqs[0].rep_cache is not None and penalty != 1.0
        if use_rep:
            cache = np.ascontiguousarray(np.stack([r.rep_cache for r in reqs], 0))
            logits = vr.rep_penalty(logits, cache, penalty)
        ids = vr.argmax(logits) if sampler is None else sampler(logits)
        if use_rep:
            vr.rep_update(cache, ids, window)
        for b, r in enumerate(reqs):
            if use_rep:
                r.rep_cache = cache[b].copy()
            r.input_ids = np.array([[ids[b]]], np.int32)
            r.input_mask = False                                                # cosyvoice2.py:1062-1064
            if self.cfg.input_mode == 1:
                r.input_features = np.zeros((1, self.cfg.stack.hidden), np.uint16)
            r.tokens.append(int(ids[b]))
        return ids, logits
