"""CPU oracle for the CSM-1B speech-LM path, composed from oracle/voxref.c primitives.

TEST INFRASTRUCTURE ONLY (see oracle/voxref.c header).  Follows, step by step:
  CSMModel.forward (33 masked embeddings summed)      /root/reference/vox_serve/model/csm.py:637-663
  CsmBackboneModel / CsmDecoderLayer / CsmAttention (llama-3.1 RoPE, no q/k-norm)        csm.py:55-200
  CSMModel.sampling (codebook 0, output row = c0 repeated, next inputs)                   csm.py:665-725
  run_lm_depth + depth_forward + CsmCodebooksHead + depth_sampling    worker/base.py:546-614, csm.py:202-313,727-770
Weights use the reference's state_dict names (CsmForConditionalGeneration) so the same dict loads into its module.
"""
from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np

from . import voxref as vr
from .policy import Call, Policy
from .qwen3_ref import DRY, RefRequest, RefStack, StackCfg


@dataclass
class CSMCfg:
    backbone: StackCfg = field(default_factory=lambda: StackCfg(2048, 16, 32, 8, 64, 8192, eps=1e-5, rope_theta=5e5,
                                                                rope_scale=32.0, rope_llama31=(1.0, 4.0, 8192), qk_norm=False))
    depth: StackCfg = field(default_factory=lambda: StackCfg(1024, 4, 8, 2, 128, 8192, eps=1e-5, rope_theta=5e5,
                                                             rope_scale=32.0, rope_llama31=(1.0, 4.0, 8192), qk_norm=False))
    vocab: int = 2051
    text_vocab: int = 128256
    n_codebooks: int = 32
    max_pos: int = 2048


def tiny_csm_cfg() -> CSMCfg:
    ll = (1.0, 4.0, 64)      # small original context so that the llama-3.1 frequency bands are all exercised
    return CSMCfg(backbone=StackCfg(256, 2, 4, 2, 64, 512, eps=1e-5, rope_theta=5e5, rope_scale=32.0, rope_llama31=ll, qk_norm=False),
                  depth=StackCfg(128, 2, 2, 1, 64, 256, eps=1e-5, rope_theta=5e5, rope_scale=32.0, rope_llama31=ll, qk_norm=False),
                  vocab=136, text_vocab=320, n_codebooks=6, max_pos=512)


def random_csm_state_dict(cfg: CSMCfg, seed=0, std=0.02, device=None) -> Dict[str, np.ndarray]:
    """`device`: the same bits as torch bf16 tensors on that device (vr.random_bf16), else numpy bit arrays."""
    rng = np.random.default_rng(seed)
    w = lambda *s: vr.random_bf16(rng, s, std, device)

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
            W[p + "mlp.gate_proj.weight"] = w(c.ffn, c.hidden)
            W[p + "mlp.up_proj.weight"] = w(c.ffn, c.hidden)
            W[p + "mlp.down_proj.weight"] = w(c.hidden, c.ffn)
            W[p + "input_layernorm.weight"] = ones(c.hidden)
            W[p + "post_attention_layernorm.weight"] = ones(c.hidden)
        W[prefix + ".norm.weight"] = ones(c.hidden)

    b, d, C, V = cfg.backbone, cfg.depth, cfg.n_codebooks, cfg.vocab
    stack("backbone_model", b)
    W["backbone_model.embed_tokens.embed_audio_tokens.weight"] = w(C * V, b.hidden)
    W["embed_text_tokens.weight"] = w(cfg.text_vocab, b.hidden)
    W["lm_head.weight"] = w(V, b.hidden)
    stack("depth_decoder.model", d)
    W["depth_decoder.model.embed_tokens.weight"] = w(C * V, b.hidden)          # unused by the serving path
    W["depth_decoder.model.inputs_embeds_projector.weight"] = w(d.hidden, b.hidden)
    W["depth_decoder.codebooks_head.weight"] = w(C - 1, d.hidden, V)
    return W


class CSMRef:
    def __init__(self, cfg: CSMCfg, W, page_size=128, max_pages=64, max_batch=8, policy=None, dry=False):
        """dry: page / position bookkeeping only (see Qwen3Ref)."""
        self.cfg, self.W, self.page_size, self.dry = cfg, W, page_size, dry
        self.policy = policy or Policy()
        b, d = cfg.backbone, cfg.depth
        self.free_pages = list(range(max_pages))
        if dry:
            self.kv = self.dkv = None
            return
        self.backbone = RefStack(b, W, "backbone_model", cfg.max_pos, self.policy)
        self.depth = RefStack(d, W, "depth_decoder.model", 64, self.policy)
        self.kv = [np.zeros((max_pages, 2, page_size, b.kv_heads, b.head_dim), np.uint16) for _ in range(b.layers)]
        self.dkv = [np.zeros((max_batch, 2, cfg.n_codebooks, d.kv_heads, d.head_dim), np.uint16) for _ in range(d.layers)]
        self.free_pages = list(range(max_pages))
        # codebooks_head.weight[i] is [Hd, V] (x @ W): as a linear, rows of W^T
        self.heads = [np.ascontiguousarray(W["depth_decoder.codebooks_head.weight"][i].T) for i in range(cfg.n_codebooks - 1)]

    def _pinned(self, n):
        er = self.policy.exact_rows
        return er is None or n <= er

    def embed(self, ids, masks):
        """sum over the 33 masked embeddings in fp32, columns ascending, one rounding (csm.py:647-653)"""
        cfg, W = self.cfg, self.W
        C, V = cfg.n_codebooks, cfg.vocab
        n = ids.shape[0]
        acc = np.zeros((n, cfg.backbone.hidden), np.float32)
        audio = W["backbone_model.embed_tokens.embed_audio_tokens.weight"]
        for k in range(C):
            e = vr.bf2f(vr.gather(audio[k * V:(k + 1) * V], np.clip(ids[:, k], 0, V - 1)))
            acc = np.where(masks[:, k:k + 1] != 0, acc + e, acc).astype(np.float32)
        t = vr.bf2f(vr.gather(W["embed_text_tokens.weight"], np.clip(ids[:, C], 0, cfg.text_vocab - 1)))
        acc = np.where(masks[:, C:C + 1] != 0, acc + t, acc).astype(np.float32)
        return vr.f2bf(acc)

    def prefill(self, req: RefRequest, ids, masks):
        n, ps = ids.shape[0], self.page_size
        npg = (n + ps - 1) // ps
        req.kv_pages = [self.free_pages.pop(0) for _ in range(npg)]
        req.kv_token_len, req.kv_last_page_len = n, n % ps or ps
        req.next_position_id = n + 1                       # quirk Q1 (worker/base.py:299)
        if self.dry:
            return DRY, DRY
        page = np.array([req.kv_pages[t // ps] for t in range(n)], np.int32)
        slot = np.array([t % ps for t in range(n)], np.int32)
        xs = self.backbone.forward(self.embed(ids, masks), np.arange(n, dtype=np.int32), self.kv, np.zeros(n, np.int32),
                                   np.arange(1, n + 1, dtype=np.int32), np.array([0, npg], np.int32),
                                   np.array(req.kv_pages, np.int32), page, slot, final_norm=False, fixed_order=self._pinned(n))
        hid, logits = self.backbone.norm_head(xs[-1:], self.W["lm_head.weight"], x_out=True, x_rows=True)
        return logits, hid

    def decode(self, reqs: List[RefRequest]):
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
            return DRY, DRY
        ids = np.concatenate([r.input_ids for r in reqs], 0)
        masks = np.concatenate([r.input_mask for r in reqs], 0)
        xs = self.backbone.forward(self.embed(ids, masks), np.array(pos, np.int32), self.kv, np.arange(B, dtype=np.int32),
                                   np.array(kvlen, np.int32), np.array(indptr, np.int32), np.array(indices, np.int32),
                                   np.array(page, np.int32), np.array(slot, np.int32), final_norm=False, fixed_order=self._pinned(B))
        hid, logits = self.backbone.norm_head(xs, self.W["lm_head.weight"], x_out=True)
        return logits, hid

    def depth_loop(self, hid, c0, sampler=None):
        cfg, W = self.cfg, self.W
        C, V, B = cfg.n_codebooks, cfg.vocab, hid.shape[0]
        for l in self.dkv:
            l[:] = 0
        out = np.tile(c0.reshape(B, 1), (1, C + 1)).astype(np.int32)           # output_ids.repeat(1, n_codebooks) (csm.py:697)
        audio = W["backbone_model.embed_tokens.embed_audio_tokens.weight"]
        x = np.stack([hid, vr.gather(audio[:V], c0)], 1).reshape(2 * B, -1)
        indptr, indices = np.arange(B + 1, dtype=np.int32), np.arange(B, dtype=np.int32)
        all_logits = []
        for i in range(1, C):
            if i == 1:
                pos = np.tile(np.array([0, 1], np.int32), B)
                q_req = np.repeat(np.arange(B, dtype=np.int32), 2)
                q_kvlen = np.tile(np.array([1, 2], np.int32), B)
            else:
                pos = np.full(B, i, np.int32)
                q_req = np.arange(B, dtype=np.int32)
                q_kvlen = np.full(B, i + 1, np.int32)
            # engine.hip csm_tail: canonical kernels pinned for a batch of <= exact_rows requests; steps >= 2 gather
            # inputs_embeds_projector(embedding[id]) from a table built with the canonical kernel
            pinned = self._pinned(B)
            tabulated = i >= 2 and C > 2
            op = vr.ORD_CANON if tabulated else self.policy.route(
                Call(B=x.shape[0], N=cfg.depth.hidden, K=cfg.backbone.hidden, fixed_order=pinned, splitk_ws=True))[0]
            xp = vr.linear(W["depth_decoder.model.inputs_embeds_projector.weight"], x, order=op)
            xs = self.depth.forward(xp, pos, self.dkv, q_req, q_kvlen, indptr, indices, q_req.copy(), pos.copy(), final_norm=False,
                                    fixed_order=pinned)
            if i == 1:
                xs = xs[1::2]
            h, logits = self.depth.norm_head(xs, self.heads[i - 1], x_rows=(i == 1))
            all_logits.append(logits)
            ids = vr.argmax(logits) if sampler is None else sampler(logits, i)
            out[:, i] = ids
            x = vr.gather(audio[i * V:(i + 1) * V], ids)                           # embed_audio_tokens_single(ids, i)
        return out, all_logits

    def frame(self, reqs: List[RefRequest], first_logits=None, first_hidden=None, sampler=None):
        logits, hid = (self.decode(reqs) if first_logits is None else (first_logits, first_hidden))
        if self.dry:
            return None, None, None, None
        c0 = vr.argmax(logits) if sampler is None else sampler(logits, 0)
        out, dl = self.depth_loop(hid, c0, sampler)
        C = self.cfg.n_codebooks
        for b, r in enumerate(reqs):
            ids = np.zeros((1, C + 1), np.int32)
            ids[0, :C] = out[b, :C]
            masks = np.ones((1, C + 1), np.uint8)
            masks[0, C] = 0                                                     # csm.py:708-709
            r.input_ids, r.input_mask = ids, masks
            r.frames.append(out[b].copy())
        return out, logits, hid, dl
