"""Drop-in for /root/reference/vox_serve/flashinfer_utils.py on MI355X — same names, arguments, public
attributes and error behaviour; the arithmetic runs in libvoxhip (hand-written gfx950 kernels).  No
FlashInfer, no Triton.

  rms_norm                    flashinfer_utils.py:251-267
  apply_rope_pos_ids          flashinfer_utils.py:270-324 (kwargs routing to the llama-3.1 variant kept)
  FlashInferPrefillWrapper    flashinfer_utils.py:11-145  (plan / run / set_kv_cache, .qo_indptr,
                              .token_to_page, .token_to_cache, .head_dim)
  FlashInferDecodeWrapper     flashinfer_utils.py:149-244 (plan / run / set_kv_cache, .kv_cache_locations)

plan() takes CPU int32 tensors like the reference and uploads them with an async copy; unlike the
reference it needs no device synchronisation afterwards (the caller's `torch.cuda.synchronize()` is
harmless but unnecessary).
"""
from typing import Union

import torch

from . import _native as N

_ROPE_TABLES = {}


def _rope_table(device, max_pos, rot, theta, scale, llama31):
    key = (str(device), max_pos, rot, float(theta), float(scale), llama31)
    t = _ROPE_TABLES.get(key)
    if t is None:
        from .engine import StackCfg, rope_table
        c = StackCfg(0, 0, 0, 0, rot, 0, rope_theta=theta, rope_scale=scale, rope_dim=rot, rope_llama31=llama31)
        t = _ROPE_TABLES[key] = rope_table(max_pos, c, device)
    return t


def rms_norm(hidden_states: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    x = hidden_states.contiguous()
    y = torch.empty_like(x)
    cols = x.shape[-1]
    N.check(N.lib().vox_rmsnorm(N.ctx(), N.stream(), N.ptr(x), N.ptr(weight.contiguous()), N.ptr(y),
                                x.numel() // cols, cols, float(eps)))
    return y


def apply_rope_pos_ids(query_states, key_states, position_ids, rope_scale: float = 1.0, rope_theta: float = 10000.0,
                       interleave: bool = False, **kwargs):
    llama31 = {k: kwargs[k] for k in ("low_freq_factor", "high_freq_factor", "old_context_len") if k in kwargs}
    other = {k: v for k, v in kwargs.items() if k not in llama31}
    unknown = set(other) - {"rotary_dim"}
    if unknown:
        raise TypeError(f"apply_rope_pos_ids() got unexpected keyword arguments {sorted(unknown)}")
    q, k = query_states.contiguous(), key_states.contiguous()
    n, hq, d = q.shape
    hkv = k.shape[1]
    rot = other.get("rotary_dim") or d
    l31 = None
    if llama31:
        l31 = (float(llama31.get("low_freq_factor", 1.0)), float(llama31.get("high_freq_factor", 4.0)),
               int(llama31.get("old_context_len", 8192)))
    max_pos = 8192
    cs = _rope_table(q.device, max_pos, rot, rope_theta, rope_scale, l31)
    qo, ko = torch.empty_like(q), torch.empty_like(k)
    pos = position_ids.to(device=q.device, dtype=torch.int32).contiguous()
    N.check(N.lib().vox_rope(N.ctx(), N.stream(), N.ptr(q), N.ptr(k), N.ptr(qo), N.ptr(ko), N.ptr(pos), n, hq, hkv, d,
                             rot, int(bool(interleave)), N.ptr(cs), max_pos))
    return qo, ko


class _PagedBase:
    def _common(self, n_qo_head, n_kv_head, n_state, page_size, device):
        self.device = torch.device(device)
        self.n_qo_head, self.n_kv_head, self.n_state = n_qo_head, n_kv_head, n_state
        self.head_dim = n_state // n_qo_head
        self.page_size = page_size
        self._ws = None
        self._meta = None

    def _workspace(self, nq, max_kvlen):
        need = N.lib().vox_attn_workspace_bytes(nq, self.n_qo_head, self.head_dim, max_kvlen)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(int(need), dtype=torch.uint8, device=self.device)
        return self._ws

    def _run(self, q, kv_cache):
        q = q.contiguous()
        m = self._meta
        out = torch.empty_like(q)
        ws = self._workspace(q.shape[0], m["max_kvlen"])
        N.check(N.lib().vox_paged_attention(
            N.ctx(), N.stream(), N.ptr(q), N.ptr(kv_cache), N.ptr(m["q_req"]), N.ptr(m["q_kvlen"]), N.ptr(m["indptr"]),
            N.ptr(m["indices"]), N.ptr(out), N.ptr(ws), q.shape[0], self.n_qo_head, self.n_kv_head, self.head_dim,
            kv_cache.shape[2], m["max_kvlen"], float(self.head_dim) ** -0.5))
        return out

    def _append(self, kv_cache, k, v, page, slot):
        k, v = k.contiguous(), v.contiguous()
        N.check(N.lib().vox_kv_append(N.ctx(), N.stream(), N.ptr(kv_cache), N.ptr(k), N.ptr(v), N.ptr(page),
                                      N.ptr(slot), k.shape[0], kv_cache.shape[2], self.n_kv_head, self.head_dim))


class FlashInferPrefillWrapper(_PagedBase):
    def __init__(self, attn_buffer=None, n_qo_head=None, n_kv_head=None, n_state=None, page_size=None, batch_size=None,
                 max_seq_len=None, device=torch.device("cuda"), qo_indptr_buffer=None, paged_kv_indptr_buffer=None,
                 paged_kv_indices_buffer=None, paged_kv_last_page_len_buffer=None, use_cuda_graph=False):
        self.use_cuda_graph, self.batch_size, self.max_seq_len = use_cuda_graph, batch_size, max_seq_len
        if use_cuda_graph:
            assert batch_size is not None, "batch_size must be specified for cuda graph optimization"
            assert max_seq_len is not None, "max_seq_len must be specified for cuda graph optimization"
        self._common(n_qo_head, n_kv_head, n_state, page_size, device)
        if use_cuda_graph:
            self.token_to_page = torch.zeros(max_seq_len, dtype=torch.long, device=self.device)
            self.token_to_cache = torch.zeros(max_seq_len, dtype=torch.long, device=self.device)

    def plan(self, qo_indptr, paged_kv_indptr, paged_kv_indices, paged_kv_last_page_len, dtype=torch.float16):
        self.qo_indptr, self.paged_kv_indptr = qo_indptr, paged_kv_indptr
        self.paged_kv_indices, self.paged_kv_last_page_len = paged_kv_indices, paged_kv_last_page_len
        n_req = qo_indptr.shape[0] - 1
        ps = self.page_size
        lens = (qo_indptr[1:] - qo_indptr[:-1]).to(torch.int32)
        total = int(lens.sum())
        pages_after = (paged_kv_indptr[1:] - paged_kv_indptr[:-1]).to(torch.int32)
        kv_len = (pages_after - 1) * ps + paged_kv_last_page_len.to(torch.int32)
        seg = torch.repeat_interleave(torch.arange(n_req, dtype=torch.int32), lens)
        intra = torch.arange(total, dtype=torch.int32) - torch.repeat_interleave(qo_indptr[:-1].to(torch.int32), lens)
        g = kv_len[seg] - lens[seg] + intra                                  # absolute KV index of each new token
        page_off = torch.div(g, ps, rounding_mode="floor").to(torch.int32)
        tok_page = paged_kv_indices[(paged_kv_indptr[:-1])[seg] + page_off]
        tok_slot = (g - page_off * ps).to(torch.int32)
        dev = self.device
        if self.use_cuda_graph:
            self.token_to_page[:total] = tok_page.to(dev)
            self.token_to_cache[:total] = tok_slot.to(dev)
            self.token_to_page[total:] = -1
            self.token_to_cache[total:] = -1
        else:
            self.token_to_page = tok_page.to(dev)
            self.token_to_cache = tok_slot.to(dev)
        nq = self.token_to_page.shape[0]
        page32 = self.token_to_page.to(torch.int32)
        slot32 = self.token_to_cache.to(torch.int32)
        q_req = torch.zeros(nq, dtype=torch.int32)
        q_kvlen = torch.zeros(nq, dtype=torch.int32)          # padding rows (graph mode): kvlen 0 => skipped
        q_req[:total] = seg
        q_kvlen[:total] = g + 1                                # causal, right-aligned (flashinfer_utils.py:68-80)
        self._meta = dict(q_req=q_req.to(dev), q_kvlen=q_kvlen.to(dev), page=page32, slot=slot32,
                          indptr=paged_kv_indptr.to(device=dev, dtype=torch.int32),
                          indices=paged_kv_indices.to(device=dev, dtype=torch.int32),
                          max_kvlen=int(kv_len.max()) if n_req else 1)

    def run(self, q, kv_cache):
        return self._run(q, kv_cache)

    def set_kv_cache(self, kv_cache, k, v):
        self._append(kv_cache, k, v, self._meta["page"], self._meta["slot"])


class FlashInferDecodeWrapper(_PagedBase):
    def __init__(self, attn_buffer=None, n_qo_head=None, n_kv_head=None, n_state=None, page_size=None, batch_size=None,
                 device=torch.device("cuda"), paged_kv_indptr_buffer=None, paged_kv_indices_buffer=None,
                 paged_kv_last_page_len_buffer=None, use_cuda_graph=False, use_tensor_cores=True):
        self.use_cuda_graph, self.batch_size = use_cuda_graph, batch_size
        self._common(n_qo_head, n_kv_head, n_state, page_size, device)
        if use_cuda_graph:
            assert batch_size is not None, "batch_size must be specified for cuda graph optimization"
            self.kv_cache_locations = torch.zeros((batch_size, 2), dtype=torch.long, device=self.device)

    def plan(self, paged_kv_indptr, paged_kv_indices, paged_kv_last_page_len, dtype=torch.float16):
        self.batch_size = paged_kv_indptr.shape[0] - 1
        self.paged_kv_indptr, self.paged_kv_indices = paged_kv_indptr, paged_kv_indices
        self.paged_kv_last_page_len = paged_kv_last_page_len
        page_idx = paged_kv_indices[paged_kv_indptr[1:] - 1]
        pos_idx = paged_kv_last_page_len - 1
        loc = torch.stack([page_idx, pos_idx], dim=1)
        dev = self.device
        if self.use_cuda_graph:
            self.kv_cache_locations[: self.batch_size].copy_(loc)
        else:
            self.kv_cache_locations = loc.to(dev)
        n_pages = (paged_kv_indptr[1:] - paged_kv_indptr[:-1]).to(torch.int32)
        kv_len = (n_pages - 1) * self.page_size + paged_kv_last_page_len.to(torch.int32)
        b = self.batch_size
        self._meta = dict(q_req=torch.arange(b, dtype=torch.int32, device=dev), q_kvlen=kv_len.to(dev),
                          page=page_idx.to(device=dev, dtype=torch.int32), slot=pos_idx.to(device=dev, dtype=torch.int32),
                          indptr=paged_kv_indptr.to(device=dev, dtype=torch.int32),
                          indices=paged_kv_indices.to(device=dev, dtype=torch.int32),
                          max_kvlen=int(kv_len.max()) if b else 1)

    def run(self, q, kv_cache):
        return self._run(q, kv_cache)

    def set_kv_cache(self, kv_cache, k, v):
        self._append(kv_cache, k, v, self._meta["page"], self._meta["slot"])


FlashInferWrapper = Union[FlashInferPrefillWrapper, FlashInferDecodeWrapper]
