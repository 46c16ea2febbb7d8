"""Bootstrap for importing the *reference* (vox-serve) on CPU inside the build container.

TEST INFRASTRUCTURE ONLY.  This module is used by ``tests/golden/make_goldens.py`` to generate the
committed ``.npz`` fixtures.  It needs ``/root/reference`` and therefore never runs on the GPU box;
nothing under ``vox_serve_amd/`` imports it.

The reference's only native dependency on the hot path is the ``flashinfer`` wheel (CUDA-only, absent
here).  The stand-in below is OUR torch-fp32 restatement of the eight entry points' published
mathematical contracts; the reference's own Python (page/slot math in
``FlashInfer*Wrapper.plan``, ``set_kv_cache``, ``Sampler``, the model ``nn.Module``s and the codec)
then runs unmodified on top of it.
"""
import importlib.machinery
import math
import os
import sys
import types

REF_ROOT = "/root/reference"


def have_reference() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "vox_serve"))


def _fake(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    sys.modules[name] = m
    if "." in name:
        parent, child = name.rsplit(".", 1)
        setattr(sys.modules[parent], child, m)
    return m


# --------------------------------------------------------------------------------------------------
# flashinfer stand-in (torch fp32 maths, single final rounding to the storage dtype)
# --------------------------------------------------------------------------------------------------
def _build_flashinfer_standin():
    import torch

    def rmsnorm(input, weight, eps=1e-6):
        x = input.float()
        y = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * weight.float()
        return y.to(input.dtype)

    def _rope(x, pos, rotary_dim, interleave, freqs):
        # x [N,H,D]; freqs [rotary_dim/2] fp32
        D = x.shape[-1]
        d = rotary_dim or D
        xf = x.float()
        ang = pos.float()[:, None] * freqs[None, :]          # [N, d/2]
        c, s = torch.cos(ang)[:, None, :], torch.sin(ang)[:, None, :]
        rot = xf[..., :d]
        if interleave:
            a, b = rot[..., 0::2], rot[..., 1::2]
            ra, rb = a * c - b * s, b * c + a * s
            out = torch.stack([ra, rb], dim=-1).flatten(-2)
        else:
            a, b = rot[..., : d // 2], rot[..., d // 2:]
            out = torch.cat([a * c - b * s, b * c + a * s], dim=-1)
        return torch.cat([out, xf[..., d:]], dim=-1).to(x.dtype)

    def apply_rope_pos_ids(q, k, pos_ids, rotary_dim=None, interleave=False, rope_scale=1.0, rope_theta=1e4):
        d = rotary_dim or q.shape[-1]
        i = torch.arange(0, d, 2, dtype=torch.float32)
        freqs = 1.0 / (rope_scale * (rope_theta ** (i / d)))
        return _rope(q, pos_ids, d, interleave, freqs), _rope(k, pos_ids, d, interleave, freqs)

    def apply_llama31_rope_pos_ids(q, k, pos_ids, rotary_dim=None, interleave=False, rope_scale=8.0,
                                   rope_theta=5e5, low_freq_factor=1.0, high_freq_factor=4.0,
                                   old_context_len=8192):
        d = rotary_dim or q.shape[-1]
        i = torch.arange(0, d, 2, dtype=torch.float32)
        f = 1.0 / (rope_theta ** (i / d))
        smooth = (f * old_context_len / (2 * math.pi) - low_freq_factor) / (high_freq_factor - low_freq_factor)
        smooth = smooth.clamp(0.0, 1.0)
        f = (1 - smooth) * (f / rope_scale) + smooth * f
        return _rope(q, pos_ids, d, interleave, f), _rope(k, pos_ids, d, interleave, f)

    def _attend(q, K, V):
        # q [Hq,D], K/V [T,Hkv,D] -> [Hq,D]; fp32 softmax, single rounding
        Hq, D = q.shape
        g = Hq // K.shape[1]
        Kf = K.float().repeat_interleave(g, dim=1)
        Vf = V.float().repeat_interleave(g, dim=1)
        s = torch.einsum("hd,thd->ht", q.float(), Kf) / math.sqrt(D)
        p = torch.softmax(s, dim=-1)
        return torch.einsum("ht,thd->hd", p, Vf)

    def _gather_kv(kv, pages, last, page):
        ks, vs = [], []
        for j, pg in enumerate(pages):
            n = last if j == len(pages) - 1 else page
            ks.append(kv[pg, 0, :n])
            vs.append(kv[pg, 1, :n])
        return torch.cat(ks, 0), torch.cat(vs, 0)

    class BatchDecodeWithPagedKVCacheWrapper:
        def __init__(self, *a, **kw):
            pass

        def plan(self, indptr, indices, last_page_len, num_qo_heads, num_kv_heads, head_dim, page_size, **kw):
            self.indptr, self.indices, self.last = indptr.tolist(), indices.tolist(), last_page_len.tolist()
            self.page = page_size

        def run(self, q, kv):
            out = torch.empty_like(q)
            for r in range(len(self.last)):
                pages = self.indices[self.indptr[r]: self.indptr[r + 1]]
                K, V = _gather_kv(kv, pages, self.last[r], self.page)
                out[r] = _attend(q[r], K, V).to(q.dtype)
            return out

    class BatchPrefillWithPagedKVCacheWrapper:
        def __init__(self, *a, **kw):
            pass

        def plan(self, qo_indptr, paged_kv_indptr, paged_kv_indices, paged_kv_last_page_len, num_qo_heads,
                 num_kv_heads, head_dim_qk, page_size, causal=True, **kw):
            self.qo = qo_indptr.tolist()
            self.indptr, self.indices = paged_kv_indptr.tolist(), paged_kv_indices.tolist()
            self.last, self.page, self.causal = paged_kv_last_page_len.tolist(), page_size, causal

        def run(self, q, kv):
            out = torch.empty_like(q)
            for r in range(len(self.last)):
                pages = self.indices[self.indptr[r]: self.indptr[r + 1]]
                K, V = _gather_kv(kv, pages, self.last[r], self.page)
                n = K.shape[0]
                m = self.qo[r + 1] - self.qo[r]
                for i in range(m):
                    vis = n - m + i + 1 if self.causal else n
                    out[self.qo[r] + i] = _attend(q[self.qo[r] + i], K[:vis], V[:vis]).to(q.dtype)
            return out

    # Sampling: the RNG stream is flashinfer-internal (parity unpinned). The stand-in draws
    # with torch.multinomial so the *support set* is the published one; only greedy is used for goldens.
    def top_k_sampling_from_probs(probs, top_k, deterministic=True):
        v, i = torch.topk(probs.float(), top_k, dim=-1)
        j = torch.multinomial(v / v.sum(-1, keepdim=True), 1)
        return i.gather(-1, j).squeeze(-1).int()

    def top_p_sampling_from_probs(probs, top_p, deterministic=True):
        sp, si = torch.sort(probs.float(), dim=-1, descending=True)
        keep = (sp.cumsum(-1) - sp) < top_p
        sp = sp * keep
        j = torch.multinomial(sp / sp.sum(-1, keepdim=True), 1)
        return si.gather(-1, j).squeeze(-1).int()

    def min_p_sampling_from_probs(probs, min_p, deterministic=True):
        p = probs.float()
        p = p * (p >= min_p * p.max(-1, keepdim=True).values)
        return torch.multinomial(p / p.sum(-1, keepdim=True), 1).squeeze(-1).int()

    def top_k_top_p_sampling_from_logits(logits, top_k, top_p, filter_apply_order="top_k_first", deterministic=True):
        l = logits.float()
        kth = torch.topk(l, top_k, dim=-1).values[..., -1:]
        l = l.masked_fill(l < kth, float("-inf"))
        return top_p_sampling_from_probs(torch.softmax(l, -1), top_p)

    fi = _fake("flashinfer",
               BatchDecodeWithPagedKVCacheWrapper=BatchDecodeWithPagedKVCacheWrapper,
               BatchPrefillWithPagedKVCacheWrapper=BatchPrefillWithPagedKVCacheWrapper)
    _fake("flashinfer.norm", rmsnorm=rmsnorm)
    _fake("flashinfer.rope", apply_rope_pos_ids=apply_rope_pos_ids,
          apply_llama31_rope_pos_ids=apply_llama31_rope_pos_ids)
    _fake("flashinfer.sampling", top_k_sampling_from_probs=top_k_sampling_from_probs,
          top_p_sampling_from_probs=top_p_sampling_from_probs,
          min_p_sampling_from_probs=min_p_sampling_from_probs,
          top_k_top_p_sampling_from_logits=top_k_top_p_sampling_from_logits)
    return fi


class _Permissive(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        v = type(k, (), {"__init__": lambda self, *a, **kw: None})
        setattr(self, k, v)
        return v


def _fake_permissive(name):
    m = _Permissive(name)
    m.__path__ = []
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    sys.modules[name] = m
    if "." in name:
        parent, child = name.rsplit(".", 1)
        setattr(sys.modules[parent], child, m)
    return m


_BOOTED = None


def _build_diffusers_standin():
    """diffusers==0.34.0 (pyproject.toml:34; not installed here) is used by tokenizer/glm.py:1563 for ONE class,
    `diffusers.models.attention_processor.Attention`, as a bias-free self-attention of the GLM estimator's transformer blocks.  The
    stand-in is OUR restatement of its published default path (AttnProcessor2_0): to_q / to_k / to_v (bias as given), heads split,
    torch scaled_dot_product_attention, to_out = [Linear(inner, query_dim), Dropout] — same parameter names as the checkpoint."""
    import torch
    import torch.nn.functional as F
    from torch import nn

    class Attention(nn.Module):
        def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False, upcast_attention=False, **kw):
            super().__init__()
            inner = dim_head * heads
            kv_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
            self.heads = heads
            self.to_q = nn.Linear(query_dim, inner, bias=bias)
            self.to_k = nn.Linear(kv_dim, inner, bias=bias)
            self.to_v = nn.Linear(kv_dim, inner, bias=bias)
            self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(dropout)])

        def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
            B, T, _ = hidden_states.shape
            ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
            q, k, v = self.to_q(hidden_states), self.to_k(ctx), self.to_v(ctx)
            hd = q.shape[-1] // self.heads
            q, k, v = (t.view(B, -1, self.heads, hd).transpose(1, 2) for t in (q, k, v))
            o = F.scaled_dot_product_attention(q, k, v, attn_mask=attention_mask, dropout_p=0.0, is_causal=False)
            o = o.transpose(1, 2).reshape(B, -1, self.heads * hd).to(q.dtype)
            return self.to_out[1](self.to_out[0](o))

    _fake("diffusers")
    _fake("diffusers.models")
    _fake("diffusers.models.attention_processor", Attention=Attention)


def boot():
    """Import the reference's hot-path modules on CPU.  Returns a namespace of modules."""
    global _BOOTED
    if _BOOTED is not None:
        return _BOOTED
    if not have_reference():
        raise RuntimeError("reference tree not present (golden generation runs only in the build container)")
    os.environ["TORCHDYNAMO_DISABLE"] = "1"      # Sampler methods are @torch.compile'd (sampling.py:121,149)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)

    # 1. real imports first
    from vox_serve.tokenizer import qwen3_codec, mimi, snac     # noqa
    qwen3_codec.ROPE_INIT_FUNCTIONS = None
    from transformers import (AutoTokenizer, AutoModel, LlamaConfig, LlamaPreTrainedModel,  # noqa
                              CsmConfig, CsmDepthDecoderConfig, CsmPreTrainedModel)

    # 2. stand-ins
    _build_flashinfer_standin()
    _build_diffusers_standin()
    for n in ("librosa", "librosa.filters", "torchaudio", "torchaudio.functional", "torchaudio.transforms",
              "torchaudio.compliance", "torchaudio.compliance.kaldi", "onnxruntime", "onnx", "zmq",
              "zmq.asyncio", "inflect", "tiktoken"):
        _fake_permissive(n)
    z = sys.modules["zmq"]
    for i, k in enumerate(("NOBLOCK", "DONTWAIT", "PULL", "PUSH", "RCVHWM", "SNDHWM", "LINGER")):
        setattr(z, k, i)
    z.Again = type("Again", (Exception,), {})

    # 3. namespace shim for vox_serve.model (its __init__ imports all 8 families)
    import vox_serve  # noqa
    ns = types.ModuleType("vox_serve.model")
    ns.__path__ = [os.path.join(REF_ROOT, "vox_serve", "model")]
    ns.__spec__ = importlib.machinery.ModuleSpec("vox_serve.model", None, is_package=True)
    ns.load_model = None
    sys.modules["vox_serve.model"] = ns

    from vox_serve import sampling, flashinfer_utils, requests as vreq
    from vox_serve.model import qwen3_tts
    from vox_serve.worker.base import ModelWorker

    _BOOTED = types.SimpleNamespace(
        sampling=sampling, flashinfer_utils=flashinfer_utils, requests=vreq, qwen3_tts=qwen3_tts,
        qwen3_codec=qwen3_codec, mimi=mimi, snac=snac, ModelWorker=ModelWorker)
    return _BOOTED
