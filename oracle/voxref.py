"""ctypes binding of the CPU oracle (oracle/voxref.c).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never by the product package.  bf16 tensors travel as numpy uint16 arrays.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_u16p = ctypes.POINTER(ctypes.c_uint16)
c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_f32p = ctypes.POINTER(ctypes.c_float)


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libvoxref.so")
    src = os.path.join(_HERE, "voxref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libvoxref.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        _LIB.vr_exp2.restype = ctypes.c_float
        _LIB.vr_exp2.argtypes = [ctypes.c_float]
    return _LIB


def _p(a, t):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be contiguous"
    return a.ctypes.data_as(t)


def u16(a):
    return np.ascontiguousarray(a, dtype=np.uint16)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# ---- bf16 helpers (numpy only) -------------------------------------------------------------------
def f2bf(x):
    """fp32 ndarray -> bf16 bits (uint16), round-to-nearest-even."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    # uint32 arithmetic: the rounding add can only wrap for NaN patterns, which are replaced below
    r = ((u + (np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1)))) >> np.uint32(16)).astype(np.uint16)
    nan = (u & np.uint32(0x7FFFFFFF)) > np.uint32(0x7F800000)
    if nan.any():
        r = np.where(nan, ((u >> np.uint32(16)) | np.uint32(0x40)).astype(np.uint16), r)
    return r


_HASH_A, _HASH_B = 2654435761, 2246822507      # odd 32-bit multipliers (Fibonacci hashing); i < 2^30 keeps i * A below 2^63


def hashed_bf16(blk, n, device=None):
    """n bf16 bit patterns drawn from the 2^20-entry block `blk` by a counter recipe: element i takes
    blk[((i * A) mod 2^32) >> 12] with its sign flipped when bit 31 of (i * B) mod 2^32 is set.  Pure 64-bit integer
    arithmetic without overflow, so numpy (device None -> uint16 ndarray) and torch on any device (-> int16 tensor there)
    give the same bits: a GPU test builds its full-size weights on the GPU in milliseconds while the fixture generator built
    the very same weights on the CPU (tests/golden/make_oracle_tapes.py)."""
    assert blk.size == 1 << 20 and n < (1 << 30)
    if device is None:
        out = np.empty(n, np.uint16)

        def chunk(lo):          # uint32 products wrap mod 2^32: the same value as the masked 64-bit product
            i = np.arange(lo, min(n, lo + (1 << 22)), dtype=np.uint32)
            h = i * np.uint32(_HASH_A)
            h >>= np.uint32(12)
            o = out[lo:lo + i.size]
            np.take(blk, h, out=o, mode="wrap")
            i *= np.uint32(_HASH_B)
            i >>= np.uint32(31)
            o ^= i.astype(np.uint16) << np.uint16(15)
        los = range(0, n, 1 << 22)
        if n <= (1 << 24):
            for lo in los:
                chunk(lo)
        else:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max(1, min(16, len(os.sched_getaffinity(0))))) as ex:
                list(ex.map(chunk, los))
        return out
    import torch
    b = torch.from_numpy(blk.view(np.int16).copy()).to(device)
    out = torch.empty(n, dtype=torch.int16, device=device)
    for lo in range(0, n, 1 << 26):
        i = torch.arange(lo, min(n, lo + (1 << 26)), dtype=torch.int64, device=device)
        idx = ((i * _HASH_A) & 0xFFFFFFFF) >> 12
        sgn = ((((i * _HASH_B) & 0xFFFFFFFF) >> 31) << 15).to(torch.int16)      # 0x8000 wraps to the int16 sign bit
        out[lo:lo + i.numel()] = b[idx] ^ sgn
    return out


def random_bf16(rng, shape, std, device=None):
    """N(0, std^2) bf16 bits of the given shape.  Tensors above 2^21 elements (the vocabulary tables and wide projections of the
    full-size test configurations; never the tiny configurations the goldens are generated from) are a 2^20-value normal block
    spread by `hashed_bf16`: drawing and rounding 1.7 G normals costs minutes of single-threaded host time.  With `device` the
    result is a torch bf16 tensor there (large tensors are generated on that device), else a numpy uint16 array."""
    n = int(np.prod(shape))
    if n <= (1 << 21):
        out = f2bf(rng.standard_normal(shape, dtype=np.float32) * np.float32(std))
        return out if device is None else to_torch(out).to(device)
    blk = f2bf(rng.standard_normal(1 << 20, dtype=np.float32) * np.float32(std))
    out = hashed_bf16(blk, n, device)
    if device is None:
        return out.reshape(shape)
    import torch
    return out.view(torch.bfloat16).reshape(*shape)


def bf2f(b):
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def from_torch(t):
    import torch
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def to_torch(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16).copy()).view(torch.bfloat16)


# ---- ops -----------------------------------------------------------------------------------------
# summation orders of a linear (voxref.c): the canonical wave64 DOT of the <= exact_rows kernels, and the three MFMA
# kernels that take calls with more rows
ORD_CANON, ORD_FULLK, ORD_MFMA4, ORD_SPLITK = 0, 1, 2, 3


def linear(W, x, bias=None, residual=None, order=ORD_CANON, act=0):
    W, x = u16(W), u16(x)
    N, K = W.shape
    B = x.shape[0]
    y = np.empty((B, N), np.uint16)
    if order == ORD_CANON and not act:
        lib().vr_linear(_p(W, c_u16p), _p(None if bias is None else u16(bias), c_u16p), _p(x, c_u16p),
                        _p(None if residual is None else u16(residual), c_u16p), _p(y, c_u16p), B, N, K)
    else:
        lib().vr_linear_ord(_p(W, c_u16p), _p(None if bias is None else u16(bias), c_u16p), _p(x, c_u16p),
                            _p(None if residual is None else u16(residual), c_u16p), _p(y, c_u16p), B, N, K, int(order), int(act))
    return y


def linear_silu_mul(Wg, Wu, x, order=ORD_CANON):
    Wg, Wu, x = u16(Wg), u16(Wu), u16(x)
    N, K = Wg.shape
    B = x.shape[0]
    h = np.empty((B, N), np.uint16)
    if order == ORD_CANON:
        lib().vr_linear_silu_mul(_p(Wg, c_u16p), _p(Wu, c_u16p), _p(x, c_u16p), _p(h, c_u16p), B, N, K)
    else:
        lib().vr_linear_silu_mul_ord(_p(Wg, c_u16p), _p(Wu, c_u16p), _p(x, c_u16p), _p(h, c_u16p), B, N, K, int(order))
    return h


def mfma_cases(cases_u8, n, chain=1):
    """Replay a tools/mfma_probe input file through the restated matrix-core arithmetic -> D [n,16,16] fp32."""
    cases_u8 = np.ascontiguousarray(cases_u8, dtype=np.uint8)
    out = np.zeros((n, 16, 16), np.float32)
    lib().vr_mfma_cases(cases_u8.ctypes.data_as(ctypes.c_void_p), int(n), int(chain), _p(out, c_f32p))
    return out


def silu(x):
    x = u16(x)
    y = np.empty_like(x)
    lib().vr_silu(_p(x, c_u16p), _p(y, c_u16p), ctypes.c_long(x.size))
    return y


def add(a, b):
    a, b = u16(a), u16(b)
    y = np.empty_like(a)
    lib().vr_add(_p(a, c_u16p), _p(b, c_u16p), _p(y, c_u16p), ctypes.c_long(a.size))
    return y


def rmsnorm(x, w, eps=1e-6, order=ORD_CANON):
    """order ORD_FULLK: the sum of squares in the order of k_gemm_fullk's fused norm prologue (H % 256 == 0)."""
    x, w = u16(x), u16(w)
    H = x.shape[-1]
    R = x.size // H
    y = np.empty_like(x)
    fn = {ORD_FULLK: lib().vr_rmsnorm_fullk, "fullk": lib().vr_rmsnorm_fullk, "rows1024": lib().vr_rmsnorm_rows1024}.get(order, lib().vr_rmsnorm)
    fn(_p(x, c_u16p), _p(w, c_u16p), _p(y, c_u16p), R, H, ctypes.c_float(eps))
    return y


def rope_table(max_pos, rot, theta, scale=1.0, llama31=None):
    cs = np.empty((max_pos, rot // 2, 2), np.float32)
    lo, hi, ctx = (llama31 if llama31 else (1.0, 4.0, 8192))
    lib().vr_rope_table(_p(cs, c_f32p), max_pos, rot, ctypes.c_double(theta), ctypes.c_double(scale),
                        1 if llama31 else 0, ctypes.c_double(lo), ctypes.c_double(hi), int(ctx))
    return cs


def rope(x, pos, cs, rot=None, interleave=False):
    x = u16(x).copy()
    N, H, D = x.shape
    lib().vr_rope(_p(x, c_u16p), _p(i32(pos), c_i32p), N, H, D, rot or D, 1 if interleave else 0, _p(cs, c_f32p))
    return x


def kv_append(kv, k, v, page, slot):
    """kv [P,2,page_size,Hkv,D] uint16, modified in place."""
    assert kv.dtype == np.uint16 and kv.flags["C_CONTIGUOUS"]
    k, v = u16(k), u16(v)
    N, Hkv, D = k.shape
    lib().vr_kv_append(_p(kv, c_u16p), _p(k, c_u16p), _p(v, c_u16p), _p(i32(page), c_i32p), _p(i32(slot), c_i32p),
                       N, kv.shape[2], Hkv, D)


def paged_attention(q, kv, q_req, q_kvlen, indptr, indices, scale=None):
    q = u16(q)
    Nq, Hq, D = q.shape
    Hkv, page = kv.shape[3], kv.shape[2]
    out = np.empty_like(q)
    if scale is None:
        scale = np.float32(1.0) / np.sqrt(np.float32(D))
    lib().vr_paged_attention(_p(q, c_u16p), _p(kv, c_u16p), _p(i32(q_req), c_i32p), _p(i32(q_kvlen), c_i32p),
                             _p(i32(indptr), c_i32p), _p(i32(indices), c_i32p), Nq, Hq, Hkv, D, page,
                             ctypes.c_float(scale), _p(out, c_u16p))
    return out


def suppress(logits, ids):
    logits = u16(logits).copy()
    B, V = logits.shape
    ids = i32(ids)
    lib().vr_suppress(_p(logits, c_u16p), B, V, _p(ids, c_i32p), len(ids))
    return logits


def rep_penalty(logits, cache, penalty):
    logits = u16(logits).copy()
    cache = np.ascontiguousarray(cache, dtype=np.uint8)
    B, W, C, V = cache.shape
    lib().vr_rep_penalty(_p(logits, c_u16p), _p(cache, c_u8p), B, W, C, V, ctypes.c_float(penalty))
    return logits


def rep_update(cache, ids, window):
    assert cache.dtype == np.uint8 and cache.flags["C_CONTIGUOUS"]
    B, W, C, V = cache.shape
    lib().vr_rep_update(_p(cache, c_u8p), _p(i32(ids), c_i32p), B, W, C, V, int(window))


def argmax(logits):
    logits = u16(logits)
    B, V = logits.shape
    out = np.empty(B, np.int32)
    lib().vr_argmax(_p(logits, c_u16p), B, V, _p(out, c_i32p))
    return out


def sample(logits, top_k=0, top_p=1.0, min_p=0.0, temperature=1.0, seed=0, offset=0, kmax=0):
    logits = u16(logits)
    B, V = logits.shape
    out = np.empty(B, np.int32)
    sup = np.empty((B, kmax), np.int32) if kmax else None
    lib().vr_sample(_p(logits, c_u16p), B, V, int(top_k or 0), ctypes.c_float(top_p), ctypes.c_float(min_p),
                    ctypes.c_float(temperature), ctypes.c_uint64(seed), ctypes.c_uint64(offset),
                    _p(out, c_i32p), _p(sup, c_i32p), int(kmax))
    return (out, sup) if kmax else out


def gather(table, ids):
    table = u16(table)
    ids = i32(ids)
    assert ids.min() >= 0 and ids.max() < table.shape[0], "gather: id out of range"
    y = np.empty((len(ids), table.shape[1]), np.uint16)
    lib().vr_gather(_p(table, c_u16p), _p(ids, c_i32p), _p(y, c_u16p), len(ids), table.shape[1])
    return y


def qwen3_mix(text, codec, mask, feat):
    text, codec, feat = u16(text), u16(codec), u16(feat)
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    y = np.empty_like(text)
    lib().vr_qwen3_mix(_p(text, c_u16p), _p(codec, c_u16p), _p(mask, c_u8p), _p(feat, c_u16p), _p(y, c_u16p),
                       text.shape[0], text.shape[1])
    return y


def exp2(x):
    return float(lib().vr_exp2(ctypes.c_float(x)))
