"""DecoderCache protocol (drop-in for /root/reference/vox_serve/tokenizer/base.py:7-173).

The reference carries every codec's streaming state as nested tensors inside a dataclass and slices /
concatenates / copies it per request per chunk.  Here the state lives in place inside the native codec
engine, keyed by a slot id; a DecoderCache is a light handle and the same four operations keep working.
"""
import functools
import inspect
from dataclasses import dataclass, fields
from typing import Any, List

import torch


def device_bound(cls):
    """Class decorator for the native detokenizers: the constructor and every public method run with the tokenizer's own device
    current, so that its libvoxhip context (`_native.ctx()` is per device), workspace allocations, streams and graphs belong to
    that GPU whatever device the caller is on — the reference's `audio_decoder_device` (worker/base.py:55-78): the LM on one
    GPU, the detokenizer on another.  With one GPU (or an index-less device) the guard is a no-op."""
    from .. import _native as N
    init = cls.__init__
    sig = inspect.signature(init)

    @functools.wraps(init)
    def __init__(self, *a, **k):
        bound = sig.bind(self, *a, **k)
        bound.apply_defaults()
        with N.device_guard(bound.arguments.get("device", "cuda")):
            init(self, *a, **k)
    cls.__init__ = __init__

    def wrap(fn):
        @functools.wraps(fn)
        def method(self, *a, **k):
            with N.device_guard(getattr(self, "device", "cuda")):
                return fn(self, *a, **k)
        return method
    for name, fn in list(vars(cls).items()):
        if inspect.isfunction(fn) and (not name.startswith("_") or name == "__call__"):
            setattr(cls, name, wrap(fn))
    return cls


@dataclass
class DecoderCache:
    def __getitem__(self, index: Any):
        def _slice(obj):
            if torch.is_tensor(obj):
                return obj[index]
            if isinstance(obj, DecoderCache):
                return obj[index]
            if isinstance(obj, list):
                return [_slice(x) for x in obj]
            if isinstance(obj, tuple):
                return tuple(_slice(x) for x in obj)
            if isinstance(obj, dict):
                return {k: _slice(v) for k, v in obj.items()}
            return obj
        return type(self)(**{f.name: _slice(getattr(self, f.name)) for f in fields(self)})

    @torch.no_grad()
    def copy_from(self, src: "DecoderCache") -> None:
        if type(self) is not type(src):
            raise TypeError(f"Cannot copy from {type(src)} to {type(self)}")

        def _copy(dst, s):
            if dst is None and s is None:
                return
            if torch.is_tensor(dst) and torch.is_tensor(s):
                dst.copy_(s)
            elif isinstance(dst, DecoderCache) and isinstance(s, DecoderCache):
                dst.copy_from(s)
            elif isinstance(dst, (list, tuple)) and isinstance(s, (list, tuple)):
                if len(dst) != len(s):
                    raise ValueError(f"List length mismatch: {len(dst)} vs {len(s)}")
                for d, x in zip(dst, s):
                    _copy(d, x)
            elif isinstance(dst, dict) and isinstance(s, dict):
                if dst.keys() != s.keys():
                    raise ValueError(f"Dict keys mismatch: {dst.keys()} vs {s.keys()}")
                for k in dst:
                    _copy(dst[k], s[k])
            elif dst is None or s is None:
                if dst != s:
                    raise ValueError("Cannot copy non-None to None or vice versa")
        for f in fields(self):
            _copy(getattr(self, f.name), getattr(src, f.name))

    @classmethod
    def cat(cls, caches: List["DecoderCache"]) -> "DecoderCache":
        if not caches:
            raise ValueError("caches must be a non-empty list")
        ct = type(caches[0])
        if not all(isinstance(c, ct) for c in caches):
            raise TypeError("All caches must be instances of the same cache class")

        def _merge(vals):
            first = vals[0]
            if torch.is_tensor(first):
                return torch.cat(vals, dim=0)
            if isinstance(first, DecoderCache):
                return type(first).cat(vals)
            if isinstance(first, list):
                return [_merge([v[i] for v in vals]) for i in range(len(first))]
            if isinstance(first, tuple):
                return tuple(_merge([v[i] for v in vals]) for i in range(len(first)))
            if isinstance(first, dict):
                return {k: _merge([v[k] for v in vals]) for k in first}
            if all(v is None for v in vals):
                return None
            if all(v == first for v in vals):
                return first
            raise TypeError(f"Unsupported or mismatched member type for merging: {type(first)}")
        return ct(**{f.name: _merge([getattr(c, f.name) for c in caches]) for f in fields(ct)})

    def to(self, device) -> "DecoderCache":
        def _to(obj):
            if torch.is_tensor(obj):
                return obj.to(device)
            if isinstance(obj, DecoderCache):
                return obj.to(device)
            if isinstance(obj, list):
                return [_to(x) for x in obj]
            if isinstance(obj, tuple):
                return tuple(_to(x) for x in obj)
            if isinstance(obj, dict):
                return {k: _to(v) for k, v in obj.items()}
            return obj
        return type(self)(**{f.name: _to(getattr(self, f.name)) for f in fields(self)})
