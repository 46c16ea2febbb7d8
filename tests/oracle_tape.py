"""Oracle tapes: the CPU oracle's side of a heavy parity test, recorded ahead of time.

TEST INFRASTRUCTURE.  The bit-exact parity tests compare every tensor the HIP engine produces with the CPU oracle
(oracle/).  For the full-size configurations (Qwen3-TTS-1.7B at 32 requests, GLM-4-Voice width with its 168 960-entry
vocabulary, CSM-1B at 16 requests ...) the oracle's integer restatement of the matrix cores costs minutes of host time per
case — more than the whole GPU suite may take on the GPU box's 16-CPU share.  Those cases therefore run the oracle ONCE, in
the build container (`python tests/golden/make_oracle_tapes.py`), and keep what it produced as a *tape*: for every
comparison point of the test, in order, the SHA-256 of the oracle's array (its shape and dtype included; small integer
arrays — token ids — verbatim).  On the GPU box the same test body runs with the oracle in `dry` mode (page / position
bookkeeping only) and compares the digest of each GPU array with the tape: bit-exactness is what is tested either way.
Weights and inputs are regenerated from their seeds on both sides (oracle/voxref.py: random_bf16 / hashed_bf16).

A test body talks to a Tape only:
    tape.gpu      run the engine side            (False while recording: no GPU in the build container)
    tape.oracle   run the oracle's arithmetic    (False while replaying)
    tape.check(key, got, want)   got / want: zero-argument callables -> ndarray (only the needed one is called)

Modes: "live" (both sides run, arrays compared directly — tiny configurations, or VOX_ORACLE_LIVE=1 to debug a full-size
mismatch on a GPU box), "record", "replay".  A tape remembers the hash of the oracle's sources; tests/test_oracle_tapes.py
(CPU) fails when the oracle changed after a tape was recorded.
"""
import hashlib
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAPE_DIR = os.path.join(ROOT, "tests", "golden", "tapes")
ORACLE_SOURCES = ["voxref.c", "voxref.py", "policy.py", "qwen3_ref.py", "lm_ref.py", "csm_ref.py"]
VERBATIM_MAX = 1024          # integer arrays up to this many elements are stored as values, not digests


def oracle_sources_sha() -> str:
    h = hashlib.sha256()
    for f in ORACLE_SOURCES:
        h.update(open(os.path.join(ROOT, "oracle", f), "rb").read())
    return h.hexdigest()[:24]


def digest(a) -> dict:
    a = np.ascontiguousarray(a)
    d = {"shape": list(a.shape), "dtype": str(a.dtype)}
    if a.dtype.kind in "iu" and a.dtype.itemsize >= 4 and a.size <= VERBATIM_MAX:
        d["values"] = a.reshape(-1).tolist()
    else:
        d["sha256"] = hashlib.sha256(a.tobytes()).hexdigest()
    return d


class Tape:
    def __init__(self, case=None, mode="live"):
        assert mode in ("live", "record", "replay")
        self.case, self.mode, self.entries, self.i, self.meta = case, mode, [], 0, {}
        if mode == "replay":
            with open(self.path(case)) as f:
                d = json.load(f)
            self.entries, self.meta = d["entries"], d["meta"]

    @staticmethod
    def path(case):
        return os.path.join(TAPE_DIR, case + ".json")

    @classmethod
    def open(cls, case):
        """The mode a GPU test runs a heavy case in: replay when its tape exists (and VOX_ORACLE_LIVE is not set), else live."""
        if os.environ.get("VOX_ORACLE_LIVE") != "1" and os.path.exists(cls.path(case)):
            return cls(case, "replay")
        return cls(case, "live")

    gpu = property(lambda self: self.mode != "record")
    oracle = property(lambda self: self.mode != "replay")

    def check(self, key, got, want):
        if self.mode == "live":
            g, w = np.asarray(got()), np.asarray(want())
            assert g.shape == w.shape and np.array_equal(g, w), f"{self.case or ''} {key}: GPU differs from the oracle"
        elif self.mode == "record":
            self.entries.append([key, digest(want())])
        else:
            assert self.i < len(self.entries), f"{self.case} {key}: tape exhausted (test and tape out of step)"
            k, d = self.entries[self.i]
            self.i += 1
            assert k == key, f"{self.case}: tape has '{k}' where the test checks '{key}' (regenerate the tape)"
            g = np.ascontiguousarray(got())
            assert list(g.shape) == d["shape"] and str(g.dtype) == d["dtype"], f"{self.case} {key}: {g.shape} {g.dtype} vs tape {d['shape']} {d['dtype']}"
            if "values" in d:
                w = np.array(d["values"], dtype=g.dtype).reshape(g.shape)
                assert np.array_equal(g, w), f"{self.case} {key}: GPU {g.reshape(-1)[:16].tolist()}... vs oracle tape {w.reshape(-1)[:16].tolist()}..."
            else:
                assert hashlib.sha256(g.tobytes()).hexdigest() == d["sha256"], f"{self.case} {key}: GPU bits differ from the oracle tape"

    def done(self, **meta):
        if self.mode == "replay":
            assert self.i == len(self.entries), f"{self.case}: {len(self.entries) - self.i} tape entries not reached"
        elif self.mode == "record":
            os.makedirs(TAPE_DIR, exist_ok=True)
            m = dict(meta, oracle_sources_sha=oracle_sources_sha(), n_checks=len(self.entries))
            with open(self.path(self.case), "w") as f:
                json.dump({"meta": m, "entries": self.entries}, f, separators=(",", ":"))


class Weights:
    """Seeded weights of a test case, materialised where they are needed: numpy bit arrays for the oracle, torch bf16
    tensors on the GPU for the engine (generated there when the oracle side is not run).  fn(device=None|dev) -> dict."""

    def __init__(self, fn_or_dict):
        self._fn = None if isinstance(fn_or_dict, dict) else fn_or_dict
        self._np = fn_or_dict if isinstance(fn_or_dict, dict) else None
        self._t = {}

    def numpy(self):
        if self._np is None:
            self._np = self._fn(device=None)
        return self._np

    def torch(self, dev):
        key = str(dev)
        if key not in self._t:
            if self._np is not None:
                from oracle import voxref as vr
                self._t[key] = {k: vr.to_torch(v).to(dev) for k, v in self._np.items()}
            else:
                self._t[key] = self._fn(device=dev)
        return self._t[key]

    def drop_torch(self):
        self._t.clear()
