"""CPU: the oracle tapes (tests/oracle_tape.py) are present and fresh, the dry oracle keeps the live oracle's bookkeeping, and
the counter-hashed weight recipe gives the same bits in numpy and in torch."""
import json
import os

import numpy as np
import pytest

from tests.oracle_tape import Tape, digest, oracle_sources_sha


def _cases():
    from tests.golden.make_oracle_tapes import all_cases
    return all_cases()


def test_every_taped_case_has_a_fresh_tape():
    sha = oracle_sources_sha()
    missing, stale = [], []
    for name in _cases():
        p = Tape.path(name)
        if not os.path.exists(p):
            missing.append(name)
            continue
        d = json.load(open(p))
        assert d["meta"]["n_checks"] == len(d["entries"]) > 0, name
        if d["meta"]["oracle_sources_sha"] != sha:
            stale.append(name)
    assert not missing and not stale, (f"run `python tests/golden/make_oracle_tapes.py`: missing {missing}, recorded before the last "
                                       f"change to the oracle's sources {stale}")


def test_tape_record_and_replay_round_trip(tmp_path, monkeypatch):
    import tests.oracle_tape as OT
    monkeypatch.setattr(OT, "TAPE_DIR", str(tmp_path))
    a = np.arange(5000, dtype=np.uint16)
    ids = np.array([[3, 1, 4]], np.int32)
    t = Tape("x", "record")
    assert t.oracle and not t.gpu
    t.check("a", None, lambda: a)
    t.check("ids", None, lambda: ids)
    t.done(kind="unit")
    r = Tape("x", "replay")
    assert r.gpu and not r.oracle
    r.check("a", lambda: a.copy(), None)
    r.check("ids", lambda: ids.copy(), None)
    r.done()
    r = Tape("x", "replay")
    with pytest.raises(AssertionError):
        b = a.copy()
        b[17] ^= 1
        r.check("a", lambda: b, None)                      # one flipped bit
    r = Tape("x", "replay")
    with pytest.raises(AssertionError):
        r.check("ids", lambda: ids, None)                  # out of step with the tape
    r = Tape("x", "replay")
    r.check("a", lambda: a, None)
    with pytest.raises(AssertionError):
        r.done()                                           # an entry the test never reached
    assert "values" in digest(ids) and "sha256" in digest(a)


def test_dry_oracle_keeps_the_bookkeeping_of_the_live_one():
    from oracle import qwen3_ref as QR
    cfg = QR.tiny_cfg()
    W = QR.random_weights(cfg, 0, 0.08)
    live = QR.Qwen3Ref(cfg, W, page_size=16, max_pages=32, max_batch=2)
    dry = QR.Qwen3Ref(cfg, None, page_size=16, max_pages=32, max_batch=2, dry=True)
    rng = np.random.default_rng(0)
    fields = lambda q: (list(q.kv_pages), q.kv_token_len, q.kv_last_page_len, q.next_position_id)
    rl, rd = [], []
    for n in (15, 30):
        ids = np.zeros((n, cfg.n_groups + 1), np.int32)
        ids[:, -1] = rng.integers(0, cfg.text_vocab, n)
        feats = np.zeros((n, cfg.talker.hidden), np.uint16)
        a, b = QR.RefRequest(), QR.RefRequest()
        lg, hid = live.prefill(a, ids, np.ones(n, np.uint8), feats)
        live.frame([a], lg, hid)
        lg, hid = dry.prefill(b, ids, None, None)
        assert dry.frame([b], lg, hid) == (None,) * 4
        rl.append(a), rd.append(b)
    for _ in range(4):                                     # crosses a page boundary (15 + 4 > 16)
        live.frame(rl)
        dry.frame(rd, *dry.decode(rd))
        assert [fields(q) for q in rl] == [fields(q) for q in rd] and live.free_pages == dry.free_pages


def test_hashed_weights_are_the_same_bits_in_numpy_and_torch():
    import torch
    from oracle import voxref as vr
    for shape in ((3000, 1024), (5, 700, 1024)):
        a = vr.random_bf16(np.random.default_rng(7), shape, 0.02)
        b = vr.random_bf16(np.random.default_rng(7), shape, 0.02, device="cpu")
        assert b.dtype == torch.bfloat16 and tuple(b.shape) == shape and np.array_equal(a, vr.from_torch(b))
        f = vr.bf2f(a).astype(np.float64)
        assert abs(f.std() - 0.02) < 2e-4 and abs(f.mean()) < 1e-4 and abs((a >> 15).mean() - 0.5) < 1e-3
    small = vr.random_bf16(np.random.default_rng(7), (64, 64), 0.02, device="cpu")
    assert np.array_equal(vr.from_torch(small), vr.random_bf16(np.random.default_rng(7), (64, 64), 0.02))
