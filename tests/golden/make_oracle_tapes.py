"""Record the oracle tapes of the heavy GPU parity cases (tests/oracle_tape.py).

    python tests/golden/make_oracle_tapes.py            # every case whose tape is missing or stale
    python tests/golden/make_oracle_tapes.py --all      # re-record everything
    python tests/golden/make_oracle_tapes.py CASE ...   # just these

Runs the CPU oracle (oracle/) only — no GPU, no reference import: the cases are the `@taped` functions of the GPU test
modules, called with a recording Tape (their engine side is skipped).  Minutes of host time for the full-size cases; the GPU
box then replays the tapes in seconds.  Re-run after any change to the oracle's sources (tests/test_oracle_tapes.py fails
on a stale tape) — in particular after a kernel's summation order, and with it oracle/policy.py or voxref.c, changed.
"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

MODULES = ["tests.test_gpu_qwen3", "tests.test_gpu_lm", "tests.test_gpu_csm", "tests.test_gpu_ops"]


def all_cases():
    cases = {}
    for m in MODULES:
        mod = importlib.import_module(m)
        for name, fn in getattr(mod, "TAPED", {}).items():
            assert name not in cases, name
            cases[name] = fn
    return cases


def main(argv):
    import tests.conftest  # noqa: F401  (sets the OpenMP thread budget)
    from tests.oracle_tape import Tape, oracle_sources_sha
    cases = all_cases()
    want = [a for a in argv if not a.startswith("-")] or list(cases)
    sha = oracle_sources_sha()
    for name in want:
        path = Tape.path(name)
        if "--all" not in argv and not [a for a in argv if not a.startswith("-")] and os.path.exists(path):
            if json.load(open(path))["meta"].get("oracle_sources_sha") == sha:
                print(f"{name}: up to date")
                continue
        t0 = time.time()
        cases[name](Tape(name, "record"), None)
        print(f"{name}: recorded in {time.time() - t0:.1f} s -> {os.path.relpath(path, ROOT)} ({os.path.getsize(path)} B)", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
