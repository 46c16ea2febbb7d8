"""Generate probe cases for tools/mfma_probe (v_mfma_f32_16x16x32_bf16 accumulation arithmetic).

  python tools/mfma_cases.py gpurun_out/mfma_in.bin

A "dot case" is (a[32], b[32], c) -> d.  Structured probes put one dot case per A row with B = ones (16 per MFMA);
random probes fill A, B, C (256 dependent dot cases per MFMA).  The layout of the file is the probe's `Case` struct.
An index file (<out>.json) records which MFMA cases belong to which experiment.
"""
import json
import sys

import numpy as np


def f2bf(x):
    """fp32 -> bf16 bits, round-to-nearest-even (inputs here are exactly representable except the random ones)."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    return r


class Builder:
    def __init__(self):
        self.A, self.B, self.C = [], [], []
        self.index = {}

    def add_rows(self, name, rows_a, rows_c):
        """rows_a [n,32] float (bf16-exact), rows_c [n]; B = ones.  Packs 16 rows per MFMA case."""
        rows_a = np.asarray(rows_a, np.float32)
        rows_c = np.asarray(rows_c, np.float32)
        n = rows_a.shape[0]
        pad = (-n) % 16
        if pad:
            rows_a = np.concatenate([rows_a, np.zeros((pad, 32), np.float32)])
            rows_c = np.concatenate([rows_c, np.zeros(pad, np.float32)])
        start = len(self.A)
        for i in range(0, n + pad, 16):
            self.A.append(f2bf(rows_a[i:i + 16]))
            self.B.append(f2bf(np.ones((32, 16), np.float32)))
            self.C.append(np.repeat(rows_c[i:i + 16, None], 16, axis=1).astype(np.float32))
        self.index[name] = {"kind": "rows", "start": start, "n_rows": n, "n_cases": (n + pad) // 16}

    def add_full(self, name, A, B, C):
        start = len(self.A)
        for a, b, c in zip(A, B, C):
            self.A.append(f2bf(a))
            self.B.append(f2bf(b))
            self.C.append(np.asarray(c, np.float32))
        self.index[name] = {"kind": "full", "start": start, "n_cases": len(A)}

    def write(self, path):
        n = len(self.A)
        with open(path, "wb") as f:
            f.write(np.int32(n).tobytes())
            for a, b, c in zip(self.A, self.B, self.C):
                f.write(a.astype(np.uint16).tobytes())
                f.write(b.astype(np.uint16).tobytes())
                f.write(c.astype(np.float32).tobytes())
        json.dump(self.index, open(path + ".json", "w"))
        print(f"{n} MFMA cases -> {path}")


def main(path):
    rng = np.random.default_rng(0)
    b = Builder()
    # P1: big at i, one at j, -big at l (c = 0): does the 1 survive?  all ordered triples
    rows, meta = [], []
    for i in range(32):
        for j in range(32):
            for l in range(32):
                if len({i, j, l}) == 3:
                    a = np.zeros(32, np.float32)
                    a[i], a[j], a[l] = 2.0 ** 24, 1.0, -(2.0 ** 24)
                    rows.append(a)
    b.add_rows("p1_big_one_negbig", np.array(rows), np.zeros(len(rows)))
    # P2: c = 2^24, one at j, -2^24 at l
    rows = []
    for j in range(32):
        for l in range(32):
            if j != l:
                a = np.zeros(32, np.float32)
                a[j], a[l] = 1.0, -(2.0 ** 24)
                rows.append(a)
    b.add_rows("p2_c_big", np.array(rows), np.full(len(rows), 2.0 ** 24))
    # P2b: c = -2^24, 2^24 at i, one at j
    rows = []
    for i in range(32):
        for j in range(32):
            if i != j:
                a = np.zeros(32, np.float32)
                a[i], a[j] = 2.0 ** 24, 1.0
                rows.append(a)
    b.add_rows("p2b_c_negbig", np.array(rows), np.full(len(rows), -(2.0 ** 24)))
    # P3: width — 2^e at i, 1 at j, -2^e at l for e = 1..100, a few position triples; also vs c
    triples = [(0, 1, 2), (0, 2, 1), (2, 0, 1), (0, 1, 7), (0, 8, 16), (0, 1, 31), (7, 8, 9), (0, 16, 1), (15, 16, 17), (3, 4, 5), (0, 4, 1)]
    rows, cs = [], []
    for (i, j, l) in triples:
        for e in range(1, 101):
            a = np.zeros(32, np.float32)
            a[i], a[j], a[l] = 2.0 ** e, 1.0, -(2.0 ** e)
            rows.append(a)
            cs.append(0.0)
    for j in (0, 5, 31):
        for l in (1, 8, 30):
            for e in range(1, 101):          # c = 2^e, 1 at j, -2^e at l
                a = np.zeros(32, np.float32)
                a[j], a[l] = 1.0, -(2.0 ** e)
                rows.append(a)
                cs.append(2.0 ** e)
    b.add_rows("p3_width", np.array(rows), np.array(cs))
    # P4: rounding of the final add: c = 2^23 + m, products = small dyadic fractions at assorted positions
    rows, cs = [], []
    for m in (0, 1):
        for pat in range(256):
            a = np.zeros(32, np.float32)
            bits = [(pat >> t) & 1 for t in range(8)]
            pos = [0, 1, 7, 8, 15, 16, 24, 31]
            vals = [0.5, 0.25, 0.125, 0.25, 0.5, 0.125, 0.0625, 0.0625]
            for t in range(8):
                if bits[t]:
                    a[pos[t]] = vals[t]
            for sgn in (1.0, -1.0):
                rows.append(a * sgn)
                cs.append(sgn * (2.0 ** 23 + m))
    b.add_rows("p4_rounding", np.array(rows), np.array(cs))
    # P5: random, several C scales
    for name, cscale, n in (("p5_rand_c0", 0.0, 64), ("p5_rand_c1", 1.0, 128), ("p5_rand_c100", 100.0, 64), ("p5_rand_csmall", 1e-3, 64)):
        A = rng.standard_normal((n, 16, 32)).astype(np.float32)
        B = rng.standard_normal((n, 32, 16)).astype(np.float32)
        C = (rng.standard_normal((n, 16, 16)) * cscale).astype(np.float32)
        b.add_full(name, A, B, C)
    # P6: wide dynamic range
    n = 128
    A = (rng.choice([-1.0, 1.0], (n, 16, 32)) * 2.0 ** rng.integers(-12, 13, (n, 16, 32)) * (1 + rng.integers(0, 128, (n, 16, 32)) / 128.0)).astype(np.float32)
    B = (rng.choice([-1.0, 1.0], (n, 32, 16)) * 2.0 ** rng.integers(-12, 13, (n, 32, 16)) * (1 + rng.integers(0, 128, (n, 32, 16)) / 128.0)).astype(np.float32)
    C = (rng.standard_normal((n, 16, 16)) * 2.0 ** rng.integers(-10, 20, (n, 16, 16))).astype(np.float32)
    b.add_full("p6_wide", A, B, C)
    # P7: sparse random: only 2..4 nonzero k per row (isolates pairwise behaviour), B random
    n = 128
    A = np.zeros((n, 16, 32), np.float32)
    for c in range(n):
        for r in range(16):
            k = rng.choice(32, rng.integers(2, 5), replace=False)
            A[c, r, k] = rng.standard_normal(len(k)) * 2.0 ** rng.integers(-8, 9, len(k))
    B = rng.standard_normal((n, 32, 16)).astype(np.float32)
    C = np.zeros((n, 16, 16), np.float32)
    b.add_full("p7_sparse", A, B, C)
    # P8: subnormal / tiny products
    n = 16
    A = (rng.standard_normal((n, 16, 32)) * 2.0 ** -70).astype(np.float32)
    B = (rng.standard_normal((n, 32, 16)) * 2.0 ** -70).astype(np.float32)
    C = (rng.standard_normal((n, 16, 16)) * 2.0 ** -140).astype(np.float32)
    b.add_full("p8_tiny", A, B, C)
    b.write(path)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/mfma_in.bin")
