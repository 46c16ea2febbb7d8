"""Offline analysis of tools/mfma_probe output: which arithmetic does v_mfma_f32_16x16x32_bf16 perform?

  python tools/mfma_cases.py /tmp/mfma_in.bin           # regenerate the (seeded) inputs locally
  python tools/mfma_model.py /tmp/mfma_in.bin gpurun_out/r2/mfma_out.bin

Everything is exact rational arithmetic (fractions.Fraction) + explicit fp32 rounding, so a candidate model either
reproduces the hardware bit for bit on every probe or it does not.
"""
import json
import sys
from fractions import Fraction

import numpy as np


def bf2f(u16):
    return (np.asarray(u16, np.uint32) << 16).view(np.float32)


def load(path_in, path_out):
    raw = np.fromfile(path_in, dtype=np.uint8)
    n = int(raw[:4].view(np.int32)[0])
    rec = raw[4:].reshape(n, 3072)
    A = bf2f(rec[:, :1024].copy().view(np.uint16).reshape(n, 16, 32))
    B = bf2f(rec[:, 1024:2048].copy().view(np.uint16).reshape(n, 32, 16))
    C = rec[:, 2048:].copy().view(np.float32).reshape(n, 16, 16)
    D = np.fromfile(path_out, dtype=np.float32).reshape(n, 16, 16)
    idx = json.load(open(path_in + ".json"))
    return A, B, C, D, idx


def F(x):
    return Fraction(float(x))


def round_f32(q: Fraction, mode="rne") -> float:
    """Round an exact rational to fp32 (normal + subnormal range, no overflow handling beyond inf)."""
    if q == 0:
        return 0.0
    s = -1 if q < 0 else 1
    a = abs(q)
    # exponent e with 2^e <= a < 2^(e+1)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1
    elif Fraction(2) ** (e + 1) <= a:
        e += 1
    e = max(e, -126)
    ulp = Fraction(2) ** (e - 23)
    m = a / ulp
    fl = m.numerator // m.denominator
    rem = m - fl
    if mode == "rne":
        if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (fl & 1)):
            fl += 1
    elif mode == "trunc":
        pass
    elif mode == "rna":
        if rem >= Fraction(1, 2):
            fl += 1
    else:
        raise ValueError(mode)
    v = float(fl * ulp)
    return float(np.float32(s * v))


def prods(a, b):
    return [F(x) * F(y) for x, y in zip(a, b)]


# ---- candidate models: (a[32], b[32], c) -> float ----
def m_exact(a, b, c, mode="rne"):
    return round_f32(F(c) + sum(prods(a, b)), mode)


def m_seq(a, b, c, mode="rne", order=None):
    acc = F(c)
    p = prods(a, b)
    for k in (order or range(32)):
        acc = F(round_f32(acc + p[k], mode))
    return float(acc)


def m_block(a, b, c, g=8, mode="rne", c_first=True, round_block=False, order=None):
    """blocks of g products summed exactly (optionally rounded), accumulated sequentially with one rounding per block"""
    p = prods(a, b)
    ks = list(order or range(32))
    acc = F(c) if c_first else Fraction(0)
    for j in range(0, 32, g):
        s = sum(p[k] for k in ks[j:j + g])
        if round_block:
            s = F(round_f32(s, mode))
        acc = F(round_f32(acc + s, mode))
    if not c_first:
        acc = F(round_f32(acc + F(c), mode))
    return float(acc)


def bits(x):
    return int(np.float32(x).view(np.uint32))


def check(name, fn, cases, verbose=3):
    bad = 0
    for (a, b, c, d) in cases:
        r = fn(a, b, c)
        if bits(r) != bits(d):
            bad += 1
            if bad <= verbose:
                print(f"   {name}: got {r!r} ({bits(r):08x}) hw {float(d)!r} ({bits(d):08x})")
    print(f"{name}: {len(cases) - bad}/{len(cases)} bit-exact")
    return bad


def dots_rows(A, B, C, D, ent):
    out = []
    for ci in range(ent["start"], ent["start"] + ent["n_cases"]):
        for r in range(16):
            if (ci - ent["start"]) * 16 + r >= ent["n_rows"]:
                break
            assert (D[ci, r] == D[ci, r, 0]).all() or np.isnan(D[ci, r]).all(), "columns differ with B = ones"
            out.append((A[ci, r], B[ci, :, 0], C[ci, r, 0], D[ci, r, 0]))
    return out


def dots_full(A, B, C, D, ent, max_cases=4, stride=1):
    out = []
    for ci in range(ent["start"], ent["start"] + min(ent["n_cases"], max_cases)):
        for r in range(0, 16, stride):
            for col in range(0, 16, stride):
                out.append((A[ci, r], B[ci, :, col], C[ci, r, col], D[ci, r, col]))
    return out


def main():
    A, B, C, D, idx = load(sys.argv[1], sys.argv[2])
    print({k: v["n_cases"] for k, v in idx.items()})
    # ---- P1: which (i, j, l) keep the 1? ----
    p1 = dots_rows(A, B, C, D, idx["p1_big_one_negbig"])
    keep = np.zeros((32, 32, 32), np.int8) - 1
    vals = set()
    for a, b, c, d in p1:
        i, j, l = int(np.argmax(a)), int(np.where(a == 1.0)[0][0]), int(np.argmin(a))
        keep[i, j, l] = 1 if d == 1.0 else 0
        vals.add(float(d))
    print("P1 result values:", sorted(vals)[:10])
    print("P1 fraction of triples where the 1 survives:", float((keep == 1).sum()) / (keep >= 0).sum())
    # survive pattern as a function of block membership
    for g in (2, 4, 8, 16, 32):
        same = []
        for i in range(32):
            for j in range(32):
                for l in range(32):
                    if keep[i, j, l] >= 0:
                        same.append((i // g == l // g, i // g == j // g, keep[i, j, l]))
        tab = {}
        for s_il, s_ij, k in same:
            tab.setdefault((s_il, s_ij), []).append(k)
        print(f" block {g}:", {k: (sum(v), len(v)) for k, v in tab.items()})
    # ---- P2 ----
    for nm in ("p2_c_big", "p2b_c_negbig"):
        p2 = dots_rows(A, B, C, D, idx[nm])
        vals = {}
        for a, b, c, d in p2:
            vals[float(d)] = vals.get(float(d), 0) + 1
        print(nm, "values:", vals)
    # ---- P3 width ----
    p3 = dots_rows(A, B, C, D, idx["p3_width"])
    cur = None
    for a, b, c, d in p3:
        nz = tuple(np.nonzero(a)[0])
        key = (nz, float(c) != 0)
        e = int(np.log2(np.abs(a).max() if c == 0 else abs(c)))
        if key != cur:
            cur = key
            print(" P3 positions", [(int(k), float(np.sign(a[k]))) for k in nz], "c" if c != 0 else "", end=": ")
            last = None
        if d != last:
            print(f"e={e}->{float(d)}", end=" ")
            last = d
        if e == 100:
            print()
    models = {
        "exact_rne": lambda a, b, c: m_exact(a, b, c),
        "exact_trunc": lambda a, b, c: m_exact(a, b, c, "trunc"),
        "seq_rne": lambda a, b, c: m_seq(a, b, c),
        "block4_rne": lambda a, b, c: m_block(a, b, c, 4),
        "block8_rne": lambda a, b, c: m_block(a, b, c, 8),
        "block16_rne": lambda a, b, c: m_block(a, b, c, 16),
        "block8_rne_clast": lambda a, b, c: m_block(a, b, c, 8, c_first=False),
        "block4_rne_clast": lambda a, b, c: m_block(a, b, c, 4, c_first=False),
    }
    for nm in ("p4_rounding",):
        cases = dots_rows(A, B, C, D, idx[nm])
        for mn, fn in models.items():
            check(f"{nm}/{mn}", fn, cases, verbose=0)
    for nm in ("p5_rand_c0", "p5_rand_c1", "p5_rand_c100", "p6_wide", "p7_sparse"):
        cases = dots_full(A, B, C, D, idx[nm], max_cases=2, stride=2)
        for mn, fn in models.items():
            check(f"{nm}/{mn}", fn, cases, verbose=0)


if __name__ == "__main__":
    main()
