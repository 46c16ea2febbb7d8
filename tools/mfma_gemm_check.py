"""GPU: libvoxhip's > 8-row linears (the MFMA kernels) against the oracle's restated summation orders.

For every (rows, N, K) it runs vox_linear / vox_linear_silu_mul through the C ABI and reports which of the oracle's
orders (canonical, full-K, 4-wave interleave, split-K) reproduces the output bit for bit.  Development aid for the
dispatch mirror in oracle/policy.py; the parity tests proper are tests/test_gpu_ops.py.
"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from oracle import voxref as vr  # noqa: E402
from vox_serve_amd import _native as N  # noqa: E402

NAMES = ["canon", "fullk", "mfma4", "splitk"]


def T(a, dev):
    return vr.to_torch(a).to(dev)


def Bits(t):
    return vr.from_torch(t)


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    rng = np.random.default_rng(0)
    shapes = [(4096, 2048), (2048, 2048), (2048, 6144), (1024, 2048), (4096, 1024), (1024, 3072), (3072, 2048), (2048, 1024), (256, 512), (128, 96)]
    out = []
    for B in (9, 16, 17, 32, 33, 64, 75, 128, 129, 200):
        for (Nn, K) in shapes:
            W = vr.f2bf((rng.standard_normal((Nn, K)) * 0.02).astype(np.float32))
            x = vr.f2bf(rng.standard_normal((B, K)).astype(np.float32))
            bias = vr.f2bf((rng.standard_normal(Nn) * 0.1).astype(np.float32))
            y = torch.empty(B, Nn, dtype=torch.bfloat16, device=dev)
            Wt, xt, bt = T(W, dev), T(x, dev), T(bias, dev)
            N.check(N.lib().vox_linear(N.ctx(), N.stream(), N.ptr(Wt), N.ptr(bt), N.ptr(xt), None, N.ptr(y), B, Nn, K, 0))
            torch.cuda.synchronize()
            got = Bits(y)
            match = {NAMES[o]: float((vr.linear(W, x, bias, order=o) == got).mean()) for o in range(4)}
            best = max(match, key=match.get)
            out.append({"op": "linear", "B": B, "N": Nn, "K": K, "best": best, "match": match})
            print(out[-1], flush=True)
        for (Nn, K) in [(6144, 2048), (3072, 1024), (4096, 4096)]:
            Wg = vr.f2bf((rng.standard_normal((Nn, K)) * 0.05).astype(np.float32))
            Wu = vr.f2bf((rng.standard_normal((Nn, K)) * 0.05).astype(np.float32))
            x = vr.f2bf(rng.standard_normal((B, K)).astype(np.float32))
            h = torch.empty(B, Nn, dtype=torch.bfloat16, device=dev)
            Wgt, Wut, xt = T(Wg, dev), T(Wu, dev), T(x, dev)
            N.check(N.lib().vox_linear_silu_mul(N.ctx(), N.stream(), N.ptr(Wgt), N.ptr(Wut), N.ptr(xt), N.ptr(h), B, Nn, K))
            torch.cuda.synchronize()
            got = Bits(h)
            match = {NAMES[o]: float((vr.linear_silu_mul(Wg, Wu, x, order=o) == got).mean()) for o in range(4)}
            out.append({"op": "silu_mul", "B": B, "N": Nn, "K": K, "best": max(match, key=match.get), "match": match})
            print(out[-1], flush=True)
    json.dump(out, open("gpurun_out/r2/mfma_gemm_check.json", "w"))


if __name__ == "__main__":
    main()
