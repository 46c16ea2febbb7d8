"""Rank program of tests/test_gpu_multi.py::test_rccl_broadcast_weights_between_two_gpus (run under torch.distributed.run, one
rank per GPU, backend nccl = RCCL over xGMI): rank 0 holds seeded weights, the others zeros; after
worker/dp_pool.broadcast_weights every rank's checksum must equal rank 0's, the max over ranks of a timed barrier is taken the
way bench.py does, and rank 0 prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    from vox_serve_amd.worker.dp_pool import broadcast_weights
    g = torch.Generator(device="cpu").manual_seed(123)
    shapes = {"a.weight": (1024, 2048), "b.weight": (4096, 512), "c.bias": (4096,), "d.f32": (333, 7)}
    W = {}
    for k, s in shapes.items():
        dt = torch.float32 if k.endswith("f32") else torch.bfloat16
        t = torch.randn(*s, generator=g).to(dt)
        W[k] = t.to(dev) if rank == 0 else torch.zeros(*s, dtype=dt, device=dev)
    want = {k: int(torch.randn(*s, generator=torch.Generator().manual_seed(123)).numel()) for k, s in shapes.items()}      # (shapes only)
    t0 = time.perf_counter()
    broadcast_weights(W, src=0, bucket_bytes=1 << 20)           # small buckets: several messages per dtype
    torch.cuda.synchronize()
    dt_s = torch.tensor([time.perf_counter() - t0], device=dev)
    dist.all_reduce(dt_s, op=dist.ReduceOp.MAX)
    # checksum: exact integer sum of the bit patterns
    def cks(t):
        v = t.view(torch.int16) if t.dtype == torch.bfloat16 else t.view(torch.int32)
        return int(v.to(torch.int64).sum().item())
    mine = torch.tensor([cks(W[k]) for k in sorted(W)], dtype=torch.int64, device=dev)
    allc = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allc, mine)
    ok = all(torch.equal(allc[0], c) for c in allc) and int(mine.abs().sum()) != 0
    dist.barrier()
    if rank == 0:
        print(json.dumps({"ok": bool(ok), "world": world, "max_s": float(dt_s.item()), "n": want}), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
