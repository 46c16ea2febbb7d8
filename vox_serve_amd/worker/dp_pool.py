"""Data-parallel worker pool: one process per GPU, requests sharded round-robin, no collective on the token path.

The reference's `--dp-size N` spawns N scheduler processes and routes `rank = counter % dp_size`
(/root/reference/vox_serve/launch.py:183-279, :471-474); every request lives on one GPU for its whole life.  Here the
processes are torch.distributed ranks (RCCL over xGMI on the GPUs, gloo on CPU in the tests) and the collectives are
all OFF the per-token path:
  * broadcast_weights  — load-time: rank 0 reads the checkpoint once, the others receive it over xGMI, in large
                         flat buckets (ring broadcast is per-link bound, ~153 GB/s: few big messages, not many small);
  * gather_results     — end of a batch job: per-request PCM back to rank 0 (the online server uses the result socket).
"""
from typing import Callable, Dict, List

import torch
import torch.distributed as dist


def route(counter: int, dp_size: int) -> int:
    """launch.py:471-474 — plain round-robin, no load feedback."""
    return counter % dp_size


def shard_requests(requests: List, rank: int, dp_size: int) -> List:
    return [r for i, r in enumerate(requests) if route(i, dp_size) == rank]


def broadcast_weights(weights: Dict[str, torch.Tensor], src: int = 0, bucket_bytes: int = 512 << 20, group=None, force: bool = False):
    """In-place broadcast of a state dict whose keys/shapes every rank already knows (tensors pre-allocated on the
    receiving ranks).  Tensors of one dtype are packed into flat buckets so that the ring moves few, large messages.
    force: go through the collective even in a one-rank group (bench.py --force-dist: the RCCL path on one GPU)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return weights
    by_dtype = {}
    for k in sorted(weights):
        by_dtype.setdefault(weights[k].dtype, []).append(k)
    rank = dist.get_rank(group)
    for dtype, keys in by_dtype.items():
        i = 0
        while i < len(keys):
            chunk, nbytes = [], 0
            while i < len(keys) and (not chunk or nbytes + weights[keys[i]].numel() * weights[keys[i]].element_size() <= bucket_bytes):
                chunk.append(keys[i])
                nbytes += weights[keys[i]].numel() * weights[keys[i]].element_size()
                i += 1
            dev = weights[chunk[0]].device
            flat = torch.empty(sum(weights[k].numel() for k in chunk), dtype=dtype, device=dev)
            if rank == src:
                torch.cat([weights[k].reshape(-1) for k in chunk], out=flat)
            dist.broadcast(flat, src=src, group=group)
            if rank != src:
                o = 0
                for k in chunk:
                    n = weights[k].numel()
                    weights[k].copy_(flat[o:o + n].view_as(weights[k]))
                    o += n
    return weights


def run_sharded(requests: List, serve_fn: Callable[[List], Dict[str, bytes]], group=None) -> Dict[str, bytes]:
    """Batch job over the pool: every rank serves its round-robin share with `serve_fn` (its own scheduler + worker),
    rank 0 returns {request_id: pcm bytes} for all requests.  DP-N output for a request == single-GPU output."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = serve_fn(shard_requests(requests, rank, world))
    if world == 1:
        return mine
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(mine, gathered, dst=0, group=group)
    if rank != 0:
        return {}
    out = {}
    for d in gathered:
        out.update(d)
    return out
