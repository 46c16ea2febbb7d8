"""Development aid: where the time-to-first-audio goes (prefill / 9 decode frames / first codec chunk)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda")
loop = bench.Loop(1, 300, dev)
for it in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop.start_requests(); torch.cuda.synchronize()
    t1 = time.perf_counter()
    pcm = None
    while pcm is None:
        ids, pcm = loop.step()
    t2 = time.perf_counter()
    # codec alone
    torch.cuda.synchronize(); t3 = time.perf_counter()
    wav, _ = loop.codec.decode_chunk(loop.tok_ring, loop.cache, code_layout="BTQ"); torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"iter {it}: prefill {1e3*(t1-t0):6.2f} ms | 9 frames + codec + pcm {1e3*(t2-t1):6.2f} ms | codec chunk alone {1e3*(t4-t3):5.2f} ms | TTFA {1e3*(t2-t0):6.2f} ms")
    loop.codec.release_cache(loop.cache); loop.kvlen = [0]; loop.nframe = 0
