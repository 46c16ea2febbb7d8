"""Development aid: N prefills of a 75-token prompt (bench.Loop.start_requests), for rocprofv3 --kernel-trace --stats."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda")
loop = bench.Loop(1, 300, dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for it in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loop.start_requests(); torch.cuda.synchronize()
    print(f"prefill {it}: {1e3 * (time.perf_counter() - t0):.2f} ms")
    loop.codec.release_cache(loop.cache); loop.kvlen = [0]; loop.nframe = 0
