"""GPU: where does a scheduler step go?  cProfile of the serving path (encode_request -> Scheduler -> ModelWorker -> engine
-> codec -> result queue) on the full-size synthetic Qwen3-TTS, batch-1 streaming and under load."""
import cProfile
import io
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])


def main():
    load = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from vox_serve_amd.engine import Qwen3Cfg
    from vox_serve_amd.model.qwen3_tts import Qwen3TTSModel
    from vox_serve_amd.sampling import SamplingConfig
    from vox_serve_amd.scheduler import QueueTransport, Scheduler, encode_request
    from vox_serve_amd.synth import synth_qwen3_codec_weights, synth_qwen3_weights
    from vox_serve_amd.worker import ModelWorker
    mb = max(8, load + 1)
    m = Qwen3TTSModel("qwen3-tts", synth_qwen3_weights(Qwen3Cfg(), dev, seed=0), synth_qwen3_codec_weights(seed=0), device=str(dev),
                      max_batch_size=mb, page_size=128, max_num_pages=4 * mb + 8, max_seq_len=2304, max_prefill_tokens=128)
    m.default_sampling_config = SamplingConfig(greedy=True, max_tokens=400, repetition_penalty=1.05, repetition_window=-1)
    t = QueueTransport()
    w = ModelWorker(model=m, max_batch_size=mb, max_num_pages=4 * mb + 8, page_size=128, device=str(dev))
    s = Scheduler(w, max_batch_size=mb, transport=t)
    rng = np.random.default_rng(0)
    for i in range(load + 1):
        ids = [1, 2, 3] + rng.integers(0, 151000, 64).tolist() + [4, 5, 6, 7, 8]
        t.requests.put(encode_request(f"r{i}", "", model_kwargs={"prompt_token_ids": ids, "language": "english"}))
    use_async = len(sys.argv) > 3 and sys.argv[3] == "async"
    if use_async:
        import asyncio
        s.async_scheduling = True
        asyncio.run(s._run_async_loop(max_steps=load + 30))
    else:
        for _ in range(load + 30):       # prefill everything, capture graphs
            s._step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    if use_async:
        asyncio.run(s._run_async_loop(max_steps=n_steps))
    else:
        for _ in range(n_steps):
            s._step()
    torch.cuda.synchronize()
    pr.disable()
    dt = time.perf_counter() - t0
    print(f"load {load}: {n_steps} steps in {dt * 1e3:.1f} ms = {dt / n_steps * 1e3:.2f} ms/step, active {len(s.active_requests)}")
    st = io.StringIO()
    pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(45)
    print(st.getvalue())


if __name__ == "__main__":
    main()
