"""Where does the time to first audio go?  One probe request at a time through Scheduler + ModelWorker (as bench.py's
serving_ttfa), wall time of every scheduler step of the last probe, plus a cProfile of the probes (development aid)."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import PROMPT_TOKENS, INTERVAL
from vox_serve_amd.engine import Qwen3Cfg
from vox_serve_amd.model.qwen3_tts import Qwen3TTSModel
from vox_serve_amd.sampling import SamplingConfig
from vox_serve_amd.scheduler import QueueTransport, Scheduler, encode_request
from vox_serve_amd.synth import synth_qwen3_codec_weights, synth_qwen3_weights
from vox_serve_amd.worker import ModelWorker

dev = torch.device("cuda:0")
W, cW = synth_qwen3_weights(Qwen3Cfg(), dev, seed=0), synth_qwen3_codec_weights(seed=0)
m = Qwen3TTSModel("qwen3-tts", W, cW, device=str(dev), detokenize_interval=INTERVAL, max_batch_size=8, page_size=128,
                  max_num_pages=40, max_seq_len=2304, max_prefill_tokens=128)
m.default_sampling_config = SamplingConfig(greedy=True, max_tokens=PROMPT_TOKENS + INTERVAL + 6, repetition_penalty=1.05, repetition_window=-1)
t = QueueTransport()
w = ModelWorker(model=m, max_batch_size=8, max_num_pages=40, page_size=128, device=str(dev))
s = Scheduler(w, max_batch_size=8, transport=t, async_scheduling="--async" in sys.argv)
rng = np.random.default_rng(0)

def probe(i, trace=None):
    ids = [1, 2, 3] + rng.integers(0, 151000, 64).tolist() + [4, 5, 6, 7, 8]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t.requests.put(encode_request(f"p{i}", "", model_kwargs={"prompt_token_ids": ids, "language": "english"}))
    while True:
        a = time.perf_counter()
        s._step()
        b = time.perf_counter()
        if trace is not None:
            trace.append((b - a) * 1e3)
        hit = False
        while not t.results.empty():
            rid, kind, _ = t.results.get().split(b"|", 2)
            hit = hit or kind == b"AUDIO"
        if hit:
            break
    dt = (time.perf_counter() - t0) * 1e3
    s.run_until_idle(1000)
    while not t.results.empty():
        t.results.get()
    return dt

for i in range(6):
    probe(i)
tr = []
print("ttfa ms", [round(probe(10 + i), 2) for i in range(5)])
print("last probe:", round(probe(99, tr), 2), "ms; steps (ms):", [round(x, 2) for x in tr])
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    probe(200 + i)
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(45)
print(out.getvalue()[:9000])
