#!/usr/bin/env python
"""bench.py — the streaming TTS hot loop of vox-serve on MI355X: Qwen3-TTS-1.7B (bf16, random-init weights of the
named architecture, synthetic fixed-length prompts), speech-token decode + token->waveform codec.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment) re-executes itself under
torch.distributed.run with N ranks on 127.0.0.1; the JSON line reports the ranks that actually took part (`ranks_seen`).
Without --batch the headline is batch 1 (BASELINE configs[1]) and the same line carries driver-timed sub-results for
batch 8 and batch 32 (`batch8`, `batch32`: value, ms_per_step, roofline) — the metric is quoted at batch 1 / 8 / 32.

A "step" is one audio frame (1920 samples at 24 kHz) for the whole batch of B concurrent requests: one hipGraph
replay = talker decode step + codebook-0 sampling + the 15-step depth loop (+ feedback of the next inputs), the
host-side plan upload and token read-back, and every 10th step one codec chunk (10 frames -> 19200 samples per
request) with PCM16 packing and D2H; the chunk runs on its own HIP stream concurrently with the next LM frames (the
disaggregation scheduler's two pipelines on one GPU) and all audio is on the host before the clock stops.  value = audio samples/s over all ranks (weak scaling: B per GPU fixed).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md)
PROMPT_TOKENS = 75         # 64 text tokens + 11 layout tokens
INTERVAL = 10              # detokenize_interval default (qwen3_tts.py:965)
KV_BYTES_PER_TOKEN = 28 * 2 * 8 * 128 * 2


_T0 = [time.perf_counter()]


def _phase(label):
    """wall time of each bench phase on stderr (the JSON line on stdout stays the only stdout output)"""
    now = time.perf_counter()
    print(f"[bench] {label}: {now - _T0[0]:.1f} s", file=sys.stderr, flush=True)
    _T0[0] = now


def algorithmic_bytes_per_frame(B, kv_mean):
    talker = 28 * 50.33e6 * 2 + 3072 * 2048 * 2 + (2048 * 2048 * 2 + 4096) * 2     # layers + codec_head + text_projection
    depth = (5 * 15.73e6 + 2048 * 1024 + 1024) * 2 + 15 * 2048 * 1024 * 2          # read once per frame
    return talker + depth + B * kv_mean * KV_BYTES_PER_TOKEN + B * KV_BYTES_PER_TOKEN


class Loop:
    """B concurrent requests in lock-step on one GPU: the worker's decode + detokenize hot loop."""

    def __init__(self, B, max_frames, dev, W=None, codec_W=None):
        from vox_serve_amd.engine import Qwen3Cfg, Qwen3Engine
        from vox_serve_amd.synth import synth_qwen3_codec_weights, synth_qwen3_weights
        from vox_serve_amd.tokenizer.qwen3_codec import Qwen3TTSDecoder
        self.B, self.dev, self.cfg = B, dev, Qwen3Cfg()
        self.ps = 128
        self.pages_per_req = (PROMPT_TOKENS + max_frames + self.ps) // self.ps + 1
        W = self.W = W if W is not None else synth_qwen3_weights(self.cfg, dev, seed=0)
        self.eng = Qwen3Engine(self.cfg, W, max_batch=B, page_size=self.ps, max_pages=B * self.pages_per_req + 1,
                               max_seq_len=2304, max_prefill_rows=128)
        self.eng.keep_hidden = False
        self.codec = Qwen3TTSDecoder(codec_W if codec_W is not None else synth_qwen3_codec_weights(seed=0), device=dev, max_batch=B, max_slots=B,
                                     detokenize_interval=INTERVAL)
        self.sc = self.eng.sampling_cfg(greedy=True)
        self.tok_ring = torch.zeros(B, INTERVAL, self.cfg.n_groups + 1, dtype=torch.int32, device=dev)
        self.pages = [[b * self.pages_per_req + j for j in range(self.pages_per_req)] for b in range(B)]
        self._pages_np = np.asarray(self.pages, dtype=np.int64)
        self.kvlen = [0] * B
        self.nframe = 0
        self.samples = 0
        self.rng = np.random.default_rng(1)
        self.frame_ev = []
        self.codec_stream = torch.cuda.Stream(device=dev)
        self.pcm_host = torch.zeros(B, INTERVAL * 1920, dtype=torch.int16).pin_memory()
        self.pending = None
        self.interval = INTERVAL            # frames per codec chunk (the TTFA-focused measurement lowers it to 2)
        # pipeline = True (default): the reference's async-scheduling contract (scheduler/base.py:166-221) — frame N+1 is enqueued before the
        # host has read frame N's tokens (a stream-ordered snapshot in pinned memory, read one step late), so the GPU never idles between
        # frames; False: lock-step (the host reads every frame's tokens before it enqueues the next)
        self.pipeline = True
        self._ids_pin = [torch.zeros(B + 1, self.cfg.n_groups + 1, dtype=torch.int32).pin_memory() for _ in range(2)]      # row 0: the frame's status row
        self._ids_ev = [torch.cuda.Event(), torch.cuda.Event()]
        self._ids_prev = None

    def start_requests(self):
        """Prefill every request (one per step, like the scheduler) -> first frame."""
        e, c = self.eng, self.cfg
        self.cache = self.codec.init_cache(self.B)
        first_inputs = []
        for b in range(self.B):
            n = PROMPT_TOKENS
            ids = np.zeros((n, c.n_groups + 1), np.int32)
            ids[:, -1] = self.rng.integers(0, 151000, n)
            ids[:, 0] = self.rng.integers(0, 2048, n)
            e.row_ids[:n] = torch.from_numpy(ids).to(self.dev)
            e.row_masks[:n] = 0
            e.row_masks[n - 1:n] = 1
            e.row_feats[:n].zero_()
            pg = self.pages[b]
            e.upload_plan(pos=np.arange(n), kvlen=np.arange(1, n + 1), page=[pg[t // self.ps] for t in range(n)],
                          slot=[t % self.ps for t in range(n)], q_req=np.zeros(n), last_rows=[n - 1],
                          indptr=[0, (n + self.ps - 1) // self.ps], indices=pg[: (n + self.ps - 1) // self.ps])
            e.prefill(n, 1, n, self.sc, feedback=True)
            first_inputs.append((e.input_ids[0].clone(), e.input_features[0].clone(), e.out_ids[0].clone()))
            self.kvlen[b] = n
        for b, (ii, ff, oo) in enumerate(first_inputs):
            e.input_ids[b], e.input_features[b] = ii, ff
            self.tok_ring[b, 0] = oo
        e.input_masks[: self.B] = 1
        self.nframe = 1
        self.pos = [PROMPT_TOKENS + 1] * self.B              # quirk Q1: first decode position is n+1 (worker/base.py:299)

    def step(self, timed_events=None, wait_pcm=False):
        """One frame for the whole batch."""
        e, B, ps = self.eng, self.B, self.ps
        kv = np.asarray(self.kvlen, dtype=np.int64) + 1            # the paged-KV plan of this frame, vectorised over the batch
        self.kvlen = kv.tolist()
        npg = (kv + ps - 1) // ps
        indptr = np.concatenate([[0], np.cumsum(npg)])
        pages = self._pages_np
        indices = pages[np.arange(pages.shape[1])[None, :] < npg[:, None]]
        page, slot = pages[np.arange(B), npg - 1], (kv - 1) % ps
        e.upload_plan(pos=self.pos, kvlen=kv, page=page, slot=slot, indptr=indptr, indices=indices)
        if timed_events is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(e.stream)
        e.frame(B, max(self.kvlen), self.sc, feedback=True, use_graph=True)
        if timed_events is not None:
            ev1.record(e.stream)
            timed_events.append((ev0, ev1))
        col = self.nframe % self.interval
        if self.pipeline:
            k = self.nframe & 1

            def snapshot(k=k, col=col):                        # stream-ordered right behind the frame: status row + ids, and the codec's token ring
                self.tok_ring[:, col] = e.out_ids[:B]
                self._ids_pin[k].copy_(e.snapshot_src(B), non_blocking=True)
                self._ids_ev[k].record()
            snapshot()
            ids = None
            if self._ids_prev is not None:                     # the PREVIOUS frame's tokens: the scheduler sees them one step late
                kp, redo_prev = self._ids_prev
                self._ids_ev[kp].synchronize()
                if int(self._ids_pin[kp][0, 0]) != 0:          # a hand-off timed out in the previous frame: replay it and this one (bit-identical)
                    e.recover(back=2, code=int(self._ids_pin[kp][0, 0]), on_first_done=redo_prev)
                    snapshot()
                    self._ids_ev[kp].synchronize()
                ids = self._ids_pin[kp][1:].clone()
            self._ids_prev = (k, snapshot)
        else:
            # the scheduler needs the tokens (EOS / max_tokens checks): ONE blocking D2H that also brings the status row — a hand-off
            # timeout of the persistent kernels is recovered from inside read_ids, before the codec's token ring sees the frame
            ids = e.read_ids(B)
            self.tok_ring[:, col] = e.out_ids[:B]
        self.pos = [p + 1 for p in self.pos]
        self.nframe += 1
        pcm = None
        if self.nframe % self.interval == 0:
            # the codec chunk runs on its own HIP stream, concurrently with the following LM frames (the
            # disaggregation scheduler's two pipelines on one GPU); its PCM is collected one chunk later, or at once
            # when the caller waits for it (TTFA)
            done = self.collect_pcm()
            snap = self.tok_ring[:, : self.interval].clone()
            self.codec_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.codec_stream):
                wav, _ = self.codec.decode_chunk(snap, self.cache, code_layout="BTQ")
                self.pcm_host[:, : self.interval * 1920].copy_((wav[:, 0] * 32767).to(torch.int16), non_blocking=True)      # worker/base.py:658-672
                ev = torch.cuda.Event()
                ev.record(self.codec_stream)
            snap.record_stream(self.codec_stream)
            self.pending = ev
            pcm = self.collect_pcm() if wait_pcm else done
        return ids, pcm

    def collect_pcm(self):
        """PCM16 of the chunk in flight (blocks until its D2H copy has landed), or None."""
        if self.pending is None:
            return None
        self.pending.synchronize()
        self.pending = None
        pcm = self.pcm_host[:, : self.interval * 1920].numpy().copy()
        self.samples += pcm.size
        return pcm


def _host_cpu_share() -> int:
    """CPUs this process may really use (affinity mask and cgroup quota): the GPU boxes show 256 logical CPUs to a 16-CPU share."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(W, budget_s=25.0):
    """The CPU oracle (oracle/voxref.c, OpenMP; oracle/qwen3_codec_ref.py, torch CPU) on the same workload:
    talker+depth decode frames at B=1 mid-stream (kv ~ 200, as the GPU line), plus one codec chunk.  Reported, never the product path."""
    from oracle import qwen3_codec_ref as CR
    from oracle import qwen3_ref as QR
    from oracle import voxref as vr
    cores = os.cpu_count() or 1
    try:      # one OpenMP thread per CPU of the container's share (the oracle links the system libgomp; torch carries its own copy)
        import ctypes
        share = _host_cpu_share()
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(share))
        cores = share
    except Exception:
        pass
    ref_cfg = QR.Qwen3Cfg(max_pos=256)
    src = {k: v.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16) for k, v in W.items()}
    m = QR.Qwen3Ref(ref_cfg, src, page_size=128, max_pages=2, max_batch=1)
    req = QR.RefRequest()
    rng = np.random.default_rng(1)
    n = 2                                         # short prefill just to have a live request (not timed; > 2 rows would take the
                                                  # oracle's restated matrix-core datapath: a minute per prefill)
    ids = np.zeros((n, 17), np.int32)
    ids[:, -1] = rng.integers(0, 151000, n)
    _phase("cpu baseline: oracle set-up")
    lg, hid = m.prefill(req, ids, np.ones(n, np.uint8), np.zeros((n, 2048), np.uint16))
    m.frame([req], lg, hid)
    # the GPU line is measured mid-stream at a mean KV length of 200: give the CPU request the same history (random K/V bits
    # in its pages — the timed frames then attend over ~200 tokens like the GPU's)
    kv0 = 197
    req.kv_pages.append(m.free_pages.pop(0))
    for layer in m.kv:
        layer[req.kv_pages, :, :, :, :] = vr.f2bf((0.5 * rng.standard_normal((2,) + layer.shape[1:])).astype(np.float32))
    req.kv_token_len, req.kv_last_page_len, req.next_position_id = kv0, kv0 - 128, kv0 + 1
    _phase("cpu baseline: prefill + first frame (untimed)")
    t0 = time.perf_counter()
    nf = 0
    while nf < 1 or (time.perf_counter() - t0 < budget_s * 0.4 and nf < 6):
        m.frame([req])
        nf += 1
    t_lm = (time.perf_counter() - t0) / nf
    _phase("cpu baseline: LM frames")
    tthreads = min(cores, 16)                     # torch-CPU conv stops scaling (and collapses) far below 256 threads
    torch.set_num_threads(tthreads)
    ccfg = CR.CodecCfg()
    cm = CR.Qwen3CodecRef(ccfg, CR.random_codec_weights(ccfg, seed=0))
    st = cm.init_state(1)
    codes = torch.randint(0, 2048, (1, 16, INTERVAL))
    t1 = time.perf_counter()
    cm.forward_chunk(codes, st)
    t_codec = time.perf_counter() - t1
    per_frame = t_lm + t_codec / INTERVAL
    return {"value": 1920.0 / per_frame, "unit": "audio samples/s", "cores": cores, "kind": "port",
            "sample": f"{nf} LM frames (talker 28L + 15x depth 5L, B=1, kv~{kv0 + 1 + nf // 2}) {t_lm:.2f} s/frame on {cores} OpenMP threads "
                      f"+ 1 codec chunk (10 frames) {t_codec:.2f} s on {tthreads} torch threads; "
                      f"oracle = C fixed-order fp32 + torch-CPU codec"}


def pmc_traffic(B):
    """HBM bytes per LM graph launch as RECORDED by the committed rocprofv3 PMC passes of this same command
    (profiles/round*_pmc_traffic.json: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950; tools/pmc_summary.py).  Hardware counters cannot be read from inside the
    timed run, so the line labels this value `traffic_source: "recorded <file>"`; null when no pass is committed."""
    pdir = os.path.join(ROOT, "profiles")
    for name in ("round6_pmc_traffic.json", "round5_pmc_traffic.json", "round4_pmc_traffic.json", "round3_pmc_traffic.json", "round2_pmc_traffic.json", "round1_pmc_traffic.json"):
        try:
            rec = json.load(open(os.path.join(pdir, name))).get(f"batch_{B}")
            if rec:
                return float(rec["fetch_corrected_bytes_per_launch"] + rec["write_raw_bytes_per_launch"]), f"recorded profiles/{name}"
        except (OSError, ValueError, KeyError):
            continue
    return None, None


class _StubEvent:
    def __init__(self, t):
        self.t = t

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class StubLoop:
    """--dry-run: the launch / barrier / max-over-ranks / JSON path of this file with no GPU behind it (gloo on CPU): a step
    sleeps 1 ms instead of replaying a frame.  Nothing it prints is a measurement."""

    class _Closable:
        def close(self):
            pass

        def release_cache(self, _):
            pass

    def __init__(self, B, max_frames, dev, W=None, codec_W=None):
        self.B, self.kvlen, self.samples, self.interval = B, [0] * B, 0, INTERVAL
        self.eng = self.codec = self._Closable()
        self.cache = None

    def start_requests(self):
        self.kvlen = [PROMPT_TOKENS] * self.B

    def step(self, timed_events=None, wait_pcm=False):
        t0 = time.perf_counter()
        time.sleep(0.001)
        self.kvlen = [k + 1 for k in self.kvlen]
        if timed_events is not None:
            timed_events.append((_StubEvent(t0), _StubEvent(time.perf_counter())))
        return None, (np.zeros(1, np.int16) if wait_pcm else None)

    def collect_pcm(self):
        return None


def run_batch(B, args, dev, world, shared, ttfa_requests=0, lockstep=False):
    """The timed region for one batch size: W warm-up steps, a barrier, exactly K steps, barrier, max over ranks."""
    import torch.distributed as dist
    use_dist = world > 1 or args.force_dist
    dry = getattr(args, "dry_run", False)
    sync = (lambda: None) if dry else torch.cuda.synchronize
    # requests are measured mid-stream: the KV length over the timed steps averages --kv-mean (200)
    pre = max(0, int(args.kv_mean - PROMPT_TOKENS - args.warmup - args.steps / 2))
    loop = (StubLoop if dry else Loop)(B, pre + args.steps + args.warmup + 64, dev, shared["W"], shared["codec_W"])
    loop.pipeline = not lockstep
    ttfa, ttfa2 = [], []
    if ttfa_requests > 0 and B == 1:
        # engine-level TTFA (request start -> first PCM chunk on the host), lock-step loop, outside the timed steps
        for interval, acc in ((INTERVAL, ttfa), (2, ttfa2)):
            loop.interval = interval
            for _ in range(ttfa_requests + 1):
                loop.kvlen, loop.samples = [0] * B, 0
                sync()
                t0 = time.perf_counter()
                loop.start_requests()
                pcm = None
                while pcm is None:
                    _, pcm = loop.step(wait_pcm=True)
                acc.append((time.perf_counter() - t0) * 1e3)
                loop.codec.release_cache(loop.cache)
            del acc[0]                      # (first request of a setting: graph capture for the new kv bucket / chunk shape)
        loop.interval = INTERVAL
    loop.kvlen, loop.samples = [0] * B, 0
    loop.start_requests()
    for _ in range(pre + args.warmup):
        loop.step()
    sync()
    if use_dist:
        dist.barrier()
    loop.samples = 0
    kv_start = loop.kvlen[0]
    events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loop.step(events)
    loop.collect_pcm()                     # the last chunk's audio must be on the host inside the timed region
    sync()
    dt = time.perf_counter() - t0
    if use_dist:
        dist.barrier()
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    frame_gpu_s = float(np.mean([a.elapsed_time(b) for a, b in events])) * 1e-3
    kv_mean = kv_start + args.steps / 2
    alg = algorithmic_bytes_per_frame(B, kv_mean)
    samples_total = world * B * 1920 * args.steps
    traffic, src = pmc_traffic(B)
    res = {
        "value": samples_total / dt, "ms_per_step": dt / args.steps * 1e3, "batch_per_gpu": B, "loop": "lock-step" if lockstep else "pipelined one frame deep",
        "realtime_factor_per_request": samples_total / dt / 24000.0 / (world * B), "kv_mean": kv_mean,
        "roofline": {"bound": "hbm", "achieved": alg / frame_gpu_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg / frame_gpu_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
                     "launch": "one hipGraph replay = one LM frame (talker + 15 depth steps + sampling)",
                     "algorithmic_bytes_per_launch": alg, "avg_launch_ms": frame_gpu_s * 1e3},
    }
    if not dry and hasattr(loop.eng, "depth_persist_status"):
        en, err = loop.eng.depth_persist_status()
        fails = list(getattr(loop.eng, "persist_failures", []))
        res["depth_persist"] = {"enabled_for_one_request_frames": en, "handoff_timeouts": len(fails) if fails else err,
                                "checked": "every frame (status row read with the token snapshot; a timeout replays the frame on the launch chain)"}
        if err or fails:
            # recovered frames are correct, but the timed region then mixes persistent and launch-chain frames: not the number to report
            raise SystemExit(f"persistent kernels: a hand-off timed out ({fails or hex(err)}): the measurement is invalid")
    if ttfa:
        res["ttfa_ms_p50_engine"] = float(np.median(ttfa))
        res["ttfa_ms_p50_engine_detokenize_interval_2"] = float(np.median(ttfa2))
    loop.eng.close()
    loop.codec.close()
    return res


def kv_sweep(dev, shared, batches=(1, 32), kvs=(100, 200, 325, 1000, 2000), frames=20):
    """LM frame (one hipGraph replay, HIP events on the engine's stream) at fixed visible lengths, the KV state injected as random
    pages: kv 100 / 200 sit in the one-launch decode attention (<= 256 visible tokens), 325 is the END of SURVEY 8d's 250-frame
    request and 1000 / 2000 the long requests `max_tokens` = 2048 allows (qwen3_tts.py:1187-1191) — those run the chunked
    partial + merge attention.  Per kv: ms per frame and the roofline fraction with that kv's KV bytes."""
    from vox_serve_amd.engine import Qwen3Cfg, Qwen3Engine
    cfg, ps, out = Qwen3Cfg(), 128, {}
    for B in batches:
        ppr = (max(kvs) + frames + 8 + ps - 1) // ps + 1
        eng = Qwen3Engine(cfg, shared["W"], max_batch=B, page_size=ps, max_pages=B * ppr + 1, max_seq_len=2304, max_prefill_rows=128)
        eng.keep_hidden = False
        for b0 in range(0, B * ppr, 16):
            eng.kv[:, b0:b0 + 16].normal_(0, 0.5)
        sc = eng.sampling_cfg(greedy=True)
        eng.input_ids.zero_(); eng.input_ids[:, -1] = cfg.tts_pad_id
        eng.input_masks[:B] = 1

        def plan(kvlen):
            pages = [[b * ppr + j for j in range((kvlen + ps - 1) // ps)] for b in range(B)]
            indptr = np.cumsum([0] + [len(p) for p in pages]); indices = sum(pages, [])
            eng.upload_plan(pos=[kvlen] * B, kvlen=[kvlen] * B, page=[p[-1] for p in pages], slot=[(kvlen - 1) % ps] * B,
                            indptr=indptr, indices=indices)
        res = {}
        for kvl in kvs:
            for w_ in range(3):
                plan(kvl + w_); eng.frame(B, kvl + w_, sc, feedback=True, use_graph=True)
            torch.cuda.synchronize()
            ms = []
            for f in range(frames):
                plan(kvl + 3 + f)
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record(eng.stream)
                eng.frame(B, kvl + 3 + f, sc, feedback=True, use_graph=True)
                ev1.record(eng.stream)
                eng.out_ids[:B].cpu()
                ms.append(ev0.elapsed_time(ev1))
            t = float(np.mean(ms)) * 1e-3
            alg = algorithmic_bytes_per_frame(B, kvl + 3 + frames / 2)
            res[f"kv{kvl}"] = {"frame_ms": t * 1e3, "samples_per_s_lm_only": B * 1920 / t, "roofline_frac": alg / t / 1e9 / HBM_PEAK_GBS,
                               "algorithmic_bytes_per_launch": alg}
        # the same frame with the REFERENCE'S DEFAULT sampling for Qwen3-TTS (top-k 50 at temperature 0.9, qwen3_tts.py:1088-1096) instead of
        # greedy: 16 stochastic draws per frame (one-request frames take the 14 inner ones inside the persistent depth-step launches)
        scs = eng.sampling_cfg(greedy=False, top_k=50, temperature=0.9)
        kvl = 200
        for w_ in range(3):
            plan(kvl + w_); eng.frame(B, kvl + w_, scs, feedback=True, use_graph=True)
        torch.cuda.synchronize()
        ms = []
        for f in range(frames):
            plan(kvl + 3 + f)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(eng.stream)
            eng.frame(B, kvl + 3 + f, scs, feedback=True, use_graph=True)
            ev1.record(eng.stream)
            eng.out_ids[:B].cpu()
            ms.append(ev0.elapsed_time(ev1))
        t = float(np.mean(ms)) * 1e-3
        res["kv200_top_k50_temperature0.9"] = {"frame_ms": t * 1e3, "samples_per_s_lm_only": B * 1920 / t,
                                               "vs_greedy": t * 1e3 / res["kv200"]["frame_ms"] if "kv200" in res else None}
        out[f"batch{B}"] = res
        eng.close()
    out["note"] = ("LM frame alone (no codec chunk beside it), greedy unless named otherwise, random KV pages; frame_ms = HIP events around the "
                   "graph replay")
    return out


def serving_ttfa(dev, shared, n_requests, load, interval=INTERVAL, seed=0):
    """TTFA through the serving path proper: `encode_request` on the scheduler's transport -> first
    `id|AUDIO|` message on the result queue, via Scheduler -> ModelWorker -> engine -> codec.  p50 over `n_requests`
    probe requests, each submitted while `load` other requests are decoding (load 0 = batch-1 streaming)."""
    from vox_serve_amd.model.qwen3_tts import Qwen3TTSModel
    from vox_serve_amd.sampling import SamplingConfig
    from vox_serve_amd.scheduler import QueueTransport, Scheduler, encode_request
    from vox_serve_amd.worker import ModelWorker
    mb = max(8, load + 1)
    m = Qwen3TTSModel("qwen3-tts", shared["W"], shared["codec_W"], device=str(dev), detokenize_interval=interval,
                      max_batch_size=mb, page_size=128, max_num_pages=4 * mb + 8, max_seq_len=2304, max_prefill_tokens=128)
    # alone (load 0) a probe only has to reach its first chunk: a short request keeps the default run within minutes
    m.default_sampling_config = SamplingConfig(greedy=True, max_tokens=(400 if load else PROMPT_TOKENS + interval + 6),
                                               repetition_penalty=1.05, repetition_window=-1)
    t = QueueTransport()
    w = ModelWorker(model=m, max_batch_size=mb, max_num_pages=4 * mb + 8, page_size=128, device=str(dev))
    s = Scheduler(w, max_batch_size=mb, transport=t)
    rng = np.random.default_rng(seed)
    counter = [0]
    first_audio = {}

    def on_send(payload):       # the moment `id|AUDIO|...` is handed to the result transport (a consumer of the queue / socket sees it
        rid_, kind_, _ = payload.split(b"|", 2)       # then; the scheduler goes on to launch the step's LM frame after the send)
        if kind_ == b"AUDIO":
            first_audio.setdefault(rid_.decode(), time.perf_counter())
    t.on_send = on_send

    def submit(tag):
        counter[0] += 1
        rid = f"{tag}{counter[0]}"
        ids = [1, 2, 3] + rng.integers(0, 151000, 64).tolist() + [4, 5, 6, 7, 8]      # 3 role + 64 text + 5 template tail
        t.requests.put(encode_request(rid, "", model_kwargs={"prompt_token_ids": ids, "language": "english"}))
        return rid

    def drain(until_rid=None):
        hit = False
        while not t.results.empty():
            msg = t.results.get()
            rid, kind, _ = msg.split(b"|", 2)
            if until_rid is not None and kind == b"AUDIO" and rid.decode() == until_rid:
                hit = True
        return hit

    out = []
    for i in range(n_requests + 1):
        # keep `load` background requests decoding (a finished one is replaced before the probe goes in)
        while sum(1 for r in s.active_requests if r.request_id.startswith("bg")) + t.requests.qsize() < load:
            submit("bg")
        for _ in range(4 * load + 2):
            s._step()
            drain()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rid = submit("probe")
        while True:
            s._step()
            if drain(rid):
                break
        out.append((first_audio[rid] - t0) * 1e3)
        if load == 0:
            s.run_until_idle(1000)       # let the probe finish so that the next one runs alone
            drain()
    del out[0]
    m.engine.close()
    m.audio_decoder.close()
    return float(np.median(out))


def serving_throughput(dev, shared, n_req, frames, kind="base"):
    """A batch job through the serving path proper: `n_req` requests submitted at once on the scheduler's transport, each
    generating `frames` frames (greedy, max_tokens bounds the length), served by Scheduler / DisaggregationScheduler ->
    ModelWorker -> engine -> codec until every COMPLETION is on the result queue.  value = PCM samples received / wall time
    (prefills, one per step, included).  The hot loop of bench.py's headline (`Loop`) is the same engine + codec calls
    without the scheduler / worker bookkeeping; this is the number with it."""
    from vox_serve_amd.model.qwen3_tts import Qwen3TTSModel
    from vox_serve_amd.sampling import SamplingConfig
    from vox_serve_amd.scheduler import QueueTransport, Scheduler, encode_request
    from vox_serve_amd.scheduler.disaggregation import DisaggregationScheduler
    from vox_serve_amd.worker import ModelWorker
    mb = max(8, n_req)
    m = Qwen3TTSModel("qwen3-tts", shared["W"], shared["codec_W"], device=str(dev), detokenize_interval=INTERVAL,
                      max_batch_size=mb, page_size=128, max_num_pages=4 * mb + 8, max_seq_len=2304, max_prefill_tokens=128)
    m.default_sampling_config = SamplingConfig(greedy=True, max_tokens=PROMPT_TOKENS + frames, repetition_penalty=1.05, repetition_window=-1)
    t = QueueTransport()
    w = ModelWorker(model=m, max_batch_size=mb, max_num_pages=4 * mb + 8, page_size=128, device=str(dev))
    from vox_serve_amd.scheduler.offline import OfflineScheduler
    s = (DisaggregationScheduler(w, max_batch_size=mb, transport=t) if kind == "disaggregation"
         else OfflineScheduler(w, max_batch_size=mb, transport=t) if kind == "offline"
         else Scheduler(w, max_batch_size=mb, transport=t, async_scheduling=(kind == "async"),
                        detokenize_min_batch=(mb // 2 if kind == "batched_detokenize" else 0)))
    rng = np.random.default_rng(3)

    def submit(tag, n):
        for i in range(n):
            ids = [1, 2, 3] + rng.integers(0, 151000, 64).tolist() + [4, 5, 6, 7, 8]
            t.requests.put(encode_request(f"{tag}{i}", "", model_kwargs={"prompt_token_ids": ids, "language": "english"}))

    def run():
        if kind == "disaggregation":
            s.run_until_idle(600.0)
        else:
            s.run_until_idle(1000000)
        n = 0
        while not t.results.empty():
            msg = t.results.get()
            rid, k, body = msg.split(b"|", 2)
            if k == b"AUDIO":
                n += len(body) // 2
        return n
    submit("warm", n_req)                  # the same job once untimed: every batch size of the ramp has its frame / codec graph captured
    run()                                  # (the reference's worker captures all its graph shapes at start-up: cuda_graph_worker.py:383-470)
    torch.cuda.synchronize()
    sent = []                              # (time, request, kind, samples) of every message handed to the result transport

    def on_send(payload):
        rid_, kind_, body_ = payload.split(b"|", 2)
        sent.append((time.perf_counter(), rid_, kind_, len(body_) // 2 if kind_ == b"AUDIO" else 0))
    t.on_send = on_send
    hs0 = dict(w.host_stats)
    t0 = time.perf_counter()
    submit("r", n_req)
    samples = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    hs1 = dict(w.host_stats)
    m.engine.close()
    m.audio_decoder.close()
    res = {"value": samples / dt, "unit": "audio samples/s", "requests": n_req, "frames_per_request": frames, "seconds": dt,
           "scheduler": kind, "ms_per_frame_step_equiv": dt / frames * 1e3,
           "note": "value = the whole job (one prefill per step: an n-step ramp at each end); steady_state = all requests decoding"}
    # steady state: from the moment the LAST request has produced audio until the FIRST completion — every step in between
    # decodes n_req rows (continuous batching at n_req concurrent requests, BASELINE config 3)
    first_audio, t_done = {}, None
    for ts, rid_, kind_, n_ in sent:
        if kind_ == b"AUDIO":
            first_audio.setdefault(rid_, ts)
        elif kind_ == b"COMPLETION" and t_done is None:
            t_done = ts
    if len(first_audio) == n_req and t_done is not None:
        t_all = max(first_audio.values())
        if t_done > t_all:
            n_ss = sum(n_ for ts, _r, kind_, n_ in sent if kind_ == b"AUDIO" and t_all < ts <= t_done)
            v = n_ss / (t_done - t_all)
            res["steady_state"] = {"value": v, "unit": "audio samples/s", "window_s": t_done - t_all,
                                   "ms_per_step": n_req * 1920 / v * 1e3 if v > 0 else None}
    steps = max(1, hs1["steps"] - hs0["steps"])
    res["host_us_per_step"] = {"prepare_lm_inputs": (hs1["prepare_s"] - hs0["prepare_s"]) / steps * 1e6,
                               "update_requests": (hs1["after_s"] - hs0["after_s"]) / steps * 1e6, "steps": steps}
    return res


def serving_pool(n_req, frames, dp_size=1, fake=False):
    """The online data-parallel serving pool (vox_serve_amd/launch.py; the reference's `--dp-size N`: launch.py:183-279, 355-415,
    460-474): `dp_size` scheduler daemons — fresh interpreters pinned to their GPU through HIP_VISIBLE_DEVICES before torch is
    imported — behind the round-robin router, requests and audio over the AF_UNIX PUSH/PULL transports.  n_req requests per daemon
    submitted at once; value = PCM samples received by the client / wall time, TTFA = submit -> first AUDIO message at the client."""
    from vox_serve_amd.launch import ServingPool
    mb = max(8, n_req)
    t0 = time.perf_counter()
    if fake:      # --dry-run: CPU daemons over the real worker host logic with a fake LM (tests/dp_fake_worker.py): the N-daemon code path, no GPU
        pool = ServingPool("fake", dp_size=dp_size, max_batch_size=8, page_size=4, max_num_pages=64, worker_factory="tests.dp_fake_worker:make",
                           log_level="WARNING", ready_timeout_s=300.0, pin_devices=False)
    else:
        pool = ServingPool("qwen3-tts", dp_size=dp_size, max_batch_size=mb, max_num_pages=4 * mb + 8, page_size=128, synthetic=True, greedy=True,
                           max_tokens=PROMPT_TOKENS + frames, async_scheduling=True, log_level="WARNING", ready_timeout_s=240.0)
    startup = time.perf_counter() - t0
    rng = np.random.default_rng(5)
    try:
        def job(tag):
            rids = []
            for i in range(n_req * dp_size):
                ids = [1, 2, 3] + rng.integers(0, 151000, 64).tolist() + [4, 5, 6, 7, 8]
                rids.append(pool.start_streaming_request("x" * (1 + i % 7) if fake else "", model_kwargs={} if fake else {"prompt_token_ids": ids, "language": "english"},
                                                         request_id=f"{tag}{i}", block=True))
            n = sum(len(c) // 2 for rid in rids for c in pool.stream(rid, timeout_s=120))
            infos = [pool.request_info(r) for r in rids]
            for r in rids:
                pool.release(r)
            return n, infos
        job("warm")                          # graph capture of every batch size of the ramp, as in serving_throughput
        t0 = time.perf_counter()
        samples, infos = job("r")
        dt = time.perf_counter() - t0
        ttfa = sorted((i["first_audio_time"] - i["submit_time"]) * 1e3 for i in infos if i.get("first_audio_time"))
        ranks = sorted({i["rank"] for i in infos})
        return {"value": samples / dt, "unit": "audio samples/s", "dp_size": dp_size, "requests": n_req * dp_size, "frames_per_request": frames,
                "seconds": dt, "ttfa_ms_p50_client": ttfa[len(ttfa) // 2] if ttfa else None, "ranks_used": ranks,
                "daemon_startup_s": startup, "transport": "AF_UNIX PUSH/PULL" if pool.transport == "ipc" else "zmq ipc",
                "scheduler": "base + async_scheduling, one daemon per GPU (scheduler_entry.py)"}
    finally:
        pool.cleanup()


CODEC_GFLOP_PER_REQUEST_CHUNK = 49.6      # useful (one-term) multiply-adds x 2 of one request's 10-frame chunk through the Qwen3 12 Hz decoder
MFMA_BF16_PEAK_TFLOPS = 2500.0            # dense bf16 matrix-core peak (MI355X_MICROARCH.md)


def codec_chunk_stats(dev, codec_W):
    """The token->waveform half alone: one 10-frame chunk of B requests through the Qwen3 codec decoder (HIP events, nothing beside it),
    in the default two-term operand mode (waveform within 1e-4 RMS of the reference decoder in fp32) and in the bf16-operand mode (the
    precision the REFERENCE serves this decoder at; RMS 1.2e-2 from its fp32 evaluation, like the reference's own bf16 run).
    mfma_frac = useful FLOPs / chunk time / the dense bf16 matrix-core peak."""
    from vox_serve_amd.tokenizer.qwen3_codec import Qwen3TTSDecoder
    out = {"unit": "ms per 10-frame chunk", "gflop_per_request_chunk": CODEC_GFLOP_PER_REQUEST_CHUNK, "mfma_peak_tflops": MFMA_BF16_PEAK_TFLOPS}
    for prec, key in (("fp32", "two_term"), ("bf16", "bf16_operands")):
        for B in (1, 32):
            dec = Qwen3TTSDecoder(codec_W, device=dev, max_batch=B, max_slots=B, detokenize_interval=INTERVAL, operand_precision=prec)
            codes = torch.randint(0, 2048, (B, 16, INTERVAL))
            cache = dec.init_cache(B)
            for _ in range(3):
                dec.decode_chunk(codes, cache)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                dec.decode_chunk(codes, cache)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            out[f"{key}_b{B}"] = {"ms": ms, "mfma_frac": B * CODEC_GFLOP_PER_REQUEST_CHUNK / ms / MFMA_BF16_PEAK_TFLOPS,
                                  "samples_per_s": B * INTERVAL * 1920 / ms * 1e3}
            dec.release_cache(cache)
            dec.close()
    return out


def other_configs():
    """The other BASELINE.json configs, one GPU each, as sub-results of the same driver-timed command: every tool prints one JSON line
    (LM step + detokenizer in its loop, synthetic weights of the named architecture) and runs in its own process."""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    res = {}
    for name, cmd in (("cosyvoice2_0.5b_flow_hift_b8", ["tools/bench_cosyvoice2.py", "--batch", "8", "--steps", "75"]),
                      ("csm_1b_mimi_b16", ["tools/bench_csm.py", "--batch", "16", "--steps", "60"]),
                      ("glm_4_voice_9b_flow_hift_b8_per_gpu", ["tools/bench_glm.py", "--batch", "8", "--greedy", "--steps", "80"]),
                      ("qwen3_tts_voice_clone_prompt_side", ["tools/bench_clone.py", "--reps", "10"])):
        try:
            p = subprocess.run([sys.executable, os.path.join(here, cmd[0])] + cmd[1:], capture_output=True, text=True, timeout=300, cwd=here)
            line = [ln for ln in p.stdout.strip().split("\n") if ln.startswith("{")]
            res[name] = json.loads(line[-1]) if line else {"error": (p.stderr or "no output")[-200:]}
        except Exception as ex:          # a sub-result must never hide the headline
            res[name] = {"error": repr(ex)[:200]}
    return res


def spawn_ranks(args):
    """`python bench.py --gpus N` with no launcher: start N ranks of this script under torch.distributed.run."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=None, help="concurrent requests per GPU of the headline value (default 1, with "
                    "batch 8 and 32 as sub-results; an explicit --batch runs that batch size only)")
    ap.add_argument("--sub-batches", type=str, default=None, help="comma list of extra batch sizes reported as batchN objects")
    ap.add_argument("--kv-mean", type=float, default=200.0, help="mean KV length over the timed steps (SURVEY 8d: 200)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ttfa-requests", type=int, default=5, help="engine-level TTFA samples per setting (lock-step loop)")
    ap.add_argument("--serving-ttfa-requests", type=int, default=100, help="TTFA samples through Scheduler + ModelWorker (0 = skip)")
    ap.add_argument("--serving-modes", type=str, default="ttfa,ttfa2,load,throughput,pool", help="which serving-path measurements run "
                    "(comma list of ttfa, ttfa2 = detokenize_interval 2, load = TTFA under 32-way load, throughput, pool = through the "
                    "DP serving pool's daemon + sockets)")
    ap.add_argument("--force-dist", action="store_true", help="initialise the RCCL process group and run the weight broadcast / "
                    "all-reduce / barrier path even at world size 1 (exercises the multi-GPU code on one GPU)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the BASELINE configs 1, 3, 4 sub-results (CosyVoice2, CSM-1B, GLM-4-Voice)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: gloo on CPU, a stub loop instead of the engine — executes the rank launch, "
                    "weight broadcast, barrier / max-over-ranks timing and JSON path of `--gpus N` (tests/test_host_logic.py); prints no measurement")
    ap.add_argument("--pipelined", action="store_true", help="frames pipelined one deep in the headline loop (the reference's async-scheduling contract: "
                    "frame N+1 is enqueued before the host reads frame N's tokens); default: lock-step as in rounds 1-3, with the pipelined number of the "
                    "same run as the `pipelined` sub-result")
    ap.add_argument("--no-kv-sweep", action="store_true", help="skip the kv_sweep sub-result (LM frame at kv 100 / 200 / 325 / 1000 / 2000)")
    ap.add_argument("--serving-frames", type=int, default=250, help="frames per request of the serving-path jobs (SURVEY 8d: 250)")
    ap.add_argument("--exact-rows", type=int, default=None, help="rows up to which linears use the wave64 VALU kernels instead of the "
                    "matrix cores (library default 2; 1..8).  Every setting is bit-exact against the oracle under the same policy")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dry = args.dry_run
    if dry:
        dev = torch.device("cpu")
    else:
        if local >= torch.cuda.device_count():
            raise SystemExit(f"rank {rank}: local rank {local} but only {torch.cuda.device_count()} GPU(s) visible")
        torch.cuda.set_device(local)                            # one process per GPU, bound before any allocation
        dev = torch.device("cuda", local)
    sync = (lambda: None) if dry else torch.cuda.synchronize
    import torch.distributed as dist
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)     # RCCL over xGMI; off the token path

    if args.exact_rows is not None:
        from vox_serve_amd import _native as N
        N.set_exact_rows(args.exact_rows)
    if dry:
        gen = torch.Generator().manual_seed(0 if rank == 0 else 1 + rank)
        shared = {"W": {f"w{i}": torch.randn(257, 33, generator=gen).to(torch.bfloat16) for i in range(9)}, "codec_W": None}
    else:
        from vox_serve_amd.engine import Qwen3Cfg
        from vox_serve_amd.synth import synth_qwen3_codec_weights, synth_qwen3_weights
        shared = {"W": synth_qwen3_weights(Qwen3Cfg(), dev, seed=0 if rank == 0 else 1 + rank), "codec_W": synth_qwen3_codec_weights(seed=0)}
    bcast = None
    if use_dist:
        # load-time weight distribution of the DP pool: rank 0's arena -> every replica over RCCL (worker/dp_pool.py);
        # the other ranks start from different random weights, so a wrong broadcast would show in their outputs
        from vox_serve_amd.worker.dp_pool import broadcast_weights
        nbytes = sum(v.numel() * v.element_size() for v in shared["W"].values())
        sync(); dist.barrier()
        t0 = time.perf_counter()
        broadcast_weights(shared["W"], force=args.force_dist)
        sync(); dist.barrier()
        bdt = time.perf_counter() - t0
        chk = torch.stack([shared["W"][k].float().sum() for k in sorted(shared["W"])[:8]]).to(torch.float64)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        bcast = {"bytes": nbytes, "seconds": bdt, "GBps": nbytes / bdt / 1e9, "replicas_identical": bool(torch.equal(lo, hi))}
    ranks_seen = world
    if use_dist:
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)
        ranks_seen = int(one.item())

    head_B = args.batch if args.batch is not None else 1
    subs = [] if args.batch is not None else [8, 32]
    if args.sub_batches is not None:
        subs = [int(x) for x in args.sub_batches.split(",") if x.strip()]
    _phase("weights")
    head = run_batch(head_B, args, dev, world, shared, ttfa_requests=args.ttfa_requests if rank == 0 else 0, lockstep=not args.pipelined)
    _phase(f"batch {head_B}")
    head_pipe = None
    if not args.pipelined and world == 1 and not dry:
        head_pipe = run_batch(head_B, args, dev, world, shared, lockstep=False)      # the same loop with frames pipelined one deep
        _phase(f"batch {head_B} pipelined")
    sub_res = {}
    for b in subs:
        sub_res[b] = run_batch(b, args, dev, world, shared, lockstep=not args.pipelined)
        _phase(f"batch {b}")

    if use_dist:
        # every collective of the run is behind us (weight broadcast, the barriers and max-reduce around each timed region): the group
        # is torn down by all ranks together, here — what follows (serving-path sub-results, the N-daemon pool) runs on rank 0 alone
        dist.barrier()
        dist.destroy_process_group()
        use_dist = False
    if rank != 0:
        return
    kv_sweep_res = None
    if rank == 0 and world == 1 and args.batch is None and not args.no_kv_sweep and not dry:
        kv_sweep_res = kv_sweep(dev, shared)
        _phase("kv sweep")
    serving = {}
    modes = {m.strip() for m in args.serving_modes.split(",") if m.strip()}
    if world == 1 and args.serving_ttfa_requests > 0 and args.batch is None and not dry:
        n = args.serving_ttfa_requests
        if "ttfa" in modes:
            serving["ttfa_ms_p50"] = serving_ttfa(dev, shared, n, 0)
            _phase("serving ttfa")
        if "ttfa2" in modes:
            serving["ttfa_ms_p50_detokenize_interval_2"] = serving_ttfa(dev, shared, max(10, n // 2), 0, interval=2)
            _phase("serving ttfa interval 2")
        if "load" in modes:
            serving["ttfa_ms_p50_under_32way_load"] = serving_ttfa(dev, shared, max(10, n // 5), 31)
            _phase("serving ttfa under load")
        if "throughput" in modes:
            serving["throughput"] = {}
            for k in ("base", "async", "disaggregation", "offline", "batched_detokenize"):
                serving["throughput"][k] = serving_throughput(dev, shared, 32, args.serving_frames, k)
                _phase(f"serving throughput {k}")
            serving["throughput"]["batch1_base"] = serving_throughput(dev, shared, 1, min(args.serving_frames, 120), "base")
            _phase("serving throughput batch 1")
        if "pool" in modes:
            try:
                serving["pool"] = serving_pool(8, 100)
            except Exception as ex:          # a sub-result must never hide the headline
                serving["pool"] = {"error": repr(ex)[:300]}
            _phase("serving pool")
    if world > 1 and rank == 0 and "pool" in modes and args.batch is None:
        # N > 1: the serving mode of the multi-GPU run goes through the online pool — one scheduler daemon per GPU behind the round-robin
        # router (the replicas of the timed region above are done; their ranks only wait for this rank's line).  Never on the token path
        # of the headline value; a failure here is reported, not raised.
        try:
            serving["pool"] = serving_pool(8, 100, dp_size=world, fake=dry)
        except Exception as ex:
            serving["pool"] = {"error": repr(ex)[:300]}
        _phase(f"serving pool dp{world}")

    if rank == 0:
        out = {
            "metric": "audio samples/sec, Qwen3-TTS-1.7B streaming decode + codec",
            "value": head["value"], "unit": "audio samples/s", "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"Qwen3-TTS-1.7B bf16, batch={head_B}/GPU streaming, greedy, {PROMPT_TOKENS}-token prompt, "
                                   f"detokenize_interval {INTERVAL}, page_size 128, mean kv {head['kv_mean']:.0f}"
                                   + ("" if args.exact_rows is None else f", exact_rows {args.exact_rows}"), "batch_per_gpu": head_B,
                       "frames_per_request": args.steps, "parallelism": f"dp{world} (independent replicas, no collective on the data path)"},
            "realtime_factor": head["realtime_factor_per_request"],
            "ttfa_ms_p50": serving.get("ttfa_ms_p50", head.get("ttfa_ms_p50_engine")),
            "ttfa_ms_p50_detokenize_interval_2": serving.get("ttfa_ms_p50_detokenize_interval_2", head.get("ttfa_ms_p50_engine_detokenize_interval_2")),
            "ttfa_path": ("encode_request -> Scheduler -> ModelWorker -> first id|AUDIO| handed to the result transport (send time), p50" if serving
                          else "engine lock-step loop (request start -> first PCM chunk on the host), p50"),
            "roofline": head["roofline"],
        }
        for k in ("ttfa_ms_p50_under_32way_load",):
            if k in serving:
                out[k] = serving[k]
        if "throughput" in serving:
            out["serving_path_throughput"] = serving["throughput"]
        if "pool" in serving:
            out["serving_pool_dp"] = serving["pool"]
        for k in ("ttfa_ms_p50_engine", "ttfa_ms_p50_engine_detokenize_interval_2", "depth_persist"):
            if k in head:
                out[k] = head[k]
        out["loop"] = head["loop"]
        if head_pipe is not None:      # frame N+1 enqueued before the host reads frame N's tokens (stream-ordered pinned snapshot, read one step
            out["pipelined"] = {k: head_pipe[k] for k in ("value", "ms_per_step", "loop")}      # late: scheduler/base.py:166-221); same work per step
            out["pipelined"]["roofline_frac"] = head_pipe["roofline"]["frac"]
        for b, r in sub_res.items():
            out[f"batch{b}"] = {k: r[k] for k in ("value", "ms_per_step", "batch_per_gpu", "realtime_factor_per_request", "kv_mean", "roofline", "loop")}
        if kv_sweep_res is not None:
            out["kv_sweep"] = kv_sweep_res
        if bcast:
            out["weight_broadcast_rccl"] = bcast
        if dry:
            out["dry_run"] = True
            out["data"] = "none (dry run: stub loop, gloo, no GPU) - not a measurement"
        if world == 1 and args.batch is None and not args.no_other_configs and not dry:
            try:
                out["codec_chunk"] = codec_chunk_stats(dev, shared["codec_W"])
            except Exception as ex:          # a sub-result must never hide the headline
                out["codec_chunk"] = {"error": repr(ex)[:300]}
            _phase("codec chunk")
            out["other_configs"] = other_configs()
            _phase("other configs")
        if not args.no_cpu_baseline and world == 1 and not dry:      # the CPU leg runs on rank 0 at N=1 only (other ranks would idle in RCCL)
            try:
                out["cpu_baseline"] = cpu_baseline(shared["W"])
                _phase("cpu baseline")
            except Exception as ex:  # the baseline is a reported extra; never let it hide the GPU number
                out["cpu_baseline"] = {"value": None, "error": repr(ex)[:200]}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
