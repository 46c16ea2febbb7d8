"""Continuous-batching scheduler — policy and wire format of /root/reference/vox_serve/scheduler/base.py:
  _step                         :135-165   detokenize-select -> lm-select -> prepare -> detokenize -> send -> prefill|decode
  _select_lm_requests           :234-300   at most ONE new prefill per step, piggy-backed by <= 7 decode rows; else all decodes
  _select_detokenize_requests   :302-333
  _send_responses               :335-363   `id|AUDIO|<pcm16le>` and `id|COMPLETION|{json}`
  _handle_request_payload       :404-429   `{json}|audio_data_placeholder`
The transport is pluggable: ZeroMQ PUSH/PULL over ipc:// exactly like the reference when pyzmq is installed
(scheduler/base.py:103-125), in-process queues otherwise (tests, benchmarks).
"""
import json
import logging
import queue
import time
from typing import List, Optional

from ..requests import Request


class QueueTransport:
    """In-process stand-in for the two ZeroMQ sockets (same non-blocking recv / send semantics)."""

    def __init__(self):
        self.requests, self.results = queue.Queue(), queue.Queue()
        self.on_send = None      # optional callback(payload) at the moment a result is handed to the transport (latency probes)

    def recv_request(self) -> Optional[bytes]:
        try:
            return self.requests.get_nowait()
        except queue.Empty:
            return None

    def send_result(self, payload: bytes):
        self.results.put(payload)
        if self.on_send is not None:
            self.on_send(payload)

    def pending(self) -> bool:
        return not self.requests.empty()


class ZmqTransport:
    def __init__(self, request_socket_path="/tmp/vox_serve_request.ipc", result_socket_path="/tmp/vox_serve_result.ipc"):
        import zmq
        self.zmq = zmq
        ctx = zmq.Context()
        self.request_socket = ctx.socket(zmq.PULL)
        self.request_socket.setsockopt(zmq.RCVHWM, 256)
        self.request_socket.bind(f"ipc://{request_socket_path}")
        self.result_socket = ctx.socket(zmq.PUSH)
        self.result_socket.setsockopt(zmq.SNDHWM, 1024)
        self.result_socket.setsockopt(zmq.LINGER, 0)
        self.result_socket.connect(f"ipc://{result_socket_path}")

    def recv_request(self):
        try:
            return self.request_socket.recv(flags=self.zmq.NOBLOCK)
        except self.zmq.Again:
            return None

    def send_result(self, payload: bytes):
        self.result_socket.send(payload)

    def pending(self) -> bool:
        return bool(self.request_socket.poll(0))


class Scheduler:
    def __init__(self, model_worker, max_batch_size: int = 8, transport=None, async_scheduling: bool = False, **kwargs):
        self.model_worker = model_worker
        self.max_batch_size = max_batch_size
        self.transport = transport or QueueTransport()
        self.async_scheduling = async_scheduling
        self.detokenize_min_batch = int(kwargs.get("detokenize_min_batch", self.detokenize_min_batch))
        self.active_requests: List[Request] = []
        self.logger = logging.getLogger(__name__)
        self.available_batch_sizes = model_worker.available_batch_sizes
        self.sample_rate, self.bytes_per_sample, self.channels = 24000, 2, 1
        # in-process counters (the reference measures these from the client side: benchmark/goodput.py:250-262,
        # benchmark/throughput.py:312-318): time to first audio per request, audio samples sent, completions
        self._arrival = {}
        self.stats = {"started": time.time(), "requests": 0, "completed": 0, "failed": 0, "samples": 0, "ttfa_s": []}

    # ---- one iteration of the hot loop ----
    def _step(self):
        self._prepare_requests()
        detokenize_requests = self._select_detokenize_requests()
        lm_requests = self._select_lm_requests()
        fresh = [r for r in lm_requests if not r.done_lm_prefill]
        lm_inputs = None
        for _attempt in range(len(lm_requests) + 2):
            try:
                lm_inputs = self.model_worker.prepare_lm_inputs(lm_requests, detokenize_requests)
                break
            except queue.Empty as ex:
                # KV pages exhausted (worker.OutOfPages, raised before any state changed): the new prompt waits for a later
                # step; if even the decode rows cannot grow, the youngest of them is dropped so that the others go on
                if not getattr(self.model_worker, "defers_on_page_exhaustion", False):
                    raise
                if fresh:
                    lm_requests = [r for r in lm_requests if r.done_lm_prefill]
                    fresh = []
                elif lm_requests:
                    self._fail_requests(lm_requests[-1:], ex)
                    lm_requests = lm_requests[:-1]
                else:
                    break
            except Exception as ex:      # a request the worker cannot even stage (bad kwargs, missing tokenizer, ...)
                self._fail_requests(fresh or lm_requests, ex)
                for req in detokenize_requests:
                    req.audio_decode_idx = req.next_audio_decode_idx.copy()
                lm_requests, lm_inputs = [], None
                break
        pending_audio = self._launch_detokenize(detokenize_requests)
        try:
            if lm_inputs is not None and lm_inputs["is_prefill"]:
                self.model_worker.run_lm_prefill(lm_requests, lm_inputs)
            else:
                self.model_worker.run_lm_decode(lm_requests, lm_inputs)
        except Exception as ex:
            # one bad request must not take the serving loop down (the reference lets the exception escape run_forever):
            # in a prefill step the new prompt is the suspect, in a decode step every row of the failed launch is dropped
            self._fail_requests(fresh or lm_requests, ex)
            self._undo_surviving_rows(fresh, lm_requests)
        self._finish_detokenize(pending_audio, detokenize_requests)

    def _undo_surviving_rows(self, fresh, lm_requests):
        """Decode rows piggy-backed onto a prefill step whose launch failed: their bookkeeping was advanced, nothing ran."""
        if fresh and hasattr(self.model_worker, "undo_decode_advance"):
            self.model_worker.undo_decode_advance([r for r in lm_requests if r not in fresh])

    # ---- detokenize beside the LM frame ----
    overlap_detokenize = True       # False: the reference's order (decode the audio, send it, then launch the LM step)

    def _launch_detokenize(self, detokenize_requests):
        """Enqueue the codec chunk of this step.  The reference decodes and sends the audio before it launches the LM step
        (scheduler/base.py:125-160), i.e. the GPU runs the two one after the other; here the chunk is enqueued on the
        worker's detokenize stream, the LM step is launched right behind it and the audio is collected after the step, so
        both share the GPU.  A chunk that carries a request's FIRST audio (time to first audio) is still collected and sent
        at once.  Returns the handle for `_finish_detokenize`, or None when everything was already sent."""
        w = self.model_worker
        if not detokenize_requests:
            return None
        if not (self.overlap_detokenize and hasattr(w, "launch_detokenize")):
            w.run_detokenize(detokenize_requests)
            self._send_responses(detokenize_requests)
            return None
        pending = w.launch_detokenize(detokenize_requests)
        if any(r.audio_decode_idx and r.audio_decode_idx[0] == 0 for r in detokenize_requests):
            w.finish_detokenize(pending)
            self._send_responses(detokenize_requests)
            return None
        return pending

    def _finish_detokenize(self, pending, detokenize_requests):
        if pending is not None:
            self.model_worker.finish_detokenize(pending)
            self._send_responses(detokenize_requests)

    def _fail_requests(self, requests, ex):
        for req in requests:
            self.logger.error(f"request {req.request_id} failed: {ex!r}")
            req.done_lm_prefill = req.done_lm_generation = req.done_all = True
            req._failed = True           # its error COMPLETION is sent here: _send_responses must not send a second one
            req.finish_reason = f"error: {type(ex).__name__}"
            self.stats["failed"] += 1
            self._arrival.pop(req.request_id, None)
            try:
                self.model_worker.free_kv_cache(req)
            except Exception:            # never mask the original failure
                pass
            msg = {"status": "error", "reason": req.finish_reason, "detail": str(ex)[:200]}
            self.transport.send_result(req.request_id.encode("utf-8") + b"|COMPLETION|" + json.dumps(msg).encode("utf-8"))
        self.active_requests = [r for r in self.active_requests if not r.done_all]

    # ---- async scheduling: the reference's pipelined loop (scheduler/base.py:166-221) ----
    async def _step_async(self, task, lm_requests, detokenize_requests):
        """Launch the step selected LAST iteration, then (while it runs on the GPU) finish the previous step's request-state
        update and select the next step.  A request's EOS / max_tokens is therefore seen one step late (one surplus row,
        dropped by the worker), exactly as in the reference."""
        self._prepare_requests()
        fresh = [r for r in lm_requests if not r.done_lm_prefill]
        lm_inputs = None
        try:
            lm_inputs = self.model_worker.prepare_lm_inputs(lm_requests, detokenize_requests)
        except queue.Empty as ex:              # no KV pages: drop the new prompt from this step (it stays queued)
            if not getattr(self.model_worker, "defers_on_page_exhaustion", False):
                raise
            lm_requests = [r for r in lm_requests if r.done_lm_prefill]
            lm_inputs = self.model_worker.prepare_lm_inputs(lm_requests, detokenize_requests) if lm_requests else None
        except Exception as ex:
            self._fail_requests(fresh or lm_requests, ex)
            lm_requests = []
        pending_audio = self._launch_detokenize(detokenize_requests)
        next_task = None
        try:
            if lm_inputs is not None and lm_inputs["is_prefill"]:
                next_task = self.model_worker.run_lm_prefill(lm_requests, lm_inputs)
            else:
                next_task = self.model_worker.run_lm_decode(lm_requests, lm_inputs)
        except Exception as ex:
            self._fail_requests(fresh or lm_requests, ex)
            self._undo_surviving_rows(fresh, lm_requests)
        self._finish_detokenize(pending_audio, detokenize_requests)
        if task is not None:
            await task                          # step N-1's tokens -> request objects, while step N runs
        if next_task is not None and lm_inputs is not None and lm_inputs["is_prefill"]:
            await next_task                     # a prefill's first frame decides the request's next inputs: not pipelined
            next_task = None
        # requests completed by this iteration's detokenize / send must not be selected again (the synchronous loop prunes
        # them in _prepare_requests before it selects)
        self.active_requests = [r for r in self.active_requests if not r.done_all]
        detok = self._select_detokenize_requests()
        return next_task, self._select_lm_requests(), detok

    async def _run_async_loop(self, until_idle: bool = False, max_steps: int = 1 << 62):
        import asyncio
        task, lm_requests, detokenize_requests = None, [], []
        self.model_worker.async_scheduling = True       # run_lm_* now hand back the request-state update as a coroutine
        try:
            for _ in range(max_steps):
                task, lm_requests, detokenize_requests = await self._step_async(task, lm_requests, detokenize_requests)
                await asyncio.sleep(0)
                if until_idle and not self.active_requests and not self.transport.pending() and not lm_requests:
                    break
            if task is not None:
                await task
        finally:
            if hasattr(self.model_worker, "drain"):
                self.model_worker.drain()
            self.model_worker.async_scheduling = False

    idle_sleep_s = 0.0          # > 0: a step that finds nothing to do sleeps this long (the reference spins; the DP daemons set 0.5 ms)

    def run_forever(self):
        if self.async_scheduling:
            import asyncio
            asyncio.run(self._run_async_loop())
        while True:
            self._step()
            if self.idle_sleep_s > 0 and not self.active_requests and not self.transport.pending():
                time.sleep(self.idle_sleep_s)

    def run_until_idle(self, max_steps: int = 100000):
        if self.async_scheduling:
            import asyncio
            asyncio.run(self._run_async_loop(until_idle=True, max_steps=max_steps))
            return
        for _ in range(max_steps):
            self._step()
            if not self.active_requests and not self.transport.pending():
                return

    # ---- selection policies ----
    prefill_len_default = 0        # budget of a prompt whose length is not known yet (offline.py:44 uses 200)

    def _lm_candidates(self):
        """(prefill, decode) requests that may run an LM step now, in arrival order."""
        prefill, decode = [], []
        for req in self.active_requests:
            if req.done_lm_generation:
                continue
            (decode if req.done_lm_prefill else prefill).append(req)
        return prefill, decode

    def _order_decodes(self, decodes):
        return decodes

    def _select_lm_requests(self):
        """At most ONE new prefill per step (scheduler/base.py:281-282), piggy-backed by decode rows up to the prefill
        batch size; without a prefill, decode rows up to max_batch_size."""
        max_prefill_batch_size = getattr(self.model_worker, "prefill_graph_batch_size", self.max_batch_size)
        max_seq_len = max(getattr(self.model_worker, "cuda_graph_seq_len_buckets", [1024]))
        prefill, decode = self._lm_candidates()
        picked = []
        if prefill:
            first = prefill[0]
            n = first.input_length if first.input_length else self.prefill_len_default
            if max_prefill_batch_size >= 1 and n <= max_seq_len:
                picked.append(first)
            slots = max_prefill_batch_size - len(picked)
        else:
            slots = self.max_batch_size
        cap = self._lm_batch_cap(bool(prefill), max_prefill_batch_size)
        for req in self._order_decodes(decode)[: max(0, slots)]:
            if len(picked) >= cap:
                break
            picked.append(req)
        return picked

    def _lm_batch_cap(self, prefill_cycle: bool, max_prefill_batch_size: int) -> int:
        return self.max_batch_size

    # Opt-in (0 = the reference's policy: every request with a full window is decoded in the step it becomes ready).  With n > 1 a
    # ready window waits until n requests (or all that are still generating) have one, or until it is a request's first
    # audio / its tail / a second window has piled up: requests that started in different steps reach their window
    # boundaries in different steps, and a codec call for 3 rows costs a quarter of one for 32 (bench.py: serving_path_throughput).
    detokenize_min_batch = 0

    def _select_detokenize_requests(self):
        interval = self.model_worker.detokenize_interval
        step = interval - self.model_worker.detokenize_overlap
        cand, urgent = [], False
        for req in self.active_requests:
            if len(cand) >= self.max_batch_size:
                break
            nxt = req.next_audio_decode_idx[-1] + step if req.next_audio_decode_idx else 0
            n_tok = len(req.lm_output_audio_tokens)
            if req.done_lm_generation:
                cand.append((req, nxt, True))
                urgent = True
            elif nxt + interval <= n_tok:
                cand.append((req, nxt, False))
                urgent = urgent or nxt == 0 or nxt + step + interval <= n_tok
        if self.detokenize_min_batch > 1 and not urgent:
            generating = sum(1 for r in self.active_requests if not r.done_lm_generation)
            if len(cand) < min(self.detokenize_min_batch, generating):
                return []
        out = []
        for req, nxt, tail in cand:
            if tail:
                if nxt < len(req.lm_output_audio_tokens):
                    req.next_audio_decode_idx = [nxt]
                else:
                    req.done_all = True
            else:
                req.next_audio_decode_idx = [nxt]
            out.append(req)
        return out

    # ---- wire format ----
    def _send_responses(self, detokenize_requests):
        for req in detokenize_requests:
            if getattr(req, "_failed", False):       # failed in this iteration's LM launch: already answered and freed
                while not req.output_audio.empty():
                    req.output_audio.get()
                continue
            while not req.output_audio.empty():
                chunk = req.output_audio.get()
                if req.is_streaming:
                    req.chunk_send_timestamps.append(time.time())
                    req.chunk_durations.append(self._calculate_chunk_duration(chunk))
                self.transport.send_result(req.request_id.encode("utf-8") + b"|AUDIO|" + chunk)
                t0 = self._arrival.pop(req.request_id, None)
                if t0 is not None:
                    self.stats["ttfa_s"].append(time.time() - t0)
                self.stats["samples"] += len(chunk) // (self.channels * self.bytes_per_sample)
            if req.done_all:
                self.stats["completed"] += 1
                self._arrival.pop(req.request_id, None)
                self.model_worker.free_kv_cache(req)
                msg = {"status": "completed", "reason": req.finish_reason or "unknown"}
                self.transport.send_result(req.request_id.encode("utf-8") + b"|COMPLETION|" + json.dumps(msg).encode("utf-8"))

    def metrics(self) -> dict:
        """p50 / p95 time to first audio (request received -> first AUDIO message sent), audio samples/s and real-time factor
        since the scheduler started."""
        tt = sorted(self.stats["ttfa_s"])
        dt = max(time.time() - self.stats["started"], 1e-9)
        pick = lambda q: (tt[min(len(tt) - 1, int(q * len(tt)))] * 1e3 if tt else None)
        return {"requests": self.stats["requests"], "completed": self.stats["completed"], "failed": self.stats["failed"],
                "ttfa_ms_p50": pick(0.5), "ttfa_ms_p95": pick(0.95), "audio_samples_per_s": self.stats["samples"] / dt,
                "realtime_factor": self.stats["samples"] / self.sample_rate / dt}

    def _calculate_chunk_duration(self, audio_chunk: bytes) -> float:
        return len(audio_chunk) // (self.channels * self.bytes_per_sample) / self.sample_rate

    def _handle_request_payload(self, message_payload: bytes):
        pos = message_payload.find(b"|")
        if pos == -1:
            self.logger.warning(f"Received malformed audio message: {message_payload[:50]}...")
            return None
        d = json.loads(message_payload[:pos].decode("utf-8"))
        return Request(request_id=d["request_id"], prompt=d["prompt"],
                       audio_path=d.get("audio_path") if self.model_worker.supports_audio_input else None,
                       is_streaming=d.get("is_streaming", False), is_pressing=d.get("is_streaming", False),
                       model_kwargs=d.get("model_kwargs", {}))

    def _prepare_requests(self):
        while True:
            payload = self.transport.recv_request()
            if payload is None:
                break
            try:
                req = self._handle_request_payload(payload)
                if req:
                    self.active_requests.append(req)
                    self._arrival[req.request_id] = time.time()
                    self.stats["requests"] += 1
            except Exception as e:          # malformed request: log and keep serving (scheduler/base.py:444-449)
                self.logger.error(f"Error receiving requests: {e}")
        self.active_requests = [r for r in self.active_requests if not r.done_all]


def encode_request(request_id: str, prompt: str, is_streaming=True, model_kwargs=None, audio_path=None) -> bytes:
    """Client side of the wire format (launch.py:522-531)."""
    return json.dumps({"request_id": request_id, "prompt": prompt, "audio_path": audio_path, "is_streaming": is_streaming,
                       "model_kwargs": model_kwargs or {}}).encode("utf-8") + b"|audio_data_placeholder"
