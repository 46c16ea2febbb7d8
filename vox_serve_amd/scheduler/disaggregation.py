"""DisaggregationScheduler — LM and detokenizer run concurrently (drop-in for
/root/reference/vox_serve/scheduler/disaggregation.py:14-300: two asyncio loops, LM on GPU 0, codec on GPU 1).

Here the two pipelines are two threads of the one process that owns the GPU(s): the LM thread drives the frame graphs
on the engine's stream, the detokenizer thread drives the codec on its own HIP stream (same GPU, or the worker's
`detokenizer_device` when a second GPU is given) — libvoxhip is re-entrant per context and every call takes a stream,
so no global lock is needed on the device side.  Requests cross over through a queue exactly once per window:
`_queue_detokenize_requests` applies the reference's readiness rule and an in-flight set keeps a request from being
queued twice; completion (KV + codec-slot release, removal from the active list) happens on the detokenizer side under
the request-list lock."""
import json
import queue
import threading
import time
from typing import List, Set

from ..requests import Request
from ._chunks import ChunkCursor
from .base import Scheduler


class DisaggregationScheduler(Scheduler):
    def __init__(self, model_worker, max_batch_size: int = 8, transport=None, **kwargs):
        super().__init__(model_worker, max_batch_size=max_batch_size, transport=transport, async_scheduling=True, **kwargs)
        self.detokenize_queue: "queue.Queue[Request]" = queue.Queue()
        self.requests_lock = threading.RLock()
        self.detokenizing_request_ids: Set[str] = set()
        self._stop = threading.Event()
        self._threads: List[threading.Thread] = []
        self.errors: List[BaseException] = []

    # ---- LM side ----
    def _queue_detokenize_requests(self):
        w = self.model_worker
        with self.requests_lock:
            for req in self.active_requests:
                if req.request_id in self.detokenizing_request_ids:
                    continue
                cur = ChunkCursor(req, w.detokenize_interval, w.detokenize_overlap)
                if cur.ready():
                    req.next_audio_decode_idx = [cur.next]
                    self.detokenizing_request_ids.add(req.request_id)
                    self.detokenize_queue.put(req)

    def _lm_step(self) -> bool:
        with self.requests_lock:
            self._prepare_requests_locked()
        self._queue_detokenize_requests()
        with self.requests_lock:
            lm_requests = self._select_lm_requests()
        if not lm_requests:
            return False
        lm_inputs = self.model_worker.prepare_lm_inputs(lm_requests, [])
        if lm_inputs is not None and lm_inputs["is_prefill"]:
            self.model_worker.run_lm_prefill(lm_requests, lm_inputs)
        else:
            self.model_worker.run_lm_decode(lm_requests, lm_inputs)
        return True

    def _prepare_requests_locked(self):
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
            except Exception as e:
                self.logger.error(f"Error receiving requests: {e}")

    # ---- detokenizer side ----
    def _get_detokenize_batch(self) -> List[Request]:
        batch: List[Request] = []
        for _ in range(self.max_batch_size):
            try:
                batch.append(self.detokenize_queue.get(timeout=0.001) if not batch else self.detokenize_queue.get_nowait())
            except queue.Empty:
                break
        return batch

    def _detokenize_step(self) -> bool:
        batch = self._get_detokenize_batch()
        if not batch:
            return False
        for req in batch:
            req.audio_decode_idx = list(req.next_audio_decode_idx)
        self.model_worker.run_detokenize(batch)
        self._send_responses_locked(batch)
        for req in batch:
            self.detokenizing_request_ids.discard(req.request_id)
        return True

    def _send_responses_locked(self, detokenize_requests: List[Request]):
        for req in detokenize_requests:
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
                with self.requests_lock:
                    self.model_worker.free_kv_cache(req)
                    if req in self.active_requests:
                        self.active_requests.remove(req)
                msg = {"status": "completed", "reason": req.finish_reason or "unknown"}
                self.transport.send_result(req.request_id.encode("utf-8") + b"|COMPLETION|" + json.dumps(msg).encode("utf-8"))

    # ---- the two loops ----
    def _loop(self, step):
        try:
            while not self._stop.is_set():
                if not step():
                    time.sleep(0.001)
        except BaseException as e:        # surface worker errors to run_until_idle / the caller
            self.errors.append(e)
            self._stop.set()

    def start(self):
        self._stop.clear()
        self._threads = [threading.Thread(target=self._loop, args=(self._lm_step,), name="vox-lm", daemon=True),
                         threading.Thread(target=self._loop, args=(self._detokenize_step,), name="vox-detokenize", daemon=True)]
        for t in self._threads:
            t.start()

    def stop(self):
        self._stop.set()
        for t in self._threads:
            t.join(timeout=10)
        if self.errors:
            raise self.errors[0]

    def run_forever(self):
        self.start()
        try:
            while not self._stop.is_set():
                time.sleep(0.05)
        finally:
            self.stop()

    def run_until_idle(self, timeout_s: float = 120.0):
        self.start()
        t0 = time.time()
        try:
            while time.time() - t0 < timeout_s and not self._stop.is_set():
                with self.requests_lock:
                    idle = not self.active_requests and not self.transport.pending()
                if idle and self.detokenize_queue.empty() and not self.detokenizing_request_ids:
                    return
                time.sleep(0.002)
        finally:
            self.stop()
