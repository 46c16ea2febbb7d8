"""Online data-parallel serving pool: one scheduler daemon per GPU, a round-robin router in front of them
(the non-HTTP part of /root/reference/vox_serve/launch.py — `_start_schedulers` :176-353, `_process_messages` :355-415,
`_stop_scheduler` :417-447, `_sender_loop` :460-495, `start_streaming_request` :497-541; the FastAPI app, file uploads
and wav encoding are out of scope).

    pool = ServingPool("qwen3-tts", dp_size=8, max_batch_size=32, synthetic=True)
    rid = pool.start_streaming_request("hello", model_kwargs={...})
    for pcm in pool.stream(rid): ...          # `id|AUDIO|` payloads in order, ends at `id|COMPLETION|`
    pool.cleanup()

Topology (as the reference): request i goes to rank `i % dp_size` over that rank's own request transport
(`<request_socket_path>_<rank>`) and stays there for its whole life — no collective and no GPU-to-GPU traffic on the
token path; every daemon pushes `id|AUDIO|pcm` / `id|COMPLETION|json` to ONE shared result transport, demultiplexed here
by request id.  Each daemon is a fresh interpreter started with HIP_VISIBLE_DEVICES set to its GPU (scheduler_entry.py
imports torch only afterwards).  This module imports neither torch nor the HIP library: the router owns no GPU.
"""
import atexit
import collections
import json
import logging
import os
import queue
import subprocess
import sys
import tempfile
import threading
import time
import uuid
from typing import Dict, Iterator, List, Optional

from .ipc import TransportBusy, make_router_sockets, transport_kind


def visible_gpu_mapping(dp_size: int, env=None) -> List[int]:
    """GPU index of every rank: the first dp_size entries of a pre-set HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES mask
    (launch.py:190-207 respects the user's mask), else 0..dp_size-1."""
    env = os.environ if env is None else env
    mask = env.get("HIP_VISIBLE_DEVICES", env.get("CUDA_VISIBLE_DEVICES"))
    if mask is None:
        return list(range(dp_size))
    avail = [int(x) for x in (t.strip() for t in mask.split(",")) if x.isdigit()]
    if len(avail) < dp_size:
        raise ValueError(f"visible-device mask '{mask}' provides {len(avail)} GPUs, --dp-size={dp_size} needs {dp_size}")
    return avail[:dp_size]


def encode_request(request_id: str, prompt: str, is_streaming=True, model_kwargs=None, audio_path=None) -> bytes:
    """launch.py:522-531 (same bytes as scheduler.encode_request, which lives behind a torch import)."""
    return json.dumps({"request_id": request_id, "prompt": prompt, "audio_path": audio_path, "is_streaming": is_streaming,
                       "model_kwargs": model_kwargs or {}}).encode("utf-8") + b"|audio_data_placeholder"


class ServingPool:
    def __init__(self, model_name: str = "qwen3-tts", scheduler_type: str = "base", dp_size: int = 1,
                 request_socket_path: Optional[str] = None, result_socket_path: Optional[str] = None,
                 max_batch_size: int = 8, max_num_pages: Optional[int] = None, page_size: int = 128, top_p=None,
                 top_k=None, min_p=None, temperature=None, max_tokens=None, repetition_penalty=None,
                 repetition_window=None, cfg_scale=None, greedy: bool = False, enable_disaggregation: bool = False,
                 enable_nvtx: bool = False, async_scheduling: bool = False, detokenize_interval: Optional[int] = None,
                 synthetic: bool = False, checkpoint_dir: Optional[str] = None, worker_factory: Optional[str] = None,
                 log_level: str = "INFO", ready_timeout_s: float = 600.0, extra_env: Optional[Dict[str, str]] = None,
                 pin_devices: bool = True):
        self.model_name, self.scheduler_type, self.dp_size = model_name, scheduler_type, int(dp_size)
        if self.dp_size < 1:
            raise ValueError("dp_size must be >= 1")
        self._tmp = None
        if request_socket_path is None or result_socket_path is None:
            self._tmp = tempfile.mkdtemp(prefix="vox_pool_")
            request_socket_path = request_socket_path or os.path.join(self._tmp, "request.ipc")
            result_socket_path = result_socket_path or os.path.join(self._tmp, "result.ipc")
        self.request_socket_path, self.result_socket_path = request_socket_path, result_socket_path
        self.max_batch_size = max_batch_size
        self._daemon_args = dict(max_num_pages=max_num_pages, page_size=page_size, top_p=top_p, top_k=top_k, min_p=min_p,
                                 temperature=temperature, max_tokens=max_tokens, repetition_penalty=repetition_penalty,
                                 repetition_window=repetition_window, cfg_scale=cfg_scale,
                                 detokenize_interval=detokenize_interval, checkpoint_dir=checkpoint_dir,
                                 worker_factory=worker_factory)
        self._daemon_flags = dict(greedy=greedy, enable_disaggregation=enable_disaggregation, enable_nvtx=enable_nvtx,
                                  async_scheduling=async_scheduling, synthetic=synthetic)
        self.log_level, self.extra_env, self.pin_devices = log_level, dict(extra_env or {}), pin_devices
        self.logger = logging.getLogger(__name__)
        self.pending_requests: Dict[str, Dict] = {}       # request_id -> {chunks, event, rank, completion}
        self.recently_completed = collections.OrderedDict()
        self.recently_completed_ttl_sec = 5.0
        self.request_lock = threading.Lock()
        self.running = True
        self.dp_request_counter = 0                        # launch.py:131, :471-474
        self.ready: Dict[int, dict] = {}
        self._ready_event = threading.Event()
        self.scheduler_processes: List[subprocess.Popen] = []
        # ONE transport decision for both ends of the wire (ipc.transport_kind: ZeroMQ like the reference when pyzmq is importable
        # and VOX_TRANSPORT != "ipc"); the daemons get it in VOX_TRANSPORT, so they cannot pick a different framing than the router
        self.transport = transport_kind({**os.environ, **self.extra_env})
        # the result end binds BEFORE the daemons start (they connect to it and announce READY on it)
        self.result_socket, self.request_sockets = make_router_sockets(self.transport, self.request_socket_path,
                                                                       self.result_socket_path, self.dp_size)
        self.to_scheduler: "queue.Queue[bytes]" = queue.Queue(maxsize=max(1, self.max_batch_size * 2 * self.dp_size))
        # one bounded queue + sender thread per rank behind the router thread: a rank whose daemon is slow to drain its socket
        # holds back only the requests pinned to it (the reference's single sender loop retries in place and stalls every rank)
        self.rank_queues: List["queue.Queue[bytes]"] = [queue.Queue(maxsize=max(2, self.max_batch_size * 2))
                                                        for _ in range(self.dp_size)]
        self._last_reap = 0.0
        atexit.register(self.cleanup)
        try:
            self._start_schedulers()
            self.message_thread = threading.Thread(target=self._process_messages, name="vox-pool-results", daemon=True)
            self.message_thread.start()
            self.sender_thread = threading.Thread(target=self._sender_loop, name="vox-pool-sender", daemon=True)
            self.sender_thread.start()
            self.rank_threads = [threading.Thread(target=self._rank_sender_loop, args=(r,), name=f"vox-pool-sender-{r}", daemon=True)
                                 for r in range(self.dp_size)]
            for t in self.rank_threads:
                t.start()
            self._wait_ready(ready_timeout_s)
        except BaseException:
            self.cleanup()
            raise

    # ---- child processes ----
    def _daemon_cmd(self, rank: int) -> List[str]:
        cmd = [sys.executable, "-m", "vox_serve_amd.scheduler_entry", "--dp-rank", str(rank), "--dp-size", str(self.dp_size),
               "--model-name", self.model_name, "--scheduler-type", self.scheduler_type,
               "--max-batch-size", str(self.max_batch_size),
               "--request-socket-path", f"{self.request_socket_path}_{rank}",
               "--result-socket-path", self.result_socket_path, "--log-level", self.log_level]
        for k, v in self._daemon_args.items():
            if v is not None:
                cmd += ["--" + k.replace("_", "-"), str(v)]
        for k, on in self._daemon_flags.items():
            if on:
                cmd.append("--" + k.replace("_", "-"))
        return cmd

    def _start_schedulers(self):
        gpu_mapping = visible_gpu_mapping(self.dp_size) if self.pin_devices else [None] * self.dp_size
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        for rank in range(self.dp_size):
            env = os.environ.copy()
            env.update(self.extra_env)
            env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
            env["VOX_TRANSPORT"] = self.transport        # the router's decision, not the child's own probe of pyzmq
            if gpu_mapping[rank] is not None:
                # ONE mask: with both set, the CUDA_ one would be applied on top of the HIP_ one (an index into what is left)
                env["HIP_VISIBLE_DEVICES"] = str(gpu_mapping[rank])
                env.pop("CUDA_VISIBLE_DEVICES", None)
            # the daemons' stdout goes to this process's stderr: the parent's stdout may be a protocol (bench.py prints ONE JSON line)
            proc = subprocess.Popen(self._daemon_cmd(rank), env=env, stdout=2)
            self.scheduler_processes.append(proc)
            self.logger.info(f"started scheduler daemon rank {rank}/{self.dp_size} pid {proc.pid} on GPU {gpu_mapping[rank]}")

    def _wait_ready(self, timeout_s: float):
        t0 = time.time()
        while len(self.ready) < self.dp_size:
            dead = [(r, p.returncode) for r, p in enumerate(self.scheduler_processes) if p.poll() is not None]
            if dead:
                raise RuntimeError(f"scheduler daemon(s) exited during start-up: {dead}")
            if time.time() - t0 > timeout_s:
                raise TimeoutError(f"only ranks {sorted(self.ready)} of {self.dp_size} became ready in {timeout_s:.0f} s")
            self._ready_event.wait(0.05)
            self._ready_event.clear()

    def _stop_scheduler(self):
        for i, proc in enumerate(self.scheduler_processes):
            if proc.poll() is None:
                try:
                    proc.terminate()
                    try:
                        proc.wait(timeout=2)
                    except subprocess.TimeoutExpired:
                        self.logger.warning(f"scheduler {i} did not terminate, killing it")
                        proc.kill()
                        proc.wait(timeout=2)
                except Exception as e:
                    self.logger.error(f"error stopping scheduler {i}: {e}")

    def cleanup(self):
        if not self.running and not self.scheduler_processes:
            return
        self.running = False
        # The result / sender threads may be inside recv / send on these sockets (pyzmq sockets are not thread-safe, and closing the
        # PULL socket terminates the shared context, which blocks while any socket of it is open): let every loop see running = False
        # and leave — their polls are <= 0.1 s — BEFORE the sockets are closed from this thread.
        me = threading.current_thread()
        for t in [getattr(self, "message_thread", None), getattr(self, "sender_thread", None)] + list(getattr(self, "rank_threads", [])):
            if t is not None and t is not me and t.is_alive():
                t.join(timeout=2.0)
                if t.is_alive():
                    self.logger.warning(f"{t.name} did not stop in 2 s; closing its socket under it")
        self._stop_scheduler()
        self.scheduler_processes = []
        for s in self.request_sockets:
            s.close()
        self.result_socket.close()
        for r in range(self.dp_size):
            try:
                os.unlink(f"{self.request_socket_path}_{r}")
            except OSError:
                pass
        if self._tmp:
            try:
                os.rmdir(self._tmp)
            except OSError:
                pass
        try:
            atexit.unregister(self.cleanup)
        except Exception:
            pass

    __enter__ = lambda self: self

    def __exit__(self, *exc):
        self.cleanup()

    # ---- results: one shared transport, demultiplexed by request id ----
    def _process_messages(self):
        while self.running:
            try:
                message = self.result_socket.recv(0.05)
            except Exception as e:
                if self.running:
                    self.logger.error(f"result transport: {e}")
                continue
            # dead daemons are looked for on a timer, not only when the result socket is idle: under steady traffic from the healthy
            # ranks the socket never idles and the requests of a dead rank would wait for their clients' timeouts
            now = time.time()
            if message is None or now - self._last_reap > 0.25:
                self._last_reap = now
                self._reap_dead_daemons()
            if message is None:
                continue
            parts = message.split(b"|", 2)
            if len(parts) < 3:
                self.logger.warning(f"malformed message: {message[:100]!r}")
                continue
            request_id, message_type, data = parts[0].decode("utf-8"), parts[1].decode("utf-8"), parts[2]
            if message_type == "READY":
                info = json.loads(data.decode("utf-8"))
                self.ready[int(info["dp_rank"])] = info
                self._ready_event.set()
                continue
            with self.request_lock:
                now = time.time()
                while self.recently_completed and now - next(iter(self.recently_completed.values())) > self.recently_completed_ttl_sec:
                    self.recently_completed.popitem(last=False)
                entry = self.pending_requests.get(request_id)
                if entry is not None:
                    if message_type == "AUDIO":
                        if not entry["chunks"]:
                            entry["first_audio_time"] = now
                        entry["chunks"].append(data)
                    elif message_type == "COMPLETION":
                        entry["completion"] = json.loads(data.decode("utf-8"))
                        entry["done_time"] = now
                        self.recently_completed[request_id] = now
                    entry["event"].set()
                elif request_id not in self.recently_completed:
                    self.logger.warning(f"{message_type} for unknown request {request_id}")

    def _reap_dead_daemons(self):
        """A daemon that exited takes its requests with it (the reference's clients wait for their timeout): answer every request
        routed to that rank with an error COMPLETION, once."""
        if not self.ready or len(self.ready) < self.dp_size:
            return                                          # start-up failures are _wait_ready's to report
        dead = {r for r, p in enumerate(self.scheduler_processes) if p.poll() is not None}
        if not dead:
            return
        with self.request_lock:
            now = time.time()
            for rid, entry in self.pending_requests.items():
                if entry.get("rank") in dead and entry["completion"] is None:
                    code = self.scheduler_processes[entry["rank"]].returncode
                    entry["completion"] = {"status": "error", "reason": f"scheduler daemon of rank {entry['rank']} exited ({code})"}
                    entry["done_time"] = now
                    self.recently_completed[rid] = now
                    entry["event"].set()

    # ---- requests: bounded queue -> sender thread -> rank = counter % dp_size ----
    def _enqueue_request(self, payload: bytes, block: bool = False):
        try:
            self.to_scheduler.put(payload, block=block)
        except queue.Full:
            raise RuntimeError("server busy: request queue full") from None      # the reference answers HTTP 429

    def _sender_loop(self):
        while self.running:
            try:
                payload = self.to_scheduler.get(timeout=0.1)
            except queue.Empty:
                continue
            # the request is pinned to this rank even under back-pressure (launch.py:471-474)
            rank = self.dp_request_counter % self.dp_size
            self.dp_request_counter += 1
            rid = self._request_id_of(payload)
            with self.request_lock:
                if rid in self.pending_requests:
                    self.pending_requests[rid]["rank"] = rank
            while self.running:
                try:
                    self.rank_queues[rank].put(payload, timeout=0.1)
                    break
                except queue.Full:
                    if rank < len(self.scheduler_processes) and self.scheduler_processes[rank].poll() is not None:
                        break                              # its daemon is gone: _reap_dead_daemons answers the request

    def _rank_sender_loop(self, rank: int):
        backoff_initial, backoff_max = 0.001, 0.02
        while self.running:
            try:
                payload = self.rank_queues[rank].get(timeout=0.1)
            except queue.Empty:
                self.request_sockets[rank].flush()          # finish a frame the socket took only part of
                continue
            backoff = backoff_initial
            while self.running:
                try:
                    self.request_sockets[rank].send(payload)      # DONTWAIT: TransportBusy = back-pressure (zmq.Again of launch.py:484)
                    break
                except TransportBusy:
                    if rank < len(self.scheduler_processes) and self.scheduler_processes[rank].poll() is not None:
                        break                              # its daemon is gone: _reap_dead_daemons answers the request
                    time.sleep(backoff)
                    backoff = min(backoff * 2, backoff_max)
                except Exception as e:
                    self.logger.error(f"sender loop (rank {rank}): {e}")
                    break

    @staticmethod
    def _request_id_of(payload: bytes) -> Optional[str]:
        try:
            return json.loads(payload[:payload.rfind(b"|")].decode("utf-8")).get("request_id")
        except Exception:
            return None

    # ---- client surface ----
    def start_streaming_request(self, text: str = None, audio_path: str = None, model_kwargs: Dict = None,
                                request_id: Optional[str] = None, is_streaming: bool = True, block: bool = False) -> str:
        request_id = request_id or str(uuid.uuid4())
        with self.request_lock:
            self.pending_requests[request_id] = {"chunks": [], "event": threading.Event(), "streaming": is_streaming,
                                                 "consumed_chunks": 0, "rank": None, "completion": None,
                                                 "submit_time": time.time()}
        try:
            self._enqueue_request(encode_request(request_id, text, is_streaming=is_streaming, model_kwargs=model_kwargs,
                                                 audio_path=audio_path), block=block)
        except Exception:
            with self.request_lock:
                self.pending_requests.pop(request_id, None)
            raise
        return request_id

    def stream(self, request_id: str, timeout_s: float = 120.0) -> Iterator[bytes]:
        """PCM16 chunks of one request in arrival order; returns after its COMPLETION.  `completion(request_id)` holds
        the final status afterwards."""
        deadline = time.time() + timeout_s
        while True:
            with self.request_lock:
                entry = self.pending_requests[request_id]
                new = entry["chunks"][entry["consumed_chunks"]:]
                entry["consumed_chunks"] += len(new)
                done = entry["completion"] is not None
                entry["event"].clear()
            for c in new:
                yield c
            if done:
                return
            if not entry["event"].wait(max(0.0, min(1.0, deadline - time.time()))) and time.time() >= deadline:
                raise TimeoutError(f"request {request_id} timed out after {timeout_s:.0f} s")

    def generate(self, text: str, model_kwargs: Dict = None, timeout_s: float = 120.0, **kw) -> bytes:
        """One request, start to finish: its PCM16 bytes.  A request that ended with an error COMPLETION (a failed preprocess, a
        scheduler daemon that exited: `_reap_dead_daemons`) raises with the reason instead of handing back truncated audio — the
        reference answers such a request with an HTTP error (launch.py:593-640)."""
        rid = self.start_streaming_request(text, model_kwargs=model_kwargs, **kw)
        try:
            pcm = b"".join(self.stream(rid, timeout_s))
            done = self.completion(rid) or {}
            if done.get("status") == "error":
                raise RuntimeError(f"request {rid} failed: {done.get('reason')}" + (f" ({done['detail']})" if done.get("detail") else ""))
            return pcm
        finally:
            self.release(rid)             # (launch.py:593-599: the entry goes away with the response; a long-lived pool must not keep every request's PCM)

    def completion(self, request_id: str) -> Optional[dict]:
        with self.request_lock:
            e = self.pending_requests.get(request_id)
            return None if e is None else e["completion"]

    def request_info(self, request_id: str) -> dict:
        """rank the router sent the request to, and its client-side timings (submit / first audio / done)."""
        with self.request_lock:
            e = self.pending_requests[request_id]
            return {k: e.get(k) for k in ("rank", "submit_time", "first_audio_time", "done_time", "completion")}

    def release(self, request_id: str):
        with self.request_lock:
            self.pending_requests.pop(request_id, None)
