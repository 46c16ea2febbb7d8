"""PUSH / PULL message sockets over AF_UNIX for the scheduler daemons' wire (`ipc://` paths of the reference:
/root/reference/vox_serve/scheduler/base.py:103-125 on the scheduler side, launch.py:141-160 on the server side).

The reference uses ZeroMQ PUSH/PULL; pyzmq is not a dependency here, so the same roles are provided by length-prefixed
frames (little-endian uint32 + payload) on stream sockets: a PULL end binds a path and fair-drains any number of connected
PUSH ends, a PUSH end connects lazily (a peer that is not listening yet is "busy", like a full ZeroMQ pipe under
DONTWAIT) and keeps message boundaries.  `ZmqTransport` (scheduler/base.py) is used instead when pyzmq is importable and
VOX_TRANSPORT is not "ipc"; the messages on the wire are the same bytes either way.
"""
import collections
import os
import selectors
import socket
import struct
import threading
import time
from typing import Optional

_HDR = struct.Struct("<I")
MAX_FRAME = 256 << 20


class TransportBusy(Exception):
    """The peer is not accepting right now (not listening yet / send buffer full): retry later, same target."""


class PullSocket:
    def __init__(self, path: str, backlog: int = 64):
        self.path = path
        try:
            os.unlink(path)
        except FileNotFoundError:
            pass
        self._listener = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        self._listener.bind(path)
        self._listener.listen(backlog)
        self._listener.setblocking(False)
        self._sel = selectors.DefaultSelector()
        self._sel.register(self._listener, selectors.EVENT_READ)
        self._bufs = {}
        self._ready = collections.deque()
        self._closed = False

    def _drain(self, conn):
        try:
            data = conn.recv(1 << 20)
        except (BlockingIOError, InterruptedError):
            return
        except OSError:
            data = b""
        if not data:
            self._sel.unregister(conn)
            conn.close()
            self._bufs.pop(conn, None)
            return
        buf = self._bufs[conn]
        buf += data
        while len(buf) >= 4:
            (n,) = _HDR.unpack_from(buf, 0)
            if n > MAX_FRAME:                       # not one of ours: drop the connection
                self._sel.unregister(conn)
                conn.close()
                self._bufs.pop(conn, None)
                return
            if len(buf) < 4 + n:
                break
            self._ready.append(bytes(buf[4:4 + n]))
            del buf[:4 + n]

    def recv(self, timeout: float = 0.0) -> Optional[bytes]:
        """One message, or None when nothing arrives within `timeout` seconds (0 = poll, the NOBLOCK of the reference)."""
        if self._ready:
            return self._ready.popleft()
        if self._closed:
            return None
        for key, _ in self._sel.select(timeout):
            if key.fileobj is self._listener:
                while True:
                    try:
                        conn, _addr = self._listener.accept()
                    except (BlockingIOError, InterruptedError):
                        break
                    conn.setblocking(False)
                    self._bufs[conn] = bytearray()
                    self._sel.register(conn, selectors.EVENT_READ)
            else:
                self._drain(key.fileobj)
        return self._ready.popleft() if self._ready else None

    def pending(self) -> bool:
        return bool(self._ready)

    def close(self):
        if self._closed:
            return
        self._closed = True
        for conn in list(self._bufs):
            try:
                self._sel.unregister(conn)
                conn.close()
            except Exception:
                pass
        self._bufs.clear()
        try:
            self._sel.unregister(self._listener)
        except Exception:
            pass
        self._listener.close()
        self._sel.close()
        try:
            os.unlink(self.path)
        except OSError:
            pass


class PushSocket:
    def __init__(self, path: str):
        self.path = path
        self._sock = None
        self._tail = b""
        self._lock = threading.Lock()

    def _connect(self):
        s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        try:
            s.connect(self.path)
        except (FileNotFoundError, ConnectionRefusedError) as e:
            s.close()
            raise TransportBusy(str(e)) from None
        s.setblocking(False)
        self._sock = s

    def send(self, payload: bytes, wait: Optional[float] = None):
        """Send one message.  wait=None: raise TransportBusy at once when the peer is not listening (DONTWAIT);
        wait=seconds: keep trying to connect for that long (the scheduler side, whose server binds first)."""
        deadline = None if wait is None else time.time() + wait
        with self._lock:
            while self._sock is None:
                try:
                    self._connect()
                except TransportBusy:
                    if deadline is None or time.time() >= deadline:
                        raise
                    time.sleep(0.005)
            # Non-blocking (DONTWAIT): a stream socket may take only part of a frame, so the unsent remainder of an ACCEPTED frame is
            # kept in `_tail` (at most one frame) and goes out first on the next send() / flush().  While a remainder is still
            # stuck after `wait` (None: 50 ms), the new payload is NOT accepted: TransportBusy asks the caller for it again — one
            # daemon that stops draining its socket cannot block the router's sender thread behind a sendall().
            stall = time.time() + (0.05 if wait is None else max(wait, 0.05))
            if not self._flush_locked(stall):
                raise TransportBusy("peer's receive buffer is full")
            self._tail = _HDR.pack(len(payload)) + payload
            self._flush_locked(stall)

    def _flush_locked(self, deadline: float) -> bool:
        """Push `_tail` into the socket until it is empty (True) or `deadline` passes (False); a dead peer raises TransportBusy."""
        if not self._tail:
            return True
        if self._sock is None:
            self._tail = b""
            return True
        view = memoryview(self._tail)
        try:
            while len(view):
                try:
                    view = view[self._sock.send(view):]
                except (BlockingIOError, InterruptedError):
                    if time.time() >= deadline:
                        break
                    time.sleep(0.0005)
        except (BrokenPipeError, ConnectionResetError, OSError) as e:
            try:
                self._sock.close()
            finally:
                self._sock, self._tail = None, b""
            raise TransportBusy(f"peer went away: {e}") from None
        self._tail = bytes(view)
        return not self._tail

    def flush(self, wait: float = 0.0) -> bool:
        """Try to finish a partly sent frame (idle tick of the sender thread); True when nothing is pending."""
        with self._lock:
            try:
                return self._flush_locked(time.time() + wait)
            except TransportBusy:
                return True

    def close(self):
        with self._lock:
            if self._sock is not None:
                try:
                    self._flush_locked(time.time() + 0.5)
                except TransportBusy:
                    pass
            if self._sock is not None:
                self._sock.close()
                self._sock = None


class IpcTransport:
    """Scheduler-side transport (same interface as QueueTransport / ZmqTransport): PULL requests on
    `request_socket_path` (bound here), PUSH results to `result_socket_path` (bound by the server)."""

    def __init__(self, request_socket_path="/tmp/vox_serve_request.ipc", result_socket_path="/tmp/vox_serve_result.ipc"):
        self.request_socket = PullSocket(request_socket_path)
        self.result_socket = PushSocket(result_socket_path)
        self.on_send = None

    def recv_request(self) -> Optional[bytes]:
        return self.request_socket.recv(0.0)

    def send_result(self, payload: bytes):
        self.result_socket.send(payload, wait=30.0)
        if self.on_send is not None:
            self.on_send(payload)

    def pending(self) -> bool:
        return self.request_socket.pending()

    def close(self):
        self.request_socket.close()
        self.result_socket.close()


def transport_kind(env=None) -> str:
    """ONE rule for both ends of the wire: "zmq" (the reference's PUSH/PULL) when pyzmq is importable and VOX_TRANSPORT is not
    "ipc"; "ipc" (the AF_UNIX framing above) otherwise.  VOX_TRANSPORT=zmq insists on ZeroMQ (ImportError without pyzmq).
    The serving pool decides once and hands its decision to the daemons in VOX_TRANSPORT, so that router and schedulers can
    never speak different framings."""
    env = os.environ if env is None else env
    want = env.get("VOX_TRANSPORT", "")
    if want == "ipc":
        return "ipc"
    try:
        import zmq  # noqa: F401
        return "zmq"
    except ImportError:
        if want == "zmq":
            raise
        return "ipc"


def make_transport(request_socket_path: str, result_socket_path: str):
    """Scheduler side: ZeroMQ like the reference or the AF_UNIX sockets, by transport_kind()."""
    if transport_kind() == "zmq":
        from .scheduler.base import ZmqTransport
        return ZmqTransport(request_socket_path, result_socket_path)
    return IpcTransport(request_socket_path, result_socket_path)


class ZmqPullSocket:
    """Router-side result end over ZeroMQ (launch.py:141-160 of the reference: PULL bound on ipc://<path>, RCVHWM 1024,
    LINGER 0) with PullSocket's interface."""

    def __init__(self, path: str, context=None):
        import zmq
        self._zmq = zmq
        self._own = context is None
        self._ctx = context or zmq.Context()
        self.path = path
        self._sock = self._ctx.socket(zmq.PULL)
        self._sock.setsockopt(zmq.RCVHWM, 1024)
        self._sock.setsockopt(zmq.LINGER, 0)
        self._sock.bind(f"ipc://{path}")

    def recv(self, timeout: float = 0.0) -> Optional[bytes]:
        if self._sock.poll(int(timeout * 1000)):
            try:
                return self._sock.recv(flags=self._zmq.NOBLOCK)
            except self._zmq.Again:
                return None
        return None

    def pending(self) -> bool:
        return bool(self._sock.poll(0))

    def close(self):
        try:
            self._sock.close(0)
        finally:
            if self._own:
                self._ctx.term()
            try:
                os.unlink(self.path)
            except OSError:
                pass


class ZmqPushSocket:
    """Router-side request end over ZeroMQ (PUSH connected to ipc://<path>_<rank>, SNDHWM 256, LINGER 0, DONTWAIT sends:
    zmq.Again -> TransportBusy, the back-pressure signal of launch.py:476-495) with PushSocket's interface."""

    def __init__(self, path: str, context=None):
        import zmq
        self._zmq = zmq
        self._own = context is None
        self._ctx = context or zmq.Context()
        self.path = path
        self._sock = self._ctx.socket(zmq.PUSH)
        self._sock.setsockopt(zmq.SNDHWM, 256)
        self._sock.setsockopt(zmq.LINGER, 0)
        self._sock.setsockopt(zmq.IMMEDIATE, 1)      # queue only to a completed connection: a daemon that is not up yet is "busy"
        self._sock.connect(f"ipc://{path}")
        self._lock = threading.Lock()

    def send(self, payload: bytes, wait: Optional[float] = None):
        deadline = None if wait is None else time.time() + wait
        with self._lock:
            while True:
                try:
                    self._sock.send(payload, flags=self._zmq.NOBLOCK)
                    return
                except self._zmq.Again:
                    if deadline is None or time.time() >= deadline:
                        raise TransportBusy("zmq pipe full / peer not connected") from None
                    time.sleep(0.005)

    def flush(self, wait: float = 0.0) -> bool:
        return True

    def close(self):
        with self._lock:
            self._sock.close(0)
            if self._own:
                self._ctx.term()


def make_router_sockets(kind: str, request_socket_path: str, result_socket_path: str, dp_size: int):
    """(result PULL end, [request PUSH end per rank]) of the serving pool for transport `kind` ("zmq" | "ipc")."""
    if kind == "zmq":
        import zmq
        ctx = zmq.Context()
        pull = ZmqPullSocket(result_socket_path, ctx)
        pushes = [ZmqPushSocket(f"{request_socket_path}_{r}", ctx) for r in range(dp_size)]
        pull._own = True                       # the PULL end is closed last and terminates the shared context
        return pull, pushes
    return PullSocket(result_socket_path), [PushSocket(f"{request_socket_path}_{r}") for r in range(dp_size)]
