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
        self._lock = threading.Lock()

    def _connect(self):
        s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        try:
            s.connect(self.path)
        except (FileNotFoundError, ConnectionRefusedError) as e:
            s.close()
            raise TransportBusy(str(e)) from None
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
            try:
                self._sock.sendall(_HDR.pack(len(payload)) + payload)
            except (BrokenPipeError, ConnectionResetError, OSError) as e:
                try:
                    self._sock.close()
                finally:
                    self._sock = None
                raise TransportBusy(f"peer went away: {e}") from None

    def close(self):
        with self._lock:
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


def make_transport(request_socket_path: str, result_socket_path: str):
    """ZeroMQ like the reference when pyzmq is present (and VOX_TRANSPORT != "ipc"), the AF_UNIX sockets otherwise."""
    if os.environ.get("VOX_TRANSPORT", "") != "ipc":
        try:
            import zmq  # noqa: F401
            from .scheduler.base import ZmqTransport
            return ZmqTransport(request_socket_path, result_socket_path)
        except ImportError:
            pass
    return IpcTransport(request_socket_path, result_socket_path)
