"""Process-group plumbing for one-process-per-GPU runs without importing torch.

Ranks are launched by `python -m torch.distributed.run` (env RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT).  The control plane (barrier, max-reduce of timings, exchange
of small blobs such as an RCCL unique id, gathered prediction results) is a TCP star rooted
at rank 0 on MASTER_PORT + 1 + LPGP_PORT_OFFSET.  The data plane (panels of the factor) never
goes through here, except for the bring-up / test transport (`Context.dist_init(transport="host")`).

Wire format -- NOT pickle: nothing received from a socket is ever executed.  A message is a frame
`!Q length | payload | 32-byte HMAC-SHA256(key, payload)`; the payload is a tagged tree of None / bool /
int / float / str / bytes / float64-or-int64 ndarray / list / tuple / dict (`_encode` / `_decode`).  Before a
connection counts as a peer both sides prove knowledge of the job key in a challenge-response
(fresh 16-byte nonces), and until then only tiny frames are accepted.  The key is
$LPGP_COMM_SECRET if set (give the job a random one when the port range is reachable by others),
otherwise it is derived from the launcher's environment (MASTER_ADDR, MASTER_PORT, WORLD_SIZE,
TORCHELASTIC_RUN_ID) -- enough to reject a stray or stale connection, not a secret.
"""

from __future__ import annotations

import hashlib
import hmac
import os
import socket
import struct
import time

import numpy as np

MAX_FRAME = int(os.environ.get("LPGP_COMM_MAX_FRAME", str(1 << 31)))   # bytes; the host test transport ships whole panels
MAX_HANDSHAKE_FRAME = 256
_DEPTH_LIMIT = 16


# ---- typed binary encoding -------------------------------------------------------------------
def _encode(obj, out: bytearray, depth: int = 0) -> None:
    if depth > _DEPTH_LIMIT:
        raise ValueError("control plane: message nested too deeply")
    if obj is None:
        out += b"N"
    elif isinstance(obj, (bool, np.bool_)):
        out += b"T" if obj else b"F"
    elif isinstance(obj, (int, np.integer)):
        out += b"i" + struct.pack("!q", int(obj))
    elif isinstance(obj, (float, np.floating)):
        out += b"d" + struct.pack("!d", float(obj))
    elif isinstance(obj, str):
        raw = obj.encode("utf-8")
        out += b"s" + struct.pack("!Q", len(raw)) + raw
    elif isinstance(obj, (bytes, bytearray, memoryview)):
        raw = bytes(obj)
        out += b"b" + struct.pack("!Q", len(raw)) + raw
    elif isinstance(obj, np.ndarray):
        if obj.dtype == np.float64:
            code = b"f"
        elif obj.dtype == np.int64:
            code = b"q"
        else:
            raise TypeError(f"control plane: ndarray dtype {obj.dtype} is not carried (float64 / int64 only)")
        arr = np.ascontiguousarray(obj)
        out += b"a" + code + struct.pack("!B", arr.ndim) + struct.pack(f"!{arr.ndim}Q", *arr.shape) + arr.tobytes()
    elif isinstance(obj, (list, tuple)):
        out += (b"l" if isinstance(obj, list) else b"t") + struct.pack("!Q", len(obj))
        for item in obj:
            _encode(item, out, depth + 1)
    elif isinstance(obj, dict):
        out += b"m" + struct.pack("!Q", len(obj))
        for key, item in obj.items():
            if not isinstance(key, (str, int)):
                raise TypeError("control plane: dict keys must be str or int")
            _encode(key, out, depth + 1)
            _encode(item, out, depth + 1)
    else:
        raise TypeError(f"control plane: objects of type {type(obj).__name__} are not carried")


def _decode(buf: memoryview, pos: int = 0, depth: int = 0):
    if depth > _DEPTH_LIMIT:
        raise ValueError("control plane: message nested too deeply")

    def take(n):
        nonlocal pos
        if n < 0 or pos + n > len(buf):
            raise ValueError("control plane: truncated message")
        piece = buf[pos:pos + n]
        pos += n
        return piece

    tag = bytes(take(1))
    if tag == b"N":
        return None, pos
    if tag == b"T":
        return True, pos
    if tag == b"F":
        return False, pos
    if tag == b"i":
        return struct.unpack("!q", take(8))[0], pos
    if tag == b"d":
        return struct.unpack("!d", take(8))[0], pos
    if tag in (b"s", b"b"):
        (n,) = struct.unpack("!Q", take(8))
        raw = bytes(take(n))
        return (raw.decode("utf-8") if tag == b"s" else raw), pos
    if tag == b"a":
        code = bytes(take(1))
        if code not in (b"f", b"q"):
            raise ValueError("control plane: unknown array dtype")
        (ndim,) = struct.unpack("!B", take(1))
        if ndim > 8:
            raise ValueError("control plane: array rank")
        shape = struct.unpack(f"!{ndim}Q", take(8 * ndim)) if ndim else ()
        count = 1
        for s in shape:
            count *= s
        raw = take(8 * count)
        arr = np.frombuffer(raw, dtype=np.float64 if code == b"f" else np.int64).reshape(shape).copy()
        return arr, pos
    if tag in (b"l", b"t"):
        (n,) = struct.unpack("!Q", take(8))
        if n > len(buf):
            raise ValueError("control plane: bad container length")
        items = []
        for _ in range(n):
            item, pos = _decode(buf, pos, depth + 1)
            items.append(item)
        return (items if tag == b"l" else tuple(items)), pos
    if tag == b"m":
        (n,) = struct.unpack("!Q", take(8))
        if n > len(buf):
            raise ValueError("control plane: bad container length")
        d = {}
        for _ in range(n):
            key, pos = _decode(buf, pos, depth + 1)
            if not isinstance(key, (str, int)) or isinstance(key, bool):
                raise ValueError("control plane: bad dict key")
            d[key], pos = _decode(buf, pos, depth + 1)
        return d, pos
    raise ValueError(f"control plane: unknown tag {tag!r}")


def dumps(obj) -> bytes:
    out = bytearray()
    _encode(obj, out)
    return bytes(out)


def loads(blob) -> object:
    try:
        obj, pos = _decode(memoryview(blob))
    except (struct.error, OverflowError, MemoryError, RecursionError) as exc:     # malformed input never escapes as anything else
        raise ValueError(f"control plane: malformed message ({type(exc).__name__})") from None
    if pos != len(blob):
        raise ValueError("control plane: trailing bytes")
    return obj


# ---- framing + authentication ------------------------------------------------------------------
def job_key() -> bytes:
    secret = os.environ.get("LPGP_COMM_SECRET")
    if secret:
        return hashlib.sha256(b"lpgp-comm-secret|" + secret.encode()).digest()
    parts = [os.environ.get(k, "") for k in ("MASTER_ADDR", "MASTER_PORT", "WORLD_SIZE", "TORCHELASTIC_RUN_ID")]
    return hashlib.sha256(("lpgp-comm-env|" + "|".join(parts)).encode()).digest()


def _recv_exact(sock, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(1 << 20, n - len(buf)))
        if not chunk:
            raise ConnectionError("peer closed")
        buf += chunk
    return bytes(buf)


def _send_frame(sock, key: bytes, payload: bytes) -> None:
    tag = hmac.new(key, payload, hashlib.sha256).digest()
    sock.sendall(struct.pack("!Q", len(payload)) + payload + tag)


def _recv_frame(sock, key: bytes, limit: int) -> bytes:
    (n,) = struct.unpack("!Q", _recv_exact(sock, 8))
    if n > limit:
        raise ConnectionError(f"control plane: frame of {n} bytes exceeds the limit of {limit}")
    payload = _recv_exact(sock, n)
    tag = _recv_exact(sock, 32)
    if not hmac.compare_digest(tag, hmac.new(key, payload, hashlib.sha256).digest()):
        raise ConnectionError("control plane: message authentication failed")
    return payload


def _send(sock, key, obj):
    _send_frame(sock, key, dumps(obj))


def _recv(sock, key):
    return loads(_recv_frame(sock, key, MAX_FRAME))


_HELLO = b"lpgp-comm-1"


def _handshake_server(c, key: bytes, world: int) -> int:
    """Returns the authenticated peer's rank.  Only fixed-size tiny frames until the peer has proven
    knowledge of the key over a fresh nonce."""
    nonce = os.urandom(16)
    c.sendall(_HELLO + nonce)
    resp = _recv_exact(c, 8 + 16 + 32)                       # rank | client nonce | HMAC(key, hello|nonce|rank|cnonce)
    (rank,) = struct.unpack("!q", resp[:8])
    cnonce = resp[8:24]
    want = hmac.new(key, _HELLO + nonce + resp[:24] + struct.pack("!q", world), hashlib.sha256).digest()
    if not hmac.compare_digest(resp[24:], want) or not 1 <= rank < world:
        raise ConnectionError("not a peer of this job")
    c.sendall(hmac.new(key, b"srv" + cnonce + nonce, hashlib.sha256).digest())
    return int(rank)


def _handshake_client(s, key: bytes, rank: int, world: int) -> None:
    head = _recv_exact(s, len(_HELLO) + 16)
    if head[:len(_HELLO)] != _HELLO:
        raise ConnectionError("not rank 0 of an lpgp job")
    nonce = head[len(_HELLO):]
    cnonce = os.urandom(16)
    body = struct.pack("!q", rank) + cnonce
    s.sendall(body + hmac.new(key, _HELLO + nonce + body + struct.pack("!q", world), hashlib.sha256).digest())
    proof = _recv_exact(s, 32)
    if not hmac.compare_digest(proof, hmac.new(key, b"srv" + cnonce + nonce, hashlib.sha256).digest()):
        raise ConnectionError("rank 0 failed to authenticate")


class Comm:
    NPORTS = 8

    def __init__(self, rank: int, world: int, addr: str = "127.0.0.1", port: int = 29601, key: bytes | None = None):
        self.rank, self.world = rank, world
        self._peers = []
        self._sock = None
        self._key = key if key is not None else job_key()
        self.bytes_sent = 0
        if world == 1:
            return
        # Rendezvous: rank 0 listens on the first free port of [port, port + NPORTS); a client walks
        # the same list until a connection completes the handshake (so a foreign listener on one of
        # the ports, or a socket of a previous run still in TIME_WAIT, does not break the job).
        ports = [port + i for i in range(self.NPORTS)]
        if rank == 0:
            srv = None
            for prt in ports:
                cand = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                cand.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    cand.bind((addr, prt))
                    cand.listen(world)
                    srv = cand
                    break
                except OSError:
                    cand.close()
            if srv is None:
                raise OSError(f"no free control-plane port in {ports[0]}..{ports[-1]} on {addr}")
            srv.settimeout(300.0)
            peers = {}
            while len(peers) < world - 1:
                c, _ = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                try:
                    c.settimeout(10.0)
                    r = _handshake_server(c, self._key, world)
                    if r in peers:
                        raise ConnectionError("duplicate rank")
                    c.settimeout(None)
                    peers[r] = c
                except (OSError, ConnectionError, struct.error):
                    c.close()
            self._peers = [peers[r] for r in range(1, world)]
            srv.close()
        else:
            deadline = time.time() + 300.0
            s, i = None, 0
            while s is None:
                prt = ports[i % len(ports)]
                i += 1
                cand = None
                try:
                    cand = socket.create_connection((addr, prt), timeout=5.0)
                    cand.settimeout(10.0)
                    _handshake_client(cand, self._key, rank, world)
                    s = cand
                except (OSError, ConnectionError, struct.error):
                    if cand is not None:
                        cand.close()
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            s.settimeout(None)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            self._sock = s

    @classmethod
    def from_env(cls) -> "Comm":
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(os.environ.get("MASTER_PORT", "29600")) + 1 + int(os.environ.get("LPGP_PORT_OFFSET", "0"))
        return cls(rank, world, addr, port)

    def _tx(self, sock, obj):
        blob = dumps(obj)
        self.bytes_sent += len(blob)
        _send_frame(sock, self._key, blob)

    def _rx(self, sock):
        return _recv(sock, self._key)

    def gather(self, obj):
        """rank 0 gets the list of all ranks' objects, others get None."""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            return [obj] + [self._rx(p) for p in self._peers]
        self._tx(self._sock, obj)
        return None

    def bcast(self, obj):
        if self.world == 1:
            return obj
        if self.rank == 0:
            for p in self._peers:
                self._tx(p, obj)
            return obj
        return self._rx(self._sock)

    def bcast_from(self, obj, root: int):
        """Broadcast from an arbitrary rank over the star: the root hands its object to rank 0,
        which forwards it to everybody (the root included)."""
        if self.world == 1 or root == 0:
            return self.bcast(obj)
        if self.rank == 0:
            obj = self._rx(self._peers[root - 1])
            for p in self._peers:
                self._tx(p, obj)
            return obj
        if self.rank == root:
            self._tx(self._sock, obj)
        return self._rx(self._sock)

    def exchange(self, outgoing: dict) -> dict:
        """Point-to-point messages over the star (bring-up / test transport of the 2-D factorisation):
        `outgoing` maps destination rank -> bytes; returns source rank -> bytes for everything addressed
        to this rank.  Collective: every rank calls it, possibly with an empty dict."""
        if self.world == 1:
            return {self.rank: outgoing[self.rank]} if self.rank in outgoing else {}
        mine = [[int(dst), bytes(buf)] for dst, buf in outgoing.items()]
        everything = self.gather(mine)
        if self.rank == 0:
            inbox = [[] for _ in range(self.world)]
            for src, msgs in enumerate(everything):
                for dst, buf in msgs:
                    inbox[dst].append([src, buf])
            for r in range(1, self.world):
                self._tx(self._peers[r - 1], inbox[r])
            got = inbox[0]
        else:
            got = self._rx(self._sock)
        return {int(src): buf for src, buf in got}

    def allgather(self, obj):
        return self.bcast(self.gather(obj))

    def barrier(self):
        self.allgather(None)

    def allreduce_max(self, x: float) -> float:
        return max(self.allgather(float(x)))

    def close(self):
        for p in self._peers:
            p.close()
        if self._sock is not None:
            self._sock.close()
        self._peers, self._sock = [], None
