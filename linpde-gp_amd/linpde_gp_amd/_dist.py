"""Process-group plumbing for one-process-per-GPU runs without importing torch.

Ranks are launched by `python -m torch.distributed.run` (env RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT).  The control plane (barrier, max-reduce of timings, exchange
of small blobs such as an RCCL unique id) is a TCP star rooted at rank 0 on
MASTER_PORT + 1 + LPGP_PORT_OFFSET.  The data plane never goes through here.
"""

from __future__ import annotations

import os
import pickle
import socket
import struct
import time


def _send(sock, obj):
    blob = pickle.dumps(obj)
    sock.sendall(struct.pack("!Q", len(blob)) + blob)


def _recv(sock):
    hdr = b""
    while len(hdr) < 8:
        chunk = sock.recv(8 - len(hdr))
        if not chunk:
            raise ConnectionError("peer closed")
        hdr += chunk
    (n,) = struct.unpack("!Q", hdr)
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(1 << 20, n - len(buf)))
        if not chunk:
            raise ConnectionError("peer closed")
        buf += chunk
    return pickle.loads(bytes(buf))


class Comm:
    NPORTS = 8

    def __init__(self, rank: int, world: int, addr: str = "127.0.0.1", port: int = 29601):
        self.rank, self.world = rank, world
        self._peers = []
        self._sock = None
        if world == 1:
            return
        # Rendezvous: rank 0 listens on the first free port of [port, port + NPORTS); a client walks
        # the same list until a connection answers the handshake (so a foreign listener on one of
        # the ports, or a socket of a previous run still in TIME_WAIT, does not break the job).
        ports = [port + i for i in range(self.NPORTS)]
        magic = ("lpgp-comm", world)
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
                    hello = _recv(c)
                    if not (isinstance(hello, tuple) and len(hello) == 2 and hello[0] == magic):
                        raise ConnectionError("not a peer")
                    _send(c, magic)
                    c.settimeout(None)
                    peers[int(hello[1])] = c
                except (OSError, ConnectionError, pickle.UnpicklingError, EOFError):
                    c.close()
            self._peers = [peers[r] for r in range(1, world)]
            srv.close()
        else:
            deadline = time.time() + 300.0
            s, i = None, 0
            while s is None:
                prt = ports[i % len(ports)]
                i += 1
                try:
                    cand = socket.create_connection((addr, prt), timeout=5.0)
                    cand.settimeout(10.0)
                    _send(cand, (magic, rank))
                    if _recv(cand) != magic:
                        raise ConnectionError("not rank 0")
                    s = cand
                except (OSError, ConnectionError, pickle.UnpicklingError, EOFError):
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

    def gather(self, obj):
        """rank 0 gets the list of all ranks' objects, others get None."""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            return [obj] + [_recv(p) for p in self._peers]
        _send(self._sock, obj)
        return None

    def bcast(self, obj):
        if self.world == 1:
            return obj
        if self.rank == 0:
            for p in self._peers:
                _send(p, obj)
            return obj
        return _recv(self._sock)

    def bcast_from(self, obj, root: int):
        """Broadcast from an arbitrary rank over the star: the root hands its object to rank 0,
        which forwards it to everybody (the root included)."""
        if self.world == 1 or root == 0:
            return self.bcast(obj)
        if self.rank == 0:
            obj = _recv(self._peers[root - 1])
            for p in self._peers:
                _send(p, obj)
            return obj
        if self.rank == root:
            _send(self._sock, obj)
        return _recv(self._sock)

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
