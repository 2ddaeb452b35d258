"""Single-process front of the multi-GPU path.

The reference is called from ONE Python process (`prior.condition_on_observations(...)`, `_conditional.py:253-294`); the
multi-GPU path of this library is one process per GPU, every rank making every collective call (`include/lpgp.h`).
`spawn(n_gpus)` closes that gap without asking the user to launch the script SPMD: it starts `n_gpus` FRESH worker
processes (`python -m linpde_gp_amd._spawn_worker`: a new interpreter each that runs THAT module, never the caller's
script -- an unguarded script, one without `if __name__ == "__main__":`, works -- started before this process has made
any GPU call, never a fork or re-exec of a process that has touched the GPU), every worker opens its GPU, joins the job
(`Context.dist_init`: RCCL over xGMI by default) and then replays the calls this process forwards to it.  From then on

    u = prior.condition_on_observations(Y, X, L=D, b=noise)      # returns a proxy; the factor lives sharded on the GPUs
    mean, var = u.predict(x);  u.mean(x);  u.std(x);  u(x);  u.cov.matrix(x);  u.representer_weights;
    u2 = u.condition_on_observations(...)

run on all GPUs while the calling script stays what it was.  What travels: the prior, operators, noise and point arrays
(pickled over an authenticated `multiprocessing.connection` on a private AF_UNIX socket in a 0700 temporary directory --
not a network socket), results from rank 0.  The parent itself never opens a GPU.

    import linpde_gp_amd as lp
    lp.spawn(8)            # or: LPGP_SPAWN=8 in the environment, picked up at the first conditioning
"""

from __future__ import annotations

import atexit
import itertools
import multiprocessing
import os
import secrets
import shutil
import select
import socket
import subprocess
import sys
import tempfile
import time
import traceback
from multiprocessing.connection import Listener

import numpy as np

_active = None          # the process-wide WorkerGroup, if any
_in_worker = False      # set inside worker processes: conditioning there is the real thing, not a forward


def _free_port() -> int:
    # two consecutive free ports: the control plane of `_dist.Comm` listens on MASTER_PORT + 1
    for _ in range(64):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        if port + 1 < 65536:
            try:
                with socket.socket() as s2:
                    s2.bind(("127.0.0.1", port + 1))
                return port
            except OSError:
                continue
    raise RuntimeError("no free port pair for the control plane")


def _worker_main(rank, world, port, conn, transport, device, grid, extra_env):
    """Body of a worker process (fresh interpreter, entered from `_spawn_worker`).  Environment first, GPU second."""
    global _in_worker
    _in_worker = True
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(device), "LPGP_DEVICE": str(device),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    os.environ.update({k: str(v) for k, v in (extra_env or {}).items()})
    os.environ.pop("LPGP_SPAWN", None)
    objects = {}
    try:
        from . import _dist, _engine
        comm = _dist.Comm.from_env()
        ctx = _engine.default_context()
        if world > 1:
            ctx.dist_init(comm, transport=transport, grid=grid)
        conn.send(("ok", {"rank": rank, "device": ctx.device_info(), "grid": ctx.grid if world > 1 else (1, 1)}))
    except BaseException as exc:  # noqa: BLE001
        conn.send(("err", f"worker {rank}: bring-up failed: {type(exc).__name__}: {exc}\n{traceback.format_exc()}"))
        return
    while True:
        try:
            msg = conn.recv()
        except EOFError:
            break
        op = msg[0]
        try:
            if op == "close":
                conn.send(("ok", None))
                break
            if op == "condition":
                _, new_id, parent_id, prior, Y, X, L, b = msg
                base = prior if parent_id is None else objects[parent_id]
                objects[new_id] = base.condition_on_observations(Y, X, L=L, b=b)
                conn.send(("ok", None))
            elif op == "method":
                _, oid, path, args, kwargs = msg
                target = objects[oid]
                for name in path.split("."):
                    target = getattr(target, name)
                out = target(*args, **kwargs) if callable(target) else target
                if path == "__call__":                       # a Normal: ship mean and covariance
                    out = (np.asarray(out.mean), np.asarray(out.cov))
                conn.send(("ok", out if rank == 0 else None))
            elif op == "release":
                objects.pop(msg[1], None)
                conn.send(("ok", None))
            elif op == "option":
                ctx.set_option(msg[1], msg[2])
                conn.send(("ok", None))
            elif op == "info":
                conn.send(("ok", {"rank": rank, "grid": ctx.grid if world > 1 else (1, 1), "stats": ctx.dist_stats()}))
            else:
                conn.send(("err", f"worker {rank}: unknown request {op!r}"))
        except BaseException as exc:  # noqa: BLE001
            conn.send(("exc", exc if isinstance(exc, (ValueError, TypeError, NotImplementedError, np.linalg.LinAlgError)) else None,
                       f"worker {rank}: {type(exc).__name__}: {exc}"))
    try:
        objects.clear()
        comm.close()
    except Exception:  # noqa: BLE001
        pass


def _listener_socket(listener):
    """The listening socket behind a `multiprocessing.connection.Listener` (for `select`): the class exposes no file
    descriptor of its own; guarded so that a change of its internals fails loudly at bring-up, not as a hang."""
    sock = getattr(getattr(listener, "_listener", None), "_socket", None)
    if sock is None or not hasattr(sock, "fileno"):
        raise RuntimeError("multi-GPU front: cannot reach the listening socket of multiprocessing.connection.Listener")
    return sock


class WorkerGroup:
    """`n_gpus` worker processes, one per GPU, that replay the collective calls of this process."""

    def __init__(self, n_gpus: int, *, transport: str = "rccl", devices=None, grid=None, env=None, rccl_loopback: bool = False,
                 timeout: float = 600.0):
        from . import _engine
        if _engine._default_ctx is not None:
            raise RuntimeError("spawn() must be called before this process makes its first GPU call "
                               "(a context already exists; workers are never started from a process that has touched the GPU)")
        if n_gpus < 1:
            raise ValueError("n_gpus must be >= 1")
        self.world = int(n_gpus)
        self.timeout = timeout
        devices = list(range(self.world)) if devices is None else list(devices)
        port = _free_port()
        # Workers run `python -m linpde_gp_amd._spawn_worker` -- a module of this package, NOT the caller's __main__
        # (multiprocessing's "spawn" start method re-imports the caller's script in every child: an unguarded script
        # would call spawn() again there).  They dial back over an authenticated AF_UNIX connection.
        self._tmp = tempfile.mkdtemp(prefix="lpgp-spawn-")
        authkey = secrets.token_bytes(32)
        listener = Listener(os.path.join(self._tmp, "ctl"), family="AF_UNIX", authkey=authkey)
        pkg_parent = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        self._procs, self._conns = [], []
        try:
            child_env = {k: v for k, v in os.environ.items() if k != "LPGP_SPAWN"}
            child_env["PYTHONPATH"] = os.pathsep.join([pkg_parent] + [p_ for p_ in child_env.get("PYTHONPATH", "").split(os.pathsep) if p_])
            child_env["LPGP_SPAWN_AUTHKEY"] = authkey.hex()
            for r in range(self.world):
                self._procs.append(subprocess.Popen([sys.executable, "-m", "linpde_gp_amd._spawn_worker", listener.address, str(r)],
                                                    env=child_env, stdin=subprocess.DEVNULL))
            by_rank, t_end = {}, time.monotonic() + timeout
            lsock = _listener_socket(listener)
            while len(by_rank) < self.world:
                dead = [r for r, p in enumerate(self._procs) if p.poll() is not None and r not in by_rank]
                if dead:
                    raise RuntimeError(f"multi-GPU front: worker {dead[0]} died during bring-up (before it dialled back)")
                remaining = t_end - time.monotonic()
                if remaining <= 0:
                    raise RuntimeError(f"multi-GPU front: a worker did not dial back within {timeout:.0f} s")
                # wait for a pending connection on the listening socket itself: `Listener.accept` has no timeout, and its
                # authentication handshake blocks -- so it is only entered when a peer is known to be waiting
                if not select.select([lsock], [], [], min(0.5, remaining))[0]:
                    continue
                try:
                    c = listener.accept()
                except (multiprocessing.AuthenticationError, EOFError, OSError):
                    continue                       # a peer that failed the handshake or hung up: not one of ours
                # a worker that connected but died or hangs before it sent its rank must not hold the parent past `timeout`
                if not c.poll(max(0.0, min(30.0, t_end - time.monotonic()))):
                    c.close()
                    continue
                try:
                    r = c.recv()
                except (EOFError, OSError):
                    c.close()
                    continue
                if not isinstance(r, int) or not 0 <= r < self.world or r in by_rank:
                    c.close()
                    raise RuntimeError(f"multi-GPU front: a worker announced rank {r!r} (world {self.world}, registered {sorted(by_rank)})")
                by_rank[r] = c
            for r in range(self.world):
                extra = dict(env or {})
                if rccl_loopback:
                    # bring-up aid for ranks that SHARE one GPU (tests): RCCL takes them for different hosts and runs its socket
                    # transport over the loopback interface -- the product's RCCL code path, not the xGMI data path
                    extra.update(NCCL_HOSTID=f"lpgp-spawn-host-{r}", NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_NET="Socket")
                by_rank[r].send((r, self.world, port, transport, devices[r], grid, extra))
                self._conns.append(by_rank[r])
        except BaseException:
            for p in self._procs:
                p.kill()
            shutil.rmtree(self._tmp, ignore_errors=True)
            raise
        finally:
            listener.close()
        self._ids = itertools.count(1)
        self.info = self._collect("bring-up")
        atexit.register(self.close)

    # ---- plumbing ----
    def _collect(self, what):
        replies = []
        for r, c in enumerate(self._conns):
            if not c.poll(self.timeout):
                self.close(force=True)
                raise RuntimeError(f"multi-GPU front: worker {r} did not answer ({what}) within {self.timeout:.0f} s")
            try:
                replies.append(c.recv())
            except EOFError:
                self.close(force=True)
                raise RuntimeError(f"multi-GPU front: worker {r} died during {what}") from None
        bad = [rep for rep in replies if rep[0] != "ok"]
        if bad:
            first = bad[0]
            if first[0] == "exc" and first[1] is not None and all(rep[0] == "exc" for rep in replies):
                raise first[1]                          # the same API error on every rank: re-raise it as the reference would
            text = "; ".join(rep[-1] if rep[0] == "exc" else str(rep[1]) for rep in bad)
            raise RuntimeError("multi-GPU front: " + text)
        return replies[0][1]

    def request(self, *msg):
        if self._conns is None:
            raise RuntimeError("multi-GPU front: the worker group is closed")
        for c in self._conns:
            c.send(msg)
        return self._collect(msg[0])

    def new_id(self) -> int:
        return next(self._ids)

    def set_option(self, key: str, value: int):
        self.request("option", key, int(value))

    def close(self, force: bool = False):
        global _active
        conns, self._conns = self._conns, None
        if conns is None:
            return
        if not force:
            for c in conns:
                try:
                    c.send(("close",))
                except Exception:  # noqa: BLE001
                    pass
        for p in self._procs:
            try:
                p.wait(timeout=5.0 if not force else 0.5)
            except subprocess.TimeoutExpired:
                p.terminate()
                try:
                    p.wait(timeout=5.0)
                except subprocess.TimeoutExpired:
                    p.kill()
        for c in conns:
            c.close()
        shutil.rmtree(self._tmp, ignore_errors=True)
        if _active is self:
            _active = None


class _RemoteCov:
    def __init__(self, owner):
        self._o = owner

    def matrix(self, x0, x1=None):
        return self._o._call("cov.matrix", x0, x1)

    def __call__(self, x0, x1=None):
        return self._o._call("cov", x0, x1)


class RemoteConditionalGaussianProcess:
    """Proxy of a `ConditionalGaussianProcess` that lives, sharded, in the worker processes (same surface)."""

    def __init__(self, group: WorkerGroup, oid: int, prior):
        self._group, self._oid, self._prior = group, oid, prior

    @property
    def prior(self):
        return self._prior

    def _call(self, path, *args, **kwargs):
        return self._group.request("method", self._oid, path, args, kwargs)

    def condition_on_observations(self, Y, X=None, *, L=None, b=None):
        return condition(self._prior, Y, X, L=L, b=b, parent=self)

    def mean(self, x):
        return self._call("mean", x)

    def var(self, x):
        return self._call("var", x)

    def std(self, x):
        return self._call("std", x)

    def predict(self, x, *, return_var: bool = True):
        return self._call("predict", x, return_var=return_var)

    @property
    def cov(self):
        return _RemoteCov(self)

    @property
    def representer_weights(self):
        return self._call("representer_weights")

    def __call__(self, x):
        from . import randvars
        mean, cov = self._call("__call__", x)
        return randvars.Normal(mean, cov)

    def __del__(self):  # pragma: no cover
        try:
            if self._group._conns is not None:
                self._group.request("release", self._oid)
        except Exception:  # noqa: BLE001
            pass


def spawn(n_gpus: int, **kwargs) -> WorkerGroup:
    """Start the worker processes (once per process, before any GPU call) and route every following
    `condition_on_observations` of this process to them."""
    global _active
    if _in_worker:
        raise RuntimeError("spawn() inside a worker process")
    if _active is not None:
        raise RuntimeError("spawn(): a worker group is already active (close() it first)")
    _active = WorkerGroup(n_gpus, **kwargs)
    return _active


def active():
    """The worker group conditioning calls are routed to, or None (LPGP_SPAWN=n starts one on demand)."""
    global _active
    if _in_worker:
        return None
    if _active is None and int(os.environ.get("LPGP_SPAWN", "0") or 0) > 1:
        from . import _engine
        if _engine._default_ctx is None:
            kw = {}
            if os.environ.get("LPGP_SPAWN_DEVICES"):            # e.g. "0,0": ranks sharing a device (bring-up / tests)
                kw["devices"] = [int(d) for d in os.environ["LPGP_SPAWN_DEVICES"].split(",")]
            if os.environ.get("LPGP_SPAWN_TRANSPORT"):
                kw["transport"] = os.environ["LPGP_SPAWN_TRANSPORT"]
            _active = WorkerGroup(int(os.environ["LPGP_SPAWN"]), **kw)
    return _active


def condition(prior, Y, X, *, L, b, parent=None):
    """Forwarded `condition_on_observations`: same validation errors as the local call (raised by the workers and
    re-raised here), returns the proxy of the new posterior."""
    group = _active
    X = np.asarray(X) if (X is not None and type(X).__name__ == "DeviceArray") else X
    new_id = group.new_id()
    group.request("condition", new_id, None if parent is None else parent._oid, prior if parent is None else None, Y, X, L, b)
    return RemoteConditionalGaussianProcess(group, new_id, prior)
