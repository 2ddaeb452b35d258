"""Object layer over the C ABI: device context, point sets, the block Gram matrix /
Cholesky factor and right-hand-side blocks.  Pure plumbing (ctypes + NumPy); all
arithmetic of the hot path happens in liblpgp.so on the GPU."""

from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

from . import _lib, config
from ._lib import check, as_pd, lib


class Context:
    """One per process / per GPU (`lpgp_init`)."""

    def __init__(self, device: int | None = None):
        if device is None:
            device = int(os.environ.get("LPGP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        h = C.c_void_p()
        check(lib.lpgp_init(int(device), C.byref(h)), "lpgp_init")
        self._h = h
        self.device = int(device)
        self.rank, self.world, self.comm = 0, 1, None
        self.distributed = False     # joined a multi-GPU job (`lpgp_dist_init*` succeeded): every call on a matrix is collective

    def dist_init(self, comm, transport: str = "rccl", grid: "tuple[int, int] | None" = None) -> None:
        """Join a one-process-per-GPU job.  `comm` is the control plane (`_dist.Comm`).
        `grid` = (Pr, Pc) process grid of the 2-D block-cyclic tile distribution (rank = r * Pc + c); default:
        $LPGP_GRID ("2x4") or the library's choice (Pr = world, Pc = 1, see lpgp.h).
        transport "rccl" (the product path): rank 0 creates the RCCL unique id, everybody receives it, then
        `ncclCommInitRank`; panels travel as grouped point-to-point sends over xGMI.  transport "ipc": direct-peer
        pushes into IPC-mapped receive windows (device-to-device copies, barriers over the control plane; several ranks
        may share one GPU).  transport "host"
        (bring-up / tests): the same messages staged through the host and exchanged over the control plane, so
        that several ranks may share one GPU.  From here on every call on a `GramMatrix` is collective."""
        if transport not in ("rccl", "host", "ipc"):
            raise ValueError("transport must be 'rccl', 'ipc' or 'host'")
        if comm.world == 1 and not os.environ.get("LPGP_FORCE_RCCL") and transport == "rccl":
            self.comm = comm
            return
        if grid is None and os.environ.get("LPGP_GRID"):
            a, b = os.environ["LPGP_GRID"].lower().split("x")
            grid = (int(a), int(b))
        if grid is not None:
            if grid[0] * grid[1] != comm.world:
                raise ValueError(f"grid {grid[0]} x {grid[1]} does not match {comm.world} ranks")
            check(lib.lpgp_dist_set_grid(self._h, int(grid[0]), int(grid[1])), "lpgp_dist_set_grid")
        if transport in ("host", "ipc"):
            def exchange(_user, op, buf, nbytes, root):
                try:
                    view = (C.c_char * nbytes).from_address(buf)
                    if op == 0:
                        data = comm.bcast_from(bytes(view) if comm.rank == root else None, root)
                        if comm.rank != root:
                            view[:] = data
                    else:
                        vals = comm.allgather(int.from_bytes(bytes(view), "little", signed=True))
                        view[:] = int(max(vals)).to_bytes(nbytes, "little", signed=True)
                    return 0
                except Exception:  # noqa: BLE001 (reported through the C return code)
                    return 1
            self._host_exchange = _lib.HOST_EXCHANGE_FN(exchange)      # keep the callback alive
            if transport == "ipc":
                # direct-peer transport: export this rank's receive window, gather everybody's handle, map the peers'
                # windows; panel pieces then travel device to device, the control plane only carries barriers
                handle = C.create_string_buffer(64)
                window = int(os.environ.get("LPGP_IPC_WINDOW_MB", "512")) << 20
                rc = lib.lpgp_dist_ipc_export(self._h, window, handle)
                err = "" if rc == 0 else lib.lpgp_last_error().decode(errors="replace")
                got = comm.allgather((rc == 0, err, handle.raw))
                if not all(g[0] for g in got):
                    raise _lib.LpgpError("lpgp_dist_ipc_export failed: " + "; ".join(f"rank {r}: {g[1]}" for r, g in enumerate(got) if not g[0]))
                rc = lib.lpgp_dist_init_ipc(self._h, comm.rank, comm.world, b"".join(g[2] for g in got), self._host_exchange, None)
                err = "" if rc == 0 else lib.lpgp_last_error().decode(errors="replace")
                res = comm.allgather((rc == 0, err))
                if not all(o for o, _ in res):
                    raise _lib.LpgpError("lpgp_dist_init_ipc failed: " + "; ".join(f"rank {r}: {e}" for r, (o, e) in enumerate(res) if not o))
            else:
                check(lib.lpgp_dist_init_host(self._h, comm.rank, comm.world, self._host_exchange, None),
                      "lpgp_dist_init_host")
            self.rank, self.world, self.comm = comm.rank, comm.world, comm
            self.distributed = True
            return
        # every rank leaves the bring-up in step, also when it fails: rank 0 broadcasts (ok, id-or-error), and the
        # outcome of ncclCommInitRank is agreed on before anybody raises (a rank that raised alone would leave the
        # others blocked in their first collective)
        msg = None
        if comm.rank == 0:
            buf = C.create_string_buffer(128)
            rc = lib.lpgp_dist_unique_id(buf)
            msg = (True, buf.raw) if rc == 0 else (False, lib.lpgp_last_error().decode(errors="replace"))
        ok, payload = comm.bcast(msg)
        if not ok:
            raise _lib.LpgpError(f"lpgp_dist_unique_id failed on rank 0: {payload}")
        rc = lib.lpgp_dist_init(self._h, comm.rank, comm.world, payload)
        err = "" if rc == 0 else lib.lpgp_last_error().decode(errors="replace")
        results = comm.allgather((rc == 0, err))
        if not all(o for o, _ in results):
            raise _lib.LpgpError("lpgp_dist_init failed: " + "; ".join(f"rank {r}: {e}" for r, (o, e) in enumerate(results) if not o))
        self.rank, self.world, self.comm = comm.rank, comm.world, comm
        self.distributed = True

    @property
    def grid(self) -> "tuple[int, int]":
        a, b = C.c_int32(), C.c_int32()
        check(lib.lpgp_dist_grid(self._h, C.byref(a), C.byref(b)), "lpgp_dist_grid")
        return a.value, b.value

    def dist_set_grid(self, pr: int, pc: int):
        """Pr x Pc process grid (the same call on every rank; after the bring-up only while no matrix is alive)."""
        check(lib.lpgp_dist_set_grid(self._h, int(pr), int(pc)), "lpgp_dist_set_grid")

    def link_probe(self, nbytes: int = 32 << 20, reps: int = 3) -> dict:
        """Measured rates of the panel-exchange transport (`lpgp_dist_link_probe`; collective), gathered over all ranks:
        `pair_gbps[s][d]` (s -> d alone), `one_to_all_gbps[s]` (one link of s while it sends to every peer),
        `all_to_all_inbound_gbps[r]` (total inbound of rank r while everybody sends to everybody)."""
        W = self.world
        out = np.zeros(W * W + W + 2)
        check(lib.lpgp_dist_link_probe(self._h, int(nbytes), int(reps), as_pd(out)), "lpgp_dist_link_probe")
        rows = self.comm.allgather(out)
        best = np.max(np.stack(rows), axis=0)
        pair = best[:W * W].reshape(W, W)
        off = ~np.eye(W, dtype=bool)
        return {"bytes": int(nbytes), "bytes_per_message": int(out[W * W + W + 1]), "reps": int(reps),
                "pair_gbps": [[round(float(v), 2) for v in r_] for r_ in pair],
                "pair_min_gbps": float(pair[off].min()) if W > 1 else 0.0,
                "pair_median_gbps": float(np.median(pair[off])) if W > 1 else 0.0,
                "pair_max_gbps": float(pair[off].max()) if W > 1 else 0.0,
                "one_to_all_gbps": [round(float(v), 2) for v in best[W * W:W * W + W]],
                "all_to_all_inbound_gbps": [round(float(r_[W * W + W]), 2) for r_ in rows]}

    def dist_stats(self, reset: bool = False) -> dict:
        """Bytes this rank sent / received in panel exchanges (seconds inside them: `profile_get()["comm"]`)."""
        s, r = C.c_double(), C.c_double()
        check(lib.lpgp_dist_stats(self._h, C.byref(s), C.byref(r), int(bool(reset))), "lpgp_dist_stats")
        return {"bytes_sent": s.value, "bytes_received": r.value}

    def close(self):
        if self._h:
            lib.lpgp_finalize(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(lib.lpgp_sync(self._h), "lpgp_sync")

    def set_option(self, key: str, value: int):
        global _option_epoch
        check(lib.lpgp_set_option(self._h, key.encode(), int(value)), "lpgp_set_option")
        _option_epoch += 1           # cached predictions were computed with other kernels / schedules

    def get_option(self, key: str) -> int:
        v = C.c_int64()
        check(lib.lpgp_get_option(self._h, key.encode(), C.byref(v)), "lpgp_get_option")
        return int(v.value)

    def roofline_kernel_symbol(self) -> str:
        """Symbol of the kernel behind profiling slot "syrk_trailing" (rank-nb trailing update, remainder half) as rocprofv3 prints it."""
        return "gemm3_f64_kernel<false, 1>" if (self.get_option("gemm3") > 0 and self.get_option("gemm3_fact")) else "gemm_f64_kernel<false, false, 1>"

    def device_info(self) -> dict:
        name = C.create_string_buffer(256)
        cus = C.c_int()
        hbm = C.c_int64()
        check(lib.lpgp_device_info(self._h, name, 256, C.byref(cus), C.byref(hbm)), "lpgp_device_info")
        return {"name": name.value.decode(), "cus": cus.value, "hbm_bytes": hbm.value}

    # ---- measurement ----
    def profile_enable(self, on=True):
        """`on`: False/0 = off, True = every kernel, or an iterable of kernel names
        (`_lib.KERNEL_NAMES`) to bracket only those (fewer event records = less perturbation)."""
        if on is True:
            mask = -1
        elif not on:
            mask = 0
        else:
            mask = 0
            for name in on:
                mask |= 1 << _lib.KERNEL_NAMES.index(name)
        check(lib.lpgp_profile_enable(self._h, mask), "lpgp_profile_enable")

    def profile_reset(self):
        check(lib.lpgp_profile_reset(self._h), "lpgp_profile_reset")

    def profile_get(self) -> dict:
        out = {}
        for kid, name in enumerate(_lib.KERNEL_NAMES):
            ms, fl, by = C.c_double(), C.c_double(), C.c_double()
            n = C.c_int64()
            check(lib.lpgp_profile_get(self._h, kid, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)),
                  "lpgp_profile_get")
            out[name] = {"ms": ms.value, "launches": n.value, "flops": fl.value, "bytes": by.value}
        return out


_default_ctx = None
_default_lock = threading.Lock()
_option_epoch = 0


def option_epoch() -> int:
    """Bumped by every `Context.set_option`: a posterior's cached last prediction is valid for one epoch."""
    return _option_epoch


def default_context() -> Context:
    """Process-wide context on device $LPGP_DEVICE / $LOCAL_RANK / 0."""
    global _default_ctx
    with _default_lock:
        if _default_ctx is None:
            _default_ctx = Context()
        return _default_ctx


class Points:
    """Device-resident point set (`lpgp_pts_create`)."""

    def __init__(self, ctx: Context, X: np.ndarray):
        X = np.ascontiguousarray(np.asarray(X, dtype=np.double))
        if X.ndim != 2:
            raise ValueError(f"points must have shape (n, d), got {X.shape}")
        self.ctx = ctx
        self.n, self.d = X.shape
        h = C.c_void_p()
        check(lib.lpgp_pts_create(ctx._h, as_pd(X), self.n, self.d, C.byref(h)), "lpgp_pts_create")
        self._h = h
        # 1-D factor point sets if the points are a tensor grid (C order): enables the Kronecker
        # assembly `lpgp_gram_assemble_grid`
        self.grid_factors: "tuple[Points, ...] | None" = None

    def __del__(self):  # pragma: no cover
        if getattr(self, "_h", None) and self.ctx._h:
            lib.lpgp_pts_destroy(self._h)
            self._h = None


def _grid_factor_points(ctx: "Context", X_original, X_flat: np.ndarray):
    """Factor point sets of `X_original` if it is a `TensorProductGrid` (carries `.factors`) whose
    points, flattened in C order, are exactly `X_flat` (a sliced or reordered view is not)."""
    factors = getattr(X_original, "factors", None)
    if (not config.use_grid_assembly or factors is None or len(factors) != X_flat.shape[1] or len(factors) < 2
            or X_flat.shape[0] < config.grid_assembly_min_points):
        return None
    factors = [np.ascontiguousarray(f, dtype=np.double).reshape(-1) for f in factors]
    if int(np.prod([f.size for f in factors])) != X_flat.shape[0]:
        return None
    mesh = np.stack(np.meshgrid(*factors, indexing="ij"), axis=-1).reshape(-1, len(factors))
    if not np.array_equal(mesh, X_flat):
        return None
    return tuple(Points(ctx, f[:, None]) for f in factors)


class DeviceArray(np.ndarray):
    """Host array whose points are already resident in HBM (`to_device`).  Passing one as
    `X` to `condition_on_observations` / `predict` skips the per-call host->device copy."""

    _lpgp_points: "Points | None" = None

    def __array_finalize__(self, obj):
        # views / slices do not inherit the device handle
        self._lpgp_points = None


def to_device(X, input_shape=None, ctx: Context | None = None) -> DeviceArray:
    """Upload a point array of shape batch + input_shape once; returns an ndarray subclass
    that carries the device handle (and, for a `TensorProductGrid`, its factor point sets)."""
    from . import _spawn
    if ctx is None and _spawn.active() is not None:
        return X                     # single-process multi-GPU front: this process holds no GPU; the workers upload their own copies
    X_in = X
    X = np.ascontiguousarray(np.asarray(X, dtype=np.double))
    if input_shape is None:
        d = X.shape[-1] if X.ndim >= 2 else 1
    else:
        d = max(int(np.prod(input_shape, dtype=int)), 1)
    ctx = ctx or default_context()
    out = X.view(DeviceArray)
    out._lpgp_points = Points(ctx, X.reshape(-1, d))
    out._lpgp_points.grid_factors = _grid_factor_points(ctx, X_in, X.reshape(-1, d))
    return out


def as_points(ctx: Context, X_original, X_flat: np.ndarray) -> Points:
    h = getattr(X_original, "_lpgp_points", None)
    if h is not None and h.ctx is ctx and h.n == X_flat.shape[0] and h.d == X_flat.shape[1]:
        return h
    pts = Points(ctx, X_flat)
    pts.grid_factors = _grid_factor_points(ctx, X_original, X_flat)
    return pts


class GramMatrix:
    """Block Gram matrix that becomes its own Cholesky factor (`lpgp_mat_*`, `lpgp_potrf`)."""

    def __init__(self, ctx: Context, capacity_hint: int = 0, _handle=None, _block_sizes=None):
        self.ctx = ctx
        if _handle is None:
            _handle = C.c_void_p()
            check(lib.lpgp_mat_create(ctx._h, int(capacity_hint), C.byref(_handle)), "lpgp_mat_create")
        self._h = _handle
        self.block_sizes: list[int] = list(_block_sizes or [])

    def __del__(self):  # pragma: no cover
        if getattr(self, "_h", None) and self.ctx._h:
            lib.lpgp_mat_destroy(self._h)
            self._h = None

    @property
    def n(self) -> int:
        return int(lib.lpgp_mat_size(self._h))

    @property
    def padded_n(self) -> int:
        return int(lib.lpgp_mat_padded_size(self._h))

    def add_block(self, n: int) -> int:
        bi = lib.lpgp_mat_add_block(self.ctx._h, self._h, int(n))
        if bi < 0:
            check(bi, "lpgp_mat_add_block")
        self.block_sizes.append(int(n))
        return bi

    def pop_block(self) -> None:
        """Undo the last `add_block` (its block not factored): rollback of a failed conditioning."""
        check(lib.lpgp_mat_pop_block(self.ctx._h, self._h), "lpgp_mat_pop_block")
        self.block_sizes.pop()

    @property
    def num_blocks(self) -> int:
        """Blocks in view."""
        return int(lib.lpgp_mat_num_blocks(self._h))

    @property
    def num_blocks_total(self) -> int:
        return int(lib.lpgp_mat_num_blocks_total(self._h))

    def set_view(self, nblocks: int) -> None:
        """Restrict every solve / prediction to the leading `nblocks` blocks (-1: all), i.e. to the
        factor an earlier conditioning produced (`lpgp_mat_set_view`)."""
        check(lib.lpgp_mat_set_view(self.ctx._h, self._h, int(nblocks)), "lpgp_mat_set_view")

    def clone(self, nblocks: int) -> "GramMatrix":
        h = C.c_void_p()
        check(lib.lpgp_mat_clone(self.ctx._h, self._h, int(nblocks), C.byref(h)), "lpgp_mat_clone")
        return GramMatrix(self.ctx, _handle=h, _block_sizes=self.block_sizes[:nblocks])

    def assemble(self, kdesc, X0: Points, X1: Points | None, bi: int, bj: int):
        arr = _kdesc_array(kdesc)
        if X0.grid_factors is not None and (X1 is None or X1.grid_factors is not None) and lib.lpgp_kron_fits(arr, len(arr)):
            # both point sets are tensor grids: sum of Kronecker products of 1-D kernel matrices (sums too long for
            # its fixed-size tables are assembled entry-wise from the flattened grids like any other block)
            F0 = (C.c_void_p * len(X0.grid_factors))(*[f._h for f in X0.grid_factors])
            F1 = None if X1 is None else (C.c_void_p * len(X1.grid_factors))(*[f._h for f in X1.grid_factors])
            check(lib.lpgp_gram_assemble_grid(self.ctx._h, arr, len(arr), F0, F1, self._h, bi, bj),
                  "lpgp_gram_assemble_grid")
            return
        check(lib.lpgp_gram_assemble(self.ctx._h, arr, len(arr), X0._h, X1._h if X1 is not None else None,
                                     self._h, bi, bj), "lpgp_gram_assemble")

    def add_diag(self, bi: int, v: np.ndarray | None = None, scalar: float = 0.0):
        if v is not None:
            v = np.ascontiguousarray(v, dtype=np.double)
            if v.shape != (self.block_sizes[bi],):
                raise ValueError("diagonal has the wrong length")
        check(lib.lpgp_mat_add_diag(self.ctx._h, self._h, bi, as_pd(v) if v is not None else None, float(scalar)),
              "lpgp_mat_add_diag")

    def add_dense(self, bi: int, B: np.ndarray):
        B = np.ascontiguousarray(B, dtype=np.double)
        n = self.block_sizes[bi]
        if B.shape != (n, n):
            raise ValueError("dense noise block has the wrong shape")
        check(lib.lpgp_mat_add_dense(self.ctx._h, self._h, bi, as_pd(B)), "lpgp_mat_add_dense")

    def todense(self, what: str = "gram") -> np.ndarray:
        n = self.n
        out = np.empty((n, n))
        check(lib.lpgp_mat_to_host(self.ctx._h, self._h, 0 if what == "gram" else 1, as_pd(out)), "lpgp_mat_to_host")
        return out

    def factor_diag(self) -> np.ndarray:
        """Diagonal of the Cholesky factor (`lpgp_mat_factor_diag`)."""
        out = np.empty(self.n)
        check(lib.lpgp_mat_factor_diag(self.ctx._h, self._h, as_pd(out)), "lpgp_mat_factor_diag")
        return out

    def potrf(self) -> int:
        info = C.c_int32()
        check(lib.lpgp_potrf(self.ctx._h, self._h, C.byref(info)), "lpgp_potrf")
        return info.value

    def condition(self, n: int, X_new: "Points", row, noise_scalar: float = 0.0, noise_diag=None, noise_dense=None, lazy: "bool | int" = True) -> int:
        """One conditioning in one call (`lpgp_mat_condition`): declare the block of `n` rows, assemble its block row --
        `row` = [(kdesc, X_j or None)] for every earlier block j and, last, the diagonal block --, add the noise, factor
        (`lazy` = 1: enqueue; 2: assemble only, the factorisation is left to the first call that needs the factor).  Returns
        the factorisation status (0 when lazy).  On failure -- an error, or a status != 0 --
        the block has been dropped again."""
        arr = (_lib.CondBlock * len(row))()
        keep = []
        for e, (kdesc, X1) in zip(arr, row):
            kd = _kdesc_array(kdesc)
            keep.append(kd)
            e.kd, e.ngroups = C.cast(kd, C.POINTER(_lib.KDesc)), len(kd)
            if X_new.grid_factors is not None and (X1 is None or X1.grid_factors is not None) and lib.lpgp_kron_fits(kd, len(kd)):
                F0 = (C.c_void_p * len(X_new.grid_factors))(*[f._h for f in X_new.grid_factors])
                keep.append(F0)
                e.F0 = C.cast(F0, C.POINTER(C.c_void_p))
                if X1 is not None:
                    F1 = (C.c_void_p * len(X1.grid_factors))(*[f._h for f in X1.grid_factors])
                    keep.append(F1)
                    e.F1 = C.cast(F1, C.POINTER(C.c_void_p))
            elif X1 is not None:
                e.X1 = X1._h
        nd = None if noise_diag is None else np.ascontiguousarray(noise_diag, dtype=np.double)
        nD = None if noise_dense is None else np.ascontiguousarray(noise_dense, dtype=np.double)
        if nd is not None and nd.shape != (int(n),):
            raise ValueError("diagonal has the wrong length")
        if nD is not None and nD.shape != (int(n), int(n)):
            raise ValueError("dense noise block has the wrong shape")
        info = C.c_int32()
        check(lib.lpgp_mat_condition(self.ctx._h, self._h, int(n), X_new._h, arr, len(arr), float(noise_scalar),
                                     as_pd(nd) if nd is not None else None, as_pd(nD) if nD is not None else None, int(lazy), C.byref(info)),
              "lpgp_mat_condition")
        if info.value == 0:
            self.block_sizes.append(int(n))
        return info.value

    def potrf_enqueue(self) -> None:
        """The factorisation enqueued, no host synchronisation (`lpgp_potrf_enqueue`): its status is read by `check`."""
        check(lib.lpgp_potrf_enqueue(self.ctx._h, self._h), "lpgp_potrf_enqueue")

    def check(self):
        """(info, block): status of everything enqueued since the last check (waits for the device); info = 0: fine;
        k > 0: the k-th leading minor of the padded matrix is not positive definite, in observation block `block`."""
        info, block = C.c_int32(), C.c_int32()
        check(lib.lpgp_mat_check(self.ctx._h, self._h, C.byref(info), C.byref(block)), "lpgp_mat_check")
        return info.value, block.value

    def truncate(self, nblocks: int) -> None:
        """Drop the blocks from `nblocks` on (rollback of failed conditionings found by `check`)."""
        check(lib.lpgp_mat_truncate(self.ctx._h, self._h, int(nblocks)), "lpgp_mat_truncate")
        del self.block_sizes[int(nblocks):]

    def potrs(self, B: np.ndarray) -> np.ndarray:
        """Solve G X = B for B of shape (n,) or (n, nrhs)."""
        B = np.asarray(B, dtype=np.double)
        vec = B.ndim == 1
        Bc = np.asfortranarray(B.reshape(self.n, -1)).copy(order="F")
        buf = np.ascontiguousarray(Bc.T)          # (nrhs, n) C-order == (n, nrhs) column-major
        check(lib.lpgp_potrs(self.ctx._h, self._h, as_pd(buf), buf.shape[0]), "lpgp_potrs")
        X = buf.T
        return X[:, 0].copy() if vec else np.ascontiguousarray(X)

    def solve_weights(self, r: np.ndarray) -> np.ndarray:
        r = np.ascontiguousarray(r, dtype=np.double)
        if r.shape != (self.n,):
            raise ValueError(f"residual must have shape ({self.n},), got {r.shape}")
        w = np.empty_like(r)
        check(lib.lpgp_solve_weights(self.ctx._h, self._h, as_pd(r), as_pd(w)), "lpgp_solve_weights")
        return w

    def set_residual(self, r: np.ndarray) -> None:
        """Residual for the weight-free mean of `Rhs.predict(mean + variance)` (`lpgp_mat_set_residual`)."""
        r = np.ascontiguousarray(r, dtype=np.double)
        if r.shape != (self.n,):
            raise ValueError(f"residual must have shape ({self.n},), got {r.shape}")
        check(lib.lpgp_mat_set_residual(self.ctx._h, self._h, as_pd(r)), "lpgp_mat_set_residual")


class Rhs:
    """Device-resident n x m block sharing the row layout of a GramMatrix (`lpgp_rhs_*`)."""

    def __init__(self, ctx: Context, mat: GramMatrix, m: int):
        self.ctx, self.mat, self.m = ctx, mat, int(m)
        h = C.c_void_p()
        check(lib.lpgp_rhs_create(ctx._h, mat._h, self.m, C.byref(h)), "lpgp_rhs_create")
        self._h = h

    def __del__(self):  # pragma: no cover
        if getattr(self, "_h", None) and self.ctx._h:
            lib.lpgp_rhs_destroy(self._h)
            self._h = None

    def cross_assemble(self, kdesc, X_obs: Points, X_test: Points, bi: int):
        arr = _kdesc_array(kdesc)
        check(lib.lpgp_cross_assemble(self.ctx._h, arr, len(arr), X_obs._h, X_test._h, self._h, self.mat._h, bi),
              "lpgp_cross_assemble")

    def cross_assemble_row(self, entries, X_test: Points):
        """All observation blocks in one call (`lpgp_cross_assemble_row`): entries = [(kdesc, X_obs)] per block, in block order."""
        arr = (_lib.CrossBlock * len(entries))()
        keep = []
        for e, (kdesc, X_obs) in zip(arr, entries):
            kd = _kdesc_array(kdesc)
            keep.append(kd)
            e.kd, e.ngroups, e.X_obs = C.cast(kd, C.POINTER(_lib.KDesc)), len(kd), X_obs._h
        check(lib.lpgp_cross_assemble_row(self.ctx._h, arr, len(arr), X_test._h, self._h, self.mat._h), "lpgp_cross_assemble_row")

    def trsm_lower(self):
        check(lib.lpgp_trsm_lower(self.ctx._h, self.mat._h, self._h), "lpgp_trsm_lower")

    def predict(self, prior_mean: np.ndarray | None, kxx: np.ndarray | None, want_mean=True, want_var=True):
        mean = np.empty(self.m) if want_mean else None
        var = np.empty(self.m) if want_var else None
        pm = np.ascontiguousarray(prior_mean, dtype=np.double) if prior_mean is not None else None
        kx = np.ascontiguousarray(kxx, dtype=np.double) if kxx is not None else None
        check(lib.lpgp_predict(self.ctx._h, self.mat._h, self._h,
                               as_pd(pm) if pm is not None else None,
                               as_pd(kx) if kx is not None else None,
                               as_pd(mean) if mean is not None else None,
                               as_pd(var) if var is not None else None), "lpgp_predict")
        return mean, var

    def potrf_predict(self, prior_mean: np.ndarray | None, kxx: np.ndarray):
        """Factor the blocks that are not factored yet and predict mean and variance in ONE pipeline (`lpgp_potrf_predict`):
        the forward substitution of this right-hand side rides inside the factorisation.  The status is not read here."""
        mean, var = np.empty(self.m), np.empty(self.m)
        pm = np.ascontiguousarray(prior_mean, dtype=np.double) if prior_mean is not None else None
        kx = np.ascontiguousarray(kxx, dtype=np.double)
        check(lib.lpgp_potrf_predict(self.ctx._h, self.mat._h, self._h, as_pd(pm) if pm is not None else None, as_pd(kx),
                                     as_pd(mean), as_pd(var)), "lpgp_potrf_predict")
        return mean, var

    def inner(self, other: "Rhs") -> np.ndarray:
        out = np.empty((self.m, other.m))
        check(lib.lpgp_rhs_inner(self.ctx._h, self._h, other._h, as_pd(out)), "lpgp_rhs_inner")
        return out

    def matmul(self, B: np.ndarray) -> "Rhs":
        """A new block  self[:, :m] @ B  (B: m x k on the host) with the same row layout (`lpgp_rhs_matmul`)."""
        B = np.ascontiguousarray(B, dtype=np.double)
        if B.ndim != 2 or B.shape[0] != self.m:
            raise ValueError(f"shape mismatch: ({self.mat.n}, {self.m}) @ {B.shape}")
        h = C.c_void_p()
        check(lib.lpgp_rhs_matmul(self.ctx._h, self._h, as_pd(B), B.shape[1], C.byref(h)), "lpgp_rhs_matmul")
        out = Rhs.__new__(Rhs)
        out.ctx, out.mat, out.m, out._h = self.ctx, self.mat, int(B.shape[1]), h
        return out

    def to_host(self) -> np.ndarray:
        out = np.empty((self.mat.n, self.m))
        check(lib.lpgp_rhs_to_host(self.ctx._h, self.mat._h, self._h, as_pd(out)), "lpgp_rhs_to_host")
        return out


def _kdesc_array(kdesc):
    """ctypes descriptor array of a lowered kernel; an array built earlier (`lowered_array`) passes through."""
    return kdesc if isinstance(kdesc, C.Array) else _lib.make_kdesc_array(kdesc)


def lowered_array(kdesc):
    """The C-ABI form of a lowered kernel, built once: callers that assemble the same block structure again and again
    (a chain of conditionings, one posterior per step of a sweep) keep it instead of the Python description."""
    return _lib.make_kdesc_array(kdesc)


class DeviceVectors:
    """An n x m block of vectors resident in HBM (`lpgp_dvec`): the operands of the matrix-free path."""

    def __init__(self, ctx: Context, n: int, m: int, values: np.ndarray | None = None):
        self.ctx, self.n, self.m = ctx, int(n), int(m)
        h = C.c_void_p()
        check(lib.lpgp_dvec_create(ctx._h, self.n, self.m, C.byref(h)), "lpgp_dvec_create")
        self._h = h
        if values is not None:
            self.set(values)

    def __del__(self):  # pragma: no cover
        if getattr(self, "_h", None) and self.ctx._h:
            lib.lpgp_dvec_destroy(self._h)
            self._h = None

    def set(self, values: np.ndarray) -> None:
        v = np.ascontiguousarray(np.asarray(values, dtype=np.double).reshape(self.n, self.m))
        check(lib.lpgp_dvec_set(self.ctx._h, self._h, as_pd(v)), "lpgp_dvec_set")

    def get(self) -> np.ndarray:
        out = np.empty((self.n, self.m))
        check(lib.lpgp_dvec_get(self.ctx._h, self._h, as_pd(out)), "lpgp_dvec_get")
        return out

    def axpby(self, a: "DeviceVectors", b: "DeviceVectors", s: float) -> None:
        """self = a + s * b"""
        check(lib.lpgp_dvec_axpby(self.ctx._h, self._h, a._h, b._h, float(s)), "lpgp_dvec_axpby")

    def scale_rows_add(self, off: int, V: "DeviceVectors", d: np.ndarray) -> None:
        """self[off : off + len(d)] += diag(d) V[off : off + len(d)]"""
        d = np.ascontiguousarray(d, dtype=np.double)
        check(lib.lpgp_dvec_scale_rows_add(self.ctx._h, self._h, int(off), V._h, int(off), as_pd(d), d.size), "lpgp_dvec_scale_rows_add")


def kernel_matvec_dev(ctx: Context, kdesc, X0: Points, X1: Points, V: DeviceVectors, v_off: int, Y: DeviceVectors, y_off: int,
                      accumulate: bool) -> None:
    """Y[y_off : y_off + n0] (+)= K(X0, X1) V[v_off : v_off + n1] with every operand resident (`lpgp_kernel_matvec_dev`)."""
    arr = _kdesc_array(kdesc)
    check(lib.lpgp_kernel_matvec_dev(ctx._h, arr, len(arr), X0._h, X1._h, V._h, int(v_off), Y._h, int(y_off), int(bool(accumulate))),
          "lpgp_kernel_matvec_dev")


class DevicePCG:
    """Preconditioned conjugate gradients whose iteration is launches only (`lpgp_pcg_*`): `matvec(P, Q)` forms Q = G P on
    `DeviceVectors`; everything else -- dots, step lengths, the low-rank preconditioner -- stays on the device."""

    def __init__(self, ctx: Context, n: int, m: int, L: np.ndarray | None, Sinv: np.ndarray | None, delta: float):
        self.ctx, self.n, self.m = ctx, int(n), int(m)
        rank = 0 if L is None else int(L.shape[0])
        Lc = None if rank == 0 else np.ascontiguousarray(L, dtype=np.double)
        Sc = None if rank == 0 else np.ascontiguousarray(Sinv, dtype=np.double)
        h = C.c_void_p()
        check(lib.lpgp_pcg_create(ctx._h, self.n, self.m, rank, as_pd(Lc) if rank else None, as_pd(Sc) if rank else None, float(delta), C.byref(h)),
              "lpgp_pcg_create")
        self._h = h

    def __del__(self):  # pragma: no cover
        if getattr(self, "_h", None) and self.ctx._h:
            lib.lpgp_pcg_destroy(self._h)
            self._h = None

    def start(self, R, Z, P, bnorm: np.ndarray, rtol: float) -> np.ndarray:
        rel = np.empty(self.m)
        bn = np.ascontiguousarray(bnorm, dtype=np.double)
        check(lib.lpgp_pcg_start(self.ctx._h, self._h, R._h, Z._h, P._h, as_pd(bn), float(rtol), as_pd(rel)), "lpgp_pcg_start")
        return rel

    def step(self, X, R, Z, P, Q, rtol: float) -> np.ndarray:
        rel = np.empty(self.m)
        check(lib.lpgp_pcg_step(self.ctx._h, self._h, X._h, R._h, Z._h, P._h, Q._h, float(rtol), as_pd(rel)), "lpgp_pcg_step")
        return rel


def gemm(ctx: Context, A: np.ndarray, B: np.ndarray, *, transa: bool = False, transb: bool = False, alpha: float = 1.0,
         beta: float = 0.0, C: np.ndarray | None = None) -> np.ndarray:
    """alpha op(A) op(B) + beta C  on the device's fp64 MFMA kernel (`lpgp_gemm_host`): the dense products the reference leaves
    to NumPy around the path (`randvars/_normal.py:8-71`, `gram.todense()`).  Host arrays in, a fresh host array out."""
    A = np.ascontiguousarray(A, dtype=np.double)
    B = np.ascontiguousarray(B, dtype=np.double)
    if A.ndim != 2 or B.ndim != 2:
        raise ValueError("`gemm` needs two matrices")
    m, k = (A.shape[1], A.shape[0]) if transa else A.shape
    k2, n = (B.shape[1], B.shape[0]) if transb else B.shape
    if k != k2:
        raise ValueError(f"shape mismatch: op(A) is {m} x {k}, op(B) is {k2} x {n}")
    if beta != 0.0:
        if C is None or C.shape != (m, n):
            raise ValueError("`beta != 0` needs C of the result's shape")
        out = np.array(C, dtype=np.double, order="C", copy=True)
    else:
        out = np.empty((m, n))
    if m == 0 or n == 0:
        return out
    if k == 0:
        return out * beta if beta != 0.0 else np.zeros((m, n))
    check(lib.lpgp_gemm_host(ctx._h, int(transa), int(transb), m, n, k, float(alpha), as_pd(A), as_pd(B), float(beta), as_pd(out)),
          "lpgp_gemm_host")
    return out


def kernel_diag(ctx: Context, kdesc) -> float:
    arr = _kdesc_array(kdesc)
    v = C.c_double()
    check(lib.lpgp_kernel_diag(ctx._h, arr, len(arr), C.byref(v)), "lpgp_kernel_diag")
    return v.value


def kernel_matrix(ctx: Context, kdesc, X0: Points, X1: Points) -> np.ndarray:
    arr = _kdesc_array(kdesc)
    out = np.empty((X0.n, X1.n))
    check(lib.lpgp_kernel_matrix(ctx._h, arr, len(arr), X0._h, X1._h, as_pd(out)), "lpgp_kernel_matrix")
    return out


def kernel_matvec(ctx: Context, kdesc, X0: Points, X1: Points, V: np.ndarray) -> np.ndarray:
    """K(X0, X1) @ V without forming K (`lpgp_kernel_matvec`); V of shape (n1,) or (n1, nrhs)."""
    arr = _kdesc_array(kdesc)
    V = np.asarray(V, dtype=np.double)
    vec = V.ndim == 1
    V2 = np.ascontiguousarray(V.reshape(X1.n, -1))
    out = np.empty((X0.n, V2.shape[1]))
    check(lib.lpgp_kernel_matvec(ctx._h, arr, len(arr), X0._h, X1._h, as_pd(V2), V2.shape[1], as_pd(out)),
          "lpgp_kernel_matvec")
    return out[:, 0] if vec else out
