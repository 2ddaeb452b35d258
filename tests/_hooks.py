"""Test hooks (include/lpgp_test.h, liblpgp_testhooks.so): raw kernels on host buffers, peak probes, the host replay of the
distributed tile enumeration.  TEST INFRASTRUCTURE -- the product package never loads this library
(`nm -D liblpgp.so | grep lpgp_test` is empty).  Used by tests/ and scratch/."""

from __future__ import annotations

import ctypes as C
import os

import numpy as np
import scipy.linalg

from linpde_gp_amd import _lib
from linpde_gp_amd._engine import Context
from linpde_gp_amd._lib import check

HOOKS_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "liblpgp_testhooks.so")

EXPORTED = ["lpgp_test_stair_enumerate", "lpgp_test_gemm", "lpgp_test_potrf_tile", "lpgp_test_tile_step", "lpgp_test_panel_solve",
            "lpgp_debug_tile_xcc", "lpgp_probe_mfma_f64", "lpgp_probe_hbm_write", "lpgp_test_force_status"]


def _load() -> C.CDLL:
    if not os.path.exists(HOOKS_PATH):
        raise ImportError(f"{HOOKS_PATH} not found (linpde-gp_amd/csrc/build.sh builds it next to liblpgp.so)")
    lib = C.CDLL(HOOKS_PATH)          # resolves its liblpgp.so through $ORIGIN: the library the package has already loaded
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    pd = C.POINTER(C.c_double)

    def sig(name, res, *args):
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = list(args)

    sig("lpgp_test_stair_enumerate", C.c_int, i32, i32, i32, i32, i32, i32, i32, i32, C.POINTER(i32), i64)
    sig("lpgp_test_gemm", C.c_int, vp, i32, i32, i32, i64, i64, i64, dbl, pd, i64, pd, i64, dbl, pd, i64, i32, pd)
    sig("lpgp_test_potrf_tile", C.c_int, vp, pd, pd, C.POINTER(i32))
    sig("lpgp_debug_tile_xcc", C.c_int, vp, C.POINTER(i32), i32)
    sig("lpgp_test_tile_step", C.c_int, vp, i32, pd, i64, pd, pd, pd)
    sig("lpgp_test_panel_solve", C.c_int, vp, i32, pd, i32, i64, pd, pd, pd)
    sig("lpgp_probe_mfma_f64", C.c_int, vp, pd)
    sig("lpgp_probe_hbm_write", C.c_int, vp, i64, pd)
    sig("lpgp_test_force_status", C.c_int, vp, vp, i32)
    return lib


lib = _load()


def probe_mfma_f64(ctx: Context) -> float:
    v = C.c_double()
    check(lib.lpgp_probe_mfma_f64(ctx._h, C.byref(v)), "lpgp_probe_mfma_f64")
    return v.value


def probe_hbm_write(ctx: Context, nbytes: int = 1 << 30) -> float:
    v = C.c_double()
    check(lib.lpgp_probe_hbm_write(ctx._h, int(nbytes), C.byref(v)), "lpgp_probe_hbm_write")
    return v.value


def test_gemm(ctx: Context, ta: int, tb: int, lower_only: int, alpha: float, A: np.ndarray, B: np.ndarray,
              beta: float, Cm: np.ndarray, k: int, reps: int = 0):
    """Raw GEMM on column-major (Fortran-ordered) arrays; returns (C, ms_per_rep)."""
    A = np.asfortranarray(A, dtype=np.double)
    B = np.asfortranarray(B, dtype=np.double)
    Cm = np.asfortranarray(Cm, dtype=np.double).copy(order="F")
    m, n = Cm.shape
    ms = C.c_double(0.0)
    pd = C.POINTER(C.c_double)
    check(lib.lpgp_test_gemm(ctx._h, ta, tb, lower_only, m, n, k, alpha,
                             A.ctypes.data_as(pd), A.shape[0], B.ctypes.data_as(pd), B.shape[0], beta,
                             Cm.ctypes.data_as(pd), Cm.shape[0], reps, C.byref(ms)), "lpgp_test_gemm")
    return Cm, ms.value


def test_tile_step(ctx: Context, which: int, XV: np.ndarray, L: np.ndarray, Linv: np.ndarray):
    """In-place refined tile solve on host buffers (`lpgp_test_tile_step`): which = 0: X (rows x 128) <- X L^{-T},
    which = 1: V (128 x cols) <- L^{-1} V.  Returns (result, milliseconds)."""
    XV = np.asfortranarray(XV, dtype=np.double).copy(order="F")
    L = np.asfortranarray(np.tril(L), dtype=np.double)
    Linv = np.asfortranarray(np.tril(Linv), dtype=np.double)
    n = XV.shape[0] if which == 0 else XV.shape[1]
    ms = C.c_double(0.0)
    pd = C.POINTER(C.c_double)
    check(lib.lpgp_test_tile_step(ctx._h, which, XV.ctypes.data_as(pd), n, L.ctypes.data_as(pd), Linv.ctypes.data_as(pd),
                                  C.byref(ms)), "lpgp_test_tile_step")
    return XV, ms.value


def test_panel_solve(ctx: Context, V: np.ndarray, Lblk: np.ndarray, rows_form: bool = False):
    """Fused panel chain (`lpgp_test_panel_solve`): V (nt*128 x cols) <- Lblk^{-1} V, or with `rows_form`
    X (cols x nt*128) <- X Lblk^{-T}.  Returns (result, milliseconds)."""
    if rows_form:
        cols, rows = V.shape
    else:
        rows, cols = V.shape
    nt = rows // 128
    V = np.asfortranarray(V, dtype=np.double).copy(order="F")
    Lb = np.asfortranarray(np.tril(Lblk), dtype=np.double)
    Linv = np.concatenate([np.asfortranarray(np.tril(scipy.linalg.solve_triangular(
        Lb[t * 128:(t + 1) * 128, t * 128:(t + 1) * 128], np.eye(128), lower=True))).reshape(-1, order="F") for t in range(nt)])
    ms = C.c_double(0.0)
    pd = C.POINTER(C.c_double)
    check(lib.lpgp_test_panel_solve(ctx._h, int(bool(rows_form)), V.ctypes.data_as(pd), nt, cols, Lb.ctypes.data_as(pd),
                                    np.ascontiguousarray(Linv).ctypes.data_as(pd), C.byref(ms)), "lpgp_test_panel_solve")
    return V, ms.value


def test_potrf_tile(ctx: Context, T: np.ndarray):
    T = np.asfortranarray(T, dtype=np.double).copy(order="F")
    Linv = np.zeros((128, 128), order="F")
    info = C.c_int32()
    pd = C.POINTER(C.c_double)
    check(lib.lpgp_test_potrf_tile(ctx._h, T.ctypes.data_as(pd), Linv.ctypes.data_as(pd), C.byref(info)),
          "lpgp_test_potrf_tile")
    return T, Linv, info.value


def force_status(ctx: Context, mat, value: int) -> None:
    """Leave the status word of `mat` (a `_engine.GramMatrix`) as an enqueued factorisation ending with `value` would."""
    check(lib.lpgp_test_force_status(ctx._h, mat._h, int(value)), "lpgp_test_force_status")
