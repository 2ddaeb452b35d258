"""ctypes binding of liblpgp.so (C ABI: include/lpgp.h).

There is NO CPU fallback: importing this module without the built library, or creating
a `Context` without a visible MI355X, raises.  Build with
`linpde-gp_amd/csrc/build.sh` (or `__graft_entry__.build()`).
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LPGP_LIB", os.path.join(_HERE, "liblpgp.so"))     # $LPGP_LIB: a diagnostic build of the same library

MAXD, MAXT, MAXG = 4, 256, 16
MATERN_HALFINT, EXPQUAD, MATERN_ISO = 1, 2, 3
K_ASSEMBLE, K_SYRK, K_GEMM, K_POTRF_TILE, K_TRSM, K_COUNT = 0, 1, 2, 3, 4, 5
KERNEL_NAMES = ("assemble", "syrk_trailing", "gemm", "potrf_tile", "trsm_gemm", "syrk_panel", "gemm_small", "matvec", "syrk_lookahead",
                "assemble_grid", "panel_fused", "comm")


class Term(C.Structure):
    _fields_ = [("coef", C.c_double), ("n0", C.c_int32 * MAXD), ("n1", C.c_int32 * MAXD)]


class KDesc(C.Structure):
    _fields_ = [
        ("d", C.c_int32),
        ("family", C.c_int32 * MAXD),
        ("p", C.c_int32 * MAXD),
        ("lengthscale", C.c_double * MAXD),
        ("scale", C.c_double),
        ("nterms", C.c_int32),
        ("terms", Term * MAXT),
    ]


class CondBlock(C.Structure):
    """`lpgp_cond_block` (include/lpgp.h): one entry of the block row a conditioning assembles."""
    _fields_ = [("kd", C.POINTER(KDesc)), ("ngroups", C.c_int32), ("X1", C.c_void_p), ("F0", C.POINTER(C.c_void_p)), ("F1", C.POINTER(C.c_void_p))]


# int fn(void* user, int32 op, void* buf, int64 bytes, int32 root)   (lpgp_host_exchange_fn)
HOST_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int32)


class CrossBlock(C.Structure):
    """`lpgp_cross_block` (include/lpgp.h): one observation block of a cross-covariance row."""
    _fields_ = [("kd", C.POINTER(KDesc)), ("ngroups", C.c_int32), ("X_obs", C.c_void_p)]


class LpgpError(RuntimeError):
    pass


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built "
            "(run linpde-gp_amd/csrc/build.sh); there is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    pd = C.POINTER(C.c_double)
    pk = C.POINTER(KDesc)

    def sig(name, res, *args):
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = list(args)

    sig("lpgp_init", C.c_int, C.c_int, C.POINTER(vp))
    sig("lpgp_finalize", C.c_int, vp)
    sig("lpgp_last_error", C.c_char_p)
    sig("lpgp_device_info", C.c_int, vp, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(i64))
    sig("lpgp_sync", C.c_int, vp)
    sig("lpgp_set_option", C.c_int, vp, C.c_char_p, i64)
    sig("lpgp_get_option", C.c_int, vp, C.c_char_p, C.POINTER(i64))
    sig("lpgp_dist_unique_id", C.c_int, C.c_char_p)
    sig("lpgp_dist_init", C.c_int, vp, i32, i32, C.c_char_p)
    sig("lpgp_dist_info", C.c_int, vp, C.POINTER(i32), C.POINTER(i32))
    sig("lpgp_dist_set_grid", C.c_int, vp, i32, i32)
    sig("lpgp_dist_grid", C.c_int, vp, C.POINTER(i32), C.POINTER(i32))
    sig("lpgp_dist_stats", C.c_int, vp, pd, pd, i32)
    sig("lpgp_dist_link_probe", C.c_int, vp, i64, i32, pd)
    sig("lpgp_dist_init_host", C.c_int, vp, i32, i32, HOST_EXCHANGE_FN, vp)
    sig("lpgp_dist_ipc_export", C.c_int, vp, i64, C.c_char_p)
    sig("lpgp_dist_init_ipc", C.c_int, vp, i32, i32, C.c_char_p, HOST_EXCHANGE_FN, vp)
    sig("lpgp_pts_create", C.c_int, vp, pd, i64, i32, C.POINTER(vp))
    sig("lpgp_pts_destroy", C.c_int, vp)
    sig("lpgp_mat_create", C.c_int, vp, i64, C.POINTER(vp))
    sig("lpgp_mat_destroy", C.c_int, vp)
    sig("lpgp_mat_add_block", C.c_int, vp, vp, i64)
    sig("lpgp_mat_pop_block", C.c_int, vp, vp)
    sig("lpgp_mat_set_view", C.c_int, vp, vp, i32)
    sig("lpgp_mat_num_blocks", i32, vp)
    sig("lpgp_mat_num_blocks_total", i32, vp)
    sig("lpgp_mat_clone", C.c_int, vp, vp, i32, C.POINTER(vp))
    sig("lpgp_mat_size", i64, vp)
    sig("lpgp_mat_padded_size", i64, vp)
    sig("lpgp_gram_assemble", C.c_int, vp, pk, i32, vp, vp, vp, i32, i32)
    sig("lpgp_mat_add_diag", C.c_int, vp, vp, i32, pd, dbl)
    sig("lpgp_mat_add_dense", C.c_int, vp, vp, i32, pd)
    sig("lpgp_mat_to_host", C.c_int, vp, vp, i32, pd)
    sig("lpgp_mat_factor_diag", C.c_int, vp, vp, pd)
    sig("lpgp_potrf", C.c_int, vp, vp, C.POINTER(i32))
    sig("lpgp_potrf_enqueue", C.c_int, vp, vp)
    sig("lpgp_mat_check", C.c_int, vp, vp, C.POINTER(i32), C.POINTER(i32))
    sig("lpgp_mat_condition", C.c_int, vp, vp, i64, vp, C.POINTER(CondBlock), i32, dbl, pd, pd, i32, C.POINTER(i32))
    sig("lpgp_mat_truncate", C.c_int, vp, vp, i32)
    sig("lpgp_potrs", C.c_int, vp, vp, pd, i64)
    sig("lpgp_solve_weights", C.c_int, vp, vp, pd, pd)
    sig("lpgp_mat_set_residual", C.c_int, vp, vp, pd)
    sig("lpgp_rhs_create", C.c_int, vp, vp, i64, C.POINTER(vp))
    sig("lpgp_rhs_destroy", C.c_int, vp)
    sig("lpgp_cross_assemble", C.c_int, vp, pk, i32, vp, vp, vp, vp, i32)
    sig("lpgp_cross_assemble_row", C.c_int, vp, C.POINTER(CrossBlock), i32, vp, vp, vp)
    sig("lpgp_predict", C.c_int, vp, vp, vp, pd, pd, pd, pd)
    sig("lpgp_potrf_predict", C.c_int, vp, vp, vp, pd, pd, pd, pd)
    sig("lpgp_trsm_lower", C.c_int, vp, vp, vp)
    sig("lpgp_rhs_inner", C.c_int, vp, vp, vp, pd)
    i32, i64, dbl = C.c_int32, C.c_int64, C.c_double
    sig("lpgp_dvec_create", C.c_int, vp, i64, i64, C.POINTER(vp))
    sig("lpgp_dvec_destroy", C.c_int, vp)
    sig("lpgp_dvec_set", C.c_int, vp, vp, pd)
    sig("lpgp_dvec_get", C.c_int, vp, vp, pd)
    sig("lpgp_dvec_axpby", C.c_int, vp, vp, vp, vp, dbl)
    sig("lpgp_dvec_scale_rows_add", C.c_int, vp, vp, i64, vp, i64, pd, i64)
    sig("lpgp_kernel_matvec_dev", C.c_int, vp, pk, i32, vp, vp, vp, i64, vp, i64, i32)
    sig("lpgp_pcg_create", C.c_int, vp, i64, i64, i32, pd, pd, dbl, C.POINTER(vp))
    sig("lpgp_pcg_destroy", C.c_int, vp)
    sig("lpgp_pcg_start", C.c_int, vp, vp, vp, vp, vp, pd, dbl, pd)
    sig("lpgp_pcg_step", C.c_int, vp, vp, vp, vp, vp, vp, vp, dbl, pd)
    sig("lpgp_rhs_matmul", C.c_int, vp, vp, pd, C.c_int64, C.POINTER(vp))
    sig("lpgp_gemm_host", C.c_int, vp, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_double, pd, pd, C.c_double, pd)
    sig("lpgp_rhs_to_host", C.c_int, vp, vp, vp, pd)
    sig("lpgp_kernel_diag", C.c_int, vp, pk, i32, pd)
    sig("lpgp_kernel_matrix", C.c_int, vp, pk, i32, vp, vp, pd)
    sig("lpgp_gram_assemble_grid", C.c_int, vp, pk, i32, C.POINTER(vp), C.POINTER(vp), vp, i32, i32)
    sig("lpgp_kron_fits", C.c_int, pk, i32)
    sig("lpgp_kernel_matvec", C.c_int, vp, pk, i32, vp, vp, pd, i64, pd)
    sig("lpgp_profile_enable", C.c_int, vp, i32)
    sig("lpgp_profile_reset", C.c_int, vp)
    sig("lpgp_profile_get", C.c_int, vp, i32, pd, C.POINTER(i64), pd, pd)
    return lib


lib = _load()

EXPORTED = [
    "lpgp_init", "lpgp_finalize", "lpgp_last_error", "lpgp_device_info", "lpgp_sync",
    "lpgp_set_option", "lpgp_get_option", "lpgp_dist_unique_id", "lpgp_dist_init", "lpgp_dist_info", "lpgp_dist_init_host", "lpgp_dist_ipc_export", "lpgp_dist_init_ipc", "lpgp_dist_set_grid",
    "lpgp_dist_grid", "lpgp_dist_stats", "lpgp_dist_link_probe", "lpgp_pts_create", "lpgp_pts_destroy", "lpgp_mat_create",
    "lpgp_mat_destroy", "lpgp_mat_add_block", "lpgp_mat_pop_block", "lpgp_mat_set_view", "lpgp_mat_num_blocks",
    "lpgp_mat_num_blocks_total", "lpgp_mat_clone", "lpgp_mat_size", "lpgp_mat_padded_size",
    "lpgp_gram_assemble", "lpgp_mat_add_diag", "lpgp_mat_add_dense", "lpgp_mat_to_host", "lpgp_mat_factor_diag",
    "lpgp_potrf", "lpgp_potrf_enqueue", "lpgp_mat_condition", "lpgp_mat_check", "lpgp_mat_truncate", "lpgp_potrs", "lpgp_solve_weights", "lpgp_mat_set_residual", "lpgp_rhs_create", "lpgp_rhs_destroy",
    "lpgp_cross_assemble", "lpgp_cross_assemble_row", "lpgp_predict", "lpgp_potrf_predict", "lpgp_trsm_lower", "lpgp_rhs_inner", "lpgp_rhs_matmul", "lpgp_gemm_host",
    "lpgp_dvec_create", "lpgp_dvec_destroy", "lpgp_dvec_set", "lpgp_dvec_get", "lpgp_dvec_axpby", "lpgp_dvec_scale_rows_add", "lpgp_kernel_matvec_dev",
    "lpgp_pcg_create", "lpgp_pcg_destroy", "lpgp_pcg_start", "lpgp_pcg_step",
    "lpgp_rhs_to_host", "lpgp_kernel_diag", "lpgp_kernel_matrix", "lpgp_kernel_matvec", "lpgp_gram_assemble_grid", "lpgp_kron_fits", "lpgp_profile_enable", "lpgp_profile_reset",
    "lpgp_profile_get",
    ]


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib.lpgp_last_error().decode(errors="replace")
        raise LpgpError(f"{what} failed (rc={rc}): {msg}")


def as_pd(a: np.ndarray):
    assert a.dtype == np.double and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


def make_kdesc_array(groups) -> "C.Array[KDesc]":
    """groups: list of dicts {d, family[], p[], lengthscale[], scale, terms[(coef, n0[], n1[])]}."""
    if not 1 <= len(groups) <= MAXG:
        raise ValueError(f"between 1 and {MAXG} summands supported, got {len(groups)}")
    arr = (KDesc * len(groups))()
    for kd, g in zip(arr, groups):
        d = int(g["d"])
        if not 1 <= d <= MAXD:
            raise ValueError(f"input dimension {d} not supported (max {MAXD})")
        if not 1 <= len(g["terms"]) <= MAXT:
            raise ValueError(f"{len(g['terms'])} terms not supported (max {MAXT})")
        kd.d = d
        for j in range(d):
            kd.family[j] = int(g["family"][j])
            kd.p[j] = int(g["p"][j])
            kd.lengthscale[j] = float(g["lengthscale"][j])
        kd.scale = float(g["scale"])
        kd.nterms = len(g["terms"])
        for t, (coef, n0, n1) in enumerate(g["terms"]):
            kd.terms[t].coef = float(coef)
            for j in range(d):
                kd.terms[t].n0[j] = int(n0[j])
                kd.terms[t].n1[j] = int(n1[j])
    return arr
