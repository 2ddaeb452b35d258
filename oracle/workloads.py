"""Run a synthetic workload (`linpde_gp_amd.problems.Workload`-shaped data) through the CPU
oracle, phase by phase -- used by `bench.py`'s `cpu_baseline` leg, `smoke()` and the parity
tests.  Test infrastructure only; see `oracle/__init__.py`.

Phases follow BASELINE.md §3: (1) Gram assembly with vectorised NumPy, (2)
`scipy.linalg.cholesky(lower=True)` (LAPACK dpotrf), (3) `cho_solve` for the weights,
(4) cross-covariance assembly, (5) mean GEMV, (6) marginal variance via
`solve_triangular` + column norms.
"""

from __future__ import annotations

import time

import numpy as np
import scipy.linalg

from . import covfuncs, gp


def blocks_of(wl) -> list:
    return [gp.ObsBlock(o.X, o.op, o.Y, None, o.noise_var) for o in wl.observations]


def run(wl, want_var: bool = True, want_cond: bool = False) -> dict:
    """Returns mean, var and per-phase seconds (and, untimed, a 2-norm condition estimate of the Gram matrix)."""
    blocks = blocks_of(wl)
    t = {}
    t0 = time.perf_counter()
    G = gp.gram(wl.kernel, blocks)
    t["assemble"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    chol = scipy.linalg.cholesky(G, lower=True, overwrite_a=False, check_finite=False)
    t["potrf"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    w = scipy.linalg.cho_solve((chol, True), gp.residual(blocks), check_finite=False)
    t["weights"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    K = gp.cross_cov(wl.kernel, blocks, wl.Xtest)
    t["crosscov"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    mean = K @ w
    t["mean"] = time.perf_counter() - t0
    var = None
    if want_var:
        t0 = time.perf_counter()
        V = scipy.linalg.solve_triangular(chol, K.T, lower=True, check_finite=False)
        d = wl.Xtest.shape[1]
        ident = covfuncs.identity(d)
        var = covfuncs.k_diag(wl.kernel, ident, ident, wl.Xtest) - gp.colsumsq(V)
        t["var"] = time.perf_counter() - t0
    t["total"] = sum(t.values())
    out = {"mean": mean, "var": var, "seconds": t, "weights": w}
    if want_cond:
        out["cond2"] = gp.cond2_estimate(G, chol)
    return out
