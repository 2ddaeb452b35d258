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


def host_memory_needed(wl, chunk: int = 2048, workers: int = 1) -> float:
    """Bytes of host memory `run_in_place` peaks at: the Gram matrix / factor (one N x N array), K and V (two N x M
    arrays), and per worker the temporaries of one row chunk of the vectorised assembly (~12 chunk-sized arrays)."""
    n, m = float(wl.n_total), float(wl.Xtest.shape[0])
    return 8.0 * n * n + 2.0 * 8.0 * n * m + workers * 12.0 * 8.0 * chunk * n + 2e9


def run_in_place(wl, chunk: int = 2048, workers: int = 1) -> dict:
    """`run` for the LARGE configurations (c4: N_tot = 66 560 -> 35.4 GB Gram matrix; c5: 33 600): the same algorithm on
    the same libraries with ONE matrix-sized array in memory instead of ~12 -- the Gram matrix is assembled in row chunks
    of its lower triangle (the formulas of `gp.gram`, block pair by block pair; LAPACK only reads the lower triangle),
    factored in place (`overwrite_a`), and the triangular solve overwrites K^T.  `workers` > 1: the row chunks are
    evaluated by that many threads (NumPy releases the GIL inside its loops; same arithmetic per entry, so the result does
    not depend on it).  `tests/test_oracle_golden.py` pins it to `run` at a size both can do."""
    from concurrent.futures import ThreadPoolExecutor
    blocks = blocks_of(wl)
    n, m = wl.n_total, wl.Xtest.shape[0]
    off = np.cumsum([0] + [b.n for b in blocks])
    t = {}
    t0 = time.perf_counter()
    G = np.zeros((n, n), order="F")            # column-major: what LAPACK factors without a copy
    jobs = []
    for i, bi in enumerate(blocks):
        for j, bj in enumerate(blocks[:i + 1]):
            for r0 in range(0, bi.n, chunk):
                jobs.append((i, j, r0))

    def fill(job):
        i, j, r0 = job
        bi, bj = blocks[i], blocks[j]
        r1 = min(bi.n, r0 + chunk)
        c1 = bj.n if i != j else r1                  # lower triangle of a diagonal block only
        G[off[i] + r0:off[i] + r1, off[j]:off[j] + c1] = covfuncs.LkL(wl.kernel, bi.L, bj.L, bi.X[r0:r1], bj.X[:c1])

    def each(fn, items):
        if workers > 1:
            with ThreadPoolExecutor(workers) as ex:
                list(ex.map(fn, items))
        else:
            for it in items:
                fn(it)

    each(fill, jobs)
    for i, bi in enumerate(blocks):
        nc = bi.noise_cov
        if nc is not None:
            c = np.asarray(nc, dtype=np.double)
            idx = np.arange(off[i], off[i + 1])
            if c.ndim <= 1:
                G[idx, idx] += c
            else:                                    # dense noise: lower triangle is enough
                G[off[i]:off[i + 1], off[i]:off[i + 1]] += np.tril(c)
    t["assemble"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    chol = scipy.linalg.cholesky(G, lower=True, overwrite_a=True, check_finite=False)
    del G
    t["potrf"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    w = scipy.linalg.cho_solve((chol, True), gp.residual(blocks), check_finite=False)
    t["weights"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    Kt = np.empty((n, m), order="F")           # K^T, column-major: column j = cross-covariances of prediction point j

    def fill_k(r0):
        Kt[:, r0:r0 + chunk] = gp.cross_cov(wl.kernel, blocks, wl.Xtest[r0:r0 + chunk]).T

    each(fill_k, list(range(0, m, chunk)))
    t["crosscov"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    mean = w @ Kt
    t["mean"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    V = scipy.linalg.solve_triangular(chol, Kt, lower=True, overwrite_b=True, check_finite=False)
    d = wl.Xtest.shape[1]
    ident = covfuncs.identity(d)
    # columns of a column-major V are contiguous: numpy's pairwise summation applies (see gp.colsumsq)
    var = covfuncs.k_diag(wl.kernel, ident, ident, wl.Xtest) - np.array([float(np.sum(V[:, j] * V[:, j])) for j in range(m)])
    t["var"] = time.perf_counter() - t0
    t["total"] = sum(t.values())
    return {"mean": mean, "var": var, "seconds": t, "weights": w}
