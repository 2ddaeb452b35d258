"""Dense NumPy/SciPy GP conditioning (oracle; test-only).

Restates, on the reference's own CPU libraries (LAPACK dpotrf/dpotrs/dtrtrs through
`scipy.linalg`):

* the naive end-to-end GP of `tests/linpde_gp/randprocs/test_posterior_gp.py:182-221`
  (`condition_gp_on_observations`), generalised to observation blocks
  `L_i[f](X_i) + b_i` as built by `_preprocess_observations`
  (`randprocs/_gaussian_process/_conditional.py:296-399`);
* the iterative update `ConditionalGaussianProcess.condition_on_observations`
  (`_conditional.py:253-294`) with `BlockMatrix2x2.schur_update` / `.schur` /
  `._cholesky` (`linops/_block.py:192-242`);
* prediction `Mean._evaluate` (`_conditional.py:193-197`) and
  `CovarianceFunction._evaluate` (`:223-231`).

An observation block is `ObsBlock(X, L, Y, noise_mean, noise_cov)`; `L` is a
`{multi_index: coeff}` dict (see `oracle.covfuncs`), `noise_cov` is None, a scalar
variance, a vector of variances, or a dense matrix.
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import scipy.linalg

from . import covfuncs


@dataclass
class ObsBlock:
    X: np.ndarray
    L: dict
    Y: np.ndarray
    noise_mean: np.ndarray | float | None = None
    noise_cov: np.ndarray | float | None = None

    @property
    def n(self) -> int:
        return int(np.asarray(self.X).shape[0])


def _noise_cov_dense(block: ObsBlock) -> np.ndarray | None:
    if block.noise_cov is None:
        return None
    c = np.asarray(block.noise_cov, dtype=np.double)
    if c.ndim == 0:
        return float(c) * np.eye(block.n)
    if c.ndim == 1:
        return np.diag(c)
    return c


def colsumsq(V: np.ndarray) -> np.ndarray:
    """Column sums of squares, accumulated along the CONTIGUOUS axis so that numpy's pairwise summation applies.

    `np.sum(V * V, axis=0)` / `einsum("ij,ij->j")` on a C-ordered V add the N rows one after another; when the posterior
    variance is 1e-6 of the prior variance (c5) that naive N-term accumulation alone is 1.6e-13 = 2e-8 of the variance away
    from the long-double sum (measured at full c5, `profiles/r02_c5_full_parity.txt`) -- above the 1e-8 parity bar and 25x
    the device's distance from the long-double sum.  The checker must not be the noisiest part of the check."""
    Vt = np.ascontiguousarray(V.T)
    return np.sum(Vt * Vt, axis=1)


def cond2_estimate(G: np.ndarray, chol: np.ndarray, iters: int = 25, seed: int = 24) -> float:
    """2-norm condition estimate of the SPD Gram matrix ("report cond-estimate of G"): largest eigenvalue by
    power iteration on G, smallest by inverse iteration through the Cholesky factor; Rayleigh quotients, so the estimate
    is a LOWER bound that converges from below."""
    rng = np.random.default_rng(seed)
    v = rng.standard_normal(G.shape[0]); v /= np.linalg.norm(v)
    u = v.copy()
    lmax = lmin_inv = 0.0
    for _ in range(iters):
        y = G @ v
        lmax = float(v @ y)
        v = y / np.linalg.norm(y)
        z = scipy.linalg.cho_solve((chol, True), u, check_finite=False)
        lmin_inv = float(u @ z)
        u = z / np.linalg.norm(z)
    return lmax * lmin_inv


def prior_mean_L(mean_const: float, L: dict, n: int) -> np.ndarray:
    """L[m](X) for a constant prior mean: only the order-0 coefficient survives."""
    d = len(next(iter(L)))
    return np.full(n, mean_const * L.get((0,) * d, 0.0))


def gram(kernel, blocks: list[ObsBlock]) -> np.ndarray:
    """Dense Gram `[L_i k L_j'](X_i, X_j) + blockdiag(noise)`  (`_conditional.py:357-394`)."""
    rows = []
    for bi in blocks:
        rows.append([covfuncs.LkL(kernel, bi.L, bj.L, bi.X, bj.X) for bj in blocks])
    G = np.block(rows)
    off = 0
    for b in blocks:
        nc = _noise_cov_dense(b)
        if nc is not None:
            G[off:off + b.n, off:off + b.n] += nc
        off += b.n
    return G


def residual(blocks: list[ObsBlock], mean_const: float = 0.0) -> np.ndarray:
    """Y - L[m] - b.mean, concatenated (`_conditional.py:99-107`)."""
    parts = []
    for b in blocks:
        r = np.asarray(b.Y, dtype=np.double).reshape(-1) - prior_mean_L(mean_const, b.L, b.n)
        if b.noise_mean is not None:
            r = r - np.broadcast_to(np.asarray(b.noise_mean, dtype=np.double), (b.n,))
        parts.append(r)
    return np.concatenate(parts)


def cross_cov(kernel, blocks: list[ObsBlock], Xtest: np.ndarray, Ltest: dict | None = None) -> np.ndarray:
    """K_xX = [(Ltest k L_j')(x, X_j)]_j, shape (M, N_tot)  (`_conditional.py:140-153`)."""
    d = np.asarray(Xtest).shape[1]
    L0 = covfuncs.identity(d) if Ltest is None else Ltest
    return np.concatenate([covfuncs.LkL(kernel, L0, b.L, Xtest, b.X) for b in blocks], axis=1)


@dataclass
class Posterior:
    kernel: list
    blocks: list
    mean_const: float
    G: np.ndarray
    chol: np.ndarray          # lower Cholesky factor of G
    weights: np.ndarray       # representer weights G^{-1}(Y - Lm - b.mean)

    def mean(self, Xtest, Ltest: dict | None = None) -> np.ndarray:
        K = cross_cov(self.kernel, self.blocks, Xtest, Ltest)
        d = np.asarray(Xtest).shape[1]
        L0 = covfuncs.identity(d) if Ltest is None else Ltest
        return prior_mean_L(self.mean_const, L0, K.shape[0]) + K @ self.weights

    def var(self, Xtest, Ltest: dict | None = None) -> np.ndarray:
        """Marginal variance: k_xx - || L^{-1} K_Xx ||^2 column-wise."""
        d = np.asarray(Xtest).shape[1]
        L0 = covfuncs.identity(d) if Ltest is None else Ltest
        K = cross_cov(self.kernel, self.blocks, Xtest, Ltest)
        V = scipy.linalg.solve_triangular(self.chol, K.T, lower=True)
        return covfuncs.k_diag(self.kernel, L0, L0, Xtest) - colsumsq(V)

    def cov(self, X0, X1=None, Ltest: dict | None = None) -> np.ndarray:
        """Full posterior covariance (`_conditional.py:223-231`, via `cho_solve`)."""
        X1 = X0 if X1 is None else X1
        d = np.asarray(X0).shape[1]
        L0 = covfuncs.identity(d) if Ltest is None else Ltest
        K0 = cross_cov(self.kernel, self.blocks, X0, Ltest)
        K1 = cross_cov(self.kernel, self.blocks, X1, Ltest)
        kxx = covfuncs.LkL(self.kernel, L0, L0, X0, X1)
        return kxx - K0 @ scipy.linalg.cho_solve((self.chol, True), K1.T)


def refined_posterior(post: "Posterior", Xtest, Ltest: dict | None = None, iters: int = 12, K: np.ndarray | None = None):
    """Posterior mean and marginal variance of the fp64 Gram matrix to (nearly) working accuracy: every solve with `G` is
    refined with long-double residuals until it stops moving (LAPACK's factor as the preconditioner).  This is the yardstick
    for ill-conditioned problems, where LAPACK itself is cond(G) x eps away from the exact answer and a comparison of two fp64
    implementations at 1e-8 says nothing (`tests/test_gpu_random.py`, `scratch/random_diag.py`)."""
    d = np.asarray(Xtest).shape[1]
    L0 = covfuncs.identity(d) if Ltest is None else Ltest
    Gl = post.G.astype(np.longdouble)

    def solve(B):
        B = np.asarray(B, dtype=np.double)
        X = scipy.linalg.cho_solve((post.chol, True), B).astype(np.longdouble)
        for _ in range(iters):
            R = (B.astype(np.longdouble) - Gl @ X).astype(np.double)
            dX = scipy.linalg.cho_solve((post.chol, True), R)
            X = X + dX
            if np.max(np.abs(dX)) <= 1e-18 * np.max(np.abs(X)):
                break
        return X

    if K is None:                                                         # (a caller that refines with ITS OWN evaluation of the
        K = cross_cov(post.kernel, post.blocks, Xtest, Ltest)             #  matrices passes `post.G` and `K` from there)  (M, N)
    w = solve(residual(post.blocks, post.mean_const))
    mean = prior_mean_L(post.mean_const, L0, K.shape[0]).astype(np.longdouble) + K.astype(np.longdouble) @ w
    W = solve(K.T)                                                        # (N, M)
    var = covfuncs.k_diag(post.kernel, L0, L0, Xtest).astype(np.longdouble) - np.sum(K.T.astype(np.longdouble) * W, axis=0)
    return np.asarray(mean, dtype=np.double), np.asarray(var, dtype=np.double)


def condition(kernel, blocks: list[ObsBlock], mean_const: float = 0.0) -> Posterior:
    """One-shot dense conditioning (`test_posterior_gp.py:182-197`)."""
    G = gram(kernel, blocks)
    chol = scipy.linalg.cholesky(G, lower=True)
    w = scipy.linalg.cho_solve((chol, True), residual(blocks, mean_const))
    return Posterior(kernel, list(blocks), mean_const, G, chol, w)


def condition_iteratively(kernel, blocks: list[ObsBlock], mean_const: float = 0.0) -> Posterior:
    """Block-by-block conditioning with Schur-complement updates.

    `_conditional.py:253-294` + `linops/_block.py:192-242`:
        L_A_inv_B = L_A^{-1} B;  S = D - (L_A_inv_B)^T (L_A_inv_B)
        y = S^{-1}(v - C A^{-1}u);  x = A^{-1}u - A^{-1} B y
        chol([[A,B],[C,D]]) = [[L_A, 0], [(L_A_inv_B)^T, chol(S)]]
    """
    post = condition(kernel, blocks[:1], mean_const)
    for k in range(1, len(blocks)):
        prev, new = blocks[:k], blocks[k]
        # lower-left block  C = L_new(kLas_prev), `_conditional.py:270`
        C = np.concatenate(
            [covfuncs.LkL(kernel, new.L, b.L, new.X, b.X) for b in prev], axis=1
        )
        D = gram(kernel, [new])
        v = residual([new], mean_const)
        L_A = post.chol
        L_A_inv_B = scipy.linalg.solve_triangular(L_A, C.T, lower=True)
        S = D - L_A_inv_B.T @ L_A_inv_B
        L_S = scipy.linalg.cholesky(S, lower=True)
        A_inv_u = post.weights
        y = scipy.linalg.cho_solve((L_S, True), v - C @ A_inv_u)
        x = A_inv_u - scipy.linalg.cho_solve((L_A, True), C.T @ y)
        n0, n1 = L_A.shape[0], L_S.shape[0]
        chol = np.zeros((n0 + n1, n0 + n1))
        chol[:n0, :n0] = L_A
        chol[n0:, :n0] = L_A_inv_B.T
        chol[n0:, n0:] = L_S
        G = np.block([[post.G, C.T], [C, D]])
        post = Posterior(kernel, list(blocks[:k + 1]), mean_const, G, chol,
                         np.concatenate((x, y)))
    return post
