"""Minimal `Normal` / `Constant` random variables (probnum `randvars` protocol: `.mean`,
`.cov`, `.var`, `.std`, `.shape`), used for observation noise `b` and as the result type
of `GaussianProcess.__call__`."""

from __future__ import annotations

import numpy as np


class Normal:
    """Multivariate normal.  `cov` may be dense (n x n), a scalar (sigma^2 I) or a vector of
    variances (diagonal covariance, kept as a vector: a noisy block of 10^4+ observations
    must not materialise n^2 zeros)."""

    def __init__(self, mean, cov):
        self._mean = np.asarray(mean, dtype=np.double)
        cov = np.asarray(cov, dtype=np.double)
        n = self._mean.size
        self._cov_diag = None
        self._cov = None
        if self._mean.ndim == 0:
            self._cov = cov.reshape(())
        elif cov.ndim == 0:
            self._cov_diag = np.full(n, float(cov))
        elif cov.ndim == 1:
            if cov.shape != (n,):
                raise ValueError(f"variance vector has shape {cov.shape}, expected {(n,)}")
            self._cov_diag = cov.copy()
        elif cov.shape == (n, n):
            self._cov = cov
        else:
            raise ValueError(f"covariance has shape {cov.shape}, expected {(n, n)}")

    @property
    def mean(self):
        return self._mean

    @property
    def cov(self):
        if self._cov is None:
            self._cov = np.diag(self._cov_diag)
        return self._cov

    @property
    def cov_diag(self):
        """Variances if the covariance is (known to be) diagonal, else None."""
        return self._cov_diag

    @property
    def var(self):
        if self._cov_diag is not None:
            return self._cov_diag.reshape(self._mean.shape)
        if self._cov.ndim == 0:
            return self._cov
        return np.diag(self._cov).reshape(self._mean.shape)

    @property
    def std(self):
        return np.sqrt(np.maximum(self.var, 0.0))

    @property
    def shape(self):
        return self._mean.shape

    @property
    def size(self):
        return self._mean.size


class Constant:
    def __init__(self, support):
        self._support = np.asarray(support, dtype=np.double)

    @property
    def mean(self):
        return self._support

    @property
    def support(self):
        return self._support

    @property
    def cov(self):
        n = self._support.size
        return np.zeros((n, n))

    @property
    def shape(self):
        return self._support.shape


def asrandvar(b):
    if isinstance(b, (Normal, Constant)):
        return b
    if np.ndim(b) >= 0 and not hasattr(b, "mean"):
        return Constant(b)
    raise TypeError(f"`b` must be a `Normal` or a `Constant` `RandomVariable` ({type(b)=})")


def condition_normal_on_observations(prior: Normal, observations, noise: Normal | None = None, transform=None) -> Normal:
    r"""Finite-dimensional Gaussian conditioning (`randvars/_normal.py:8-71` of the reference):
    observe `y = A x + eps`, `x ~ N(mu0, Sigma0)`, `eps ~ N(b, Lambda)`.

    The Gram matrix `A Sigma0 A^T + Lambda` is factored on the device (`lpgp_potrf`) and the
    gain `Gram^{-1} (A Sigma0)` comes from ONE multi-right-hand-side solve (`lpgp_potrs`) -- the
    reference calls LAPACK `cho_solve` on the host.  Also bound as
    `Normal.condition_on_observations`."""
    from .. import _engine
    from ..randprocs import covfuncs

    observations = np.asarray(observations, dtype=np.double)
    A = None if transform is None else np.asarray(transform, dtype=np.double)
    if A is not None and A.ndim == 1:
        A = A[None, :]
        observations = observations.reshape(1)
        if noise is not None:
            noise = Normal(np.asarray(noise.mean).reshape(1), np.asarray(noise.cov).reshape(1, 1))
    mu0 = np.asarray(prior.mean, dtype=np.double).reshape(-1)
    S0 = np.asarray(prior.cov, dtype=np.double).reshape(mu0.size, mu0.size)
    ctx = _engine.default_context()
    # (the three dense products of the update run on the device's MFMA kernel since round 6 -- `lpgp_gemm_host`; NumPy until then)
    crosscov = S0 if A is None else _engine.gemm(ctx, A, S0)     # Cov(y, x), (n_obs, n)
    pred_mean = mu0 if A is None else A @ mu0
    pred_cov = S0 if A is None else _engine.gemm(ctx, crosscov, A, transb=True)
    if noise is not None:
        pred_mean = pred_mean + np.asarray(noise.mean, dtype=np.double).reshape(-1)
        pred_cov = pred_cov + np.asarray(noise.cov, dtype=np.double).reshape(pred_cov.shape)
    n_obs = pred_mean.size
    if observations.reshape(-1).size != n_obs:
        raise ValueError(f"expected {n_obs} observations, got shape {observations.shape}")
    mat = _engine.GramMatrix(ctx, n_obs)
    bi = mat.add_block(n_obs)
    pts = _engine.Points(ctx, np.zeros((n_obs, 1)))
    mat.assemble(covfuncs.Zero(()).lower(), pts, None, bi, bi)   # clear the block, then add the dense Gram
    mat.add_dense(bi, np.ascontiguousarray(pred_cov))
    info = mat.potrf()
    if info != 0:
        raise np.linalg.LinAlgError(f"{info}-th leading minor of the predictive covariance is not positive definite")
    # [gain^T | weights] = Gram^{-1} [crosscov | y - pred_mean] in one solve
    rhs = np.concatenate([crosscov, (observations.reshape(-1) - pred_mean)[:, None]], axis=1)
    sol = mat.potrs(rhs)
    gain_t, w = sol[:, :-1], sol[:, -1]
    return Normal(mean=(mu0 + crosscov.T @ w).reshape(np.shape(prior.mean)),
                  cov=_engine.gemm(ctx, crosscov, gain_t, transa=True, alpha=-1.0, beta=1.0, C=S0))


Normal.condition_on_observations = condition_normal_on_observations


