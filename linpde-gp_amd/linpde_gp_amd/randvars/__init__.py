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


__all__ = ["Normal", "Constant", "asrandvar"]
