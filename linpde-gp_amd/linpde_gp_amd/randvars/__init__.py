"""Minimal `Normal` / `Constant` random variables (probnum `randvars` protocol: `.mean`,
`.cov`, `.var`, `.std`, `.shape`), used for observation noise `b` and as the result type
of `GaussianProcess.__call__`."""

from __future__ import annotations

import numpy as np


class Normal:
    def __init__(self, mean, cov):
        self._mean = np.asarray(mean, dtype=np.double)
        cov = np.asarray(cov, dtype=np.double)
        n = self._mean.size
        if cov.ndim == 0 and self._mean.ndim > 0:
            cov = float(cov) * np.eye(n)
        if self._mean.ndim == 0:
            cov = cov.reshape(())
        elif cov.shape != (n, n):
            raise ValueError(f"covariance has shape {cov.shape}, expected {(n, n)}")
        self._cov = cov

    @property
    def mean(self):
        return self._mean

    @property
    def cov(self):
        return self._cov

    @property
    def var(self):
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
