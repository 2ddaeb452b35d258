"""Covariance between two (array-shaped) random variables in both of its representations: the array of
shape `shape0 + shape1` and the `(size0, size1)` matrix (C-order flattening of either side).

Host mirror of `randvars/_covariance.py:13-224` of the reference (`Covariance`, `ArrayCovariance`,
`LinearOperatorCovariance`): same constructor arguments, properties (`shape0/1`, `ndim0/1`, `size0/1`,
`array`, `linop`, `matrix`, `flatten0/1`), arithmetic and error behaviour.  `LinearOperatorCovariance`
is what applying two point-evaluation functionals to a covariance function returns
(`crosscov/linfunctls/_evaluation.py:163-173`); here its operator is the device-resident
`KernelLinearOperator` (matrix-free products by `lpgp_kernel_matvec`, `todense()` by
`lpgp_kernel_matrix`) or the resident Cholesky factorisation of a posterior's Gram matrix -- thin views,
no arithmetic of their own.
"""

from __future__ import annotations

import numpy as np


def _as_shape(shape) -> tuple:
    if np.ndim(shape) == 0:
        return (int(shape),)
    return tuple(int(s) for s in shape)


class _DenseOperator:
    """`pn.linops.aslinop(ndarray)` stand-in: the `@` / `.T` / `todense` / `shape` subset."""

    def __init__(self, A: np.ndarray):
        self._A = np.asarray(A, dtype=np.double)
        if self._A.ndim != 2:
            raise ValueError("a linear operator is a matrix")
        self.shape = self._A.shape
        self.dtype = self._A.dtype

    def __matmul__(self, V):
        return self._A @ np.asarray(V, dtype=np.double)

    matmul = __matmul__

    @property
    def T(self):
        return _DenseOperator(self._A.T)

    def todense(self, cache: bool = True):
        return self._A


def aslinop(A):
    """ndarray -> dense operator; anything with `shape` and `todense` is taken as it is."""
    if hasattr(A, "todense") and hasattr(A, "shape"):
        return A
    return _DenseOperator(np.asarray(A, dtype=np.double))


class Covariance:
    def __init__(self, shape0, shape1) -> None:
        self._shape0 = _as_shape(shape0)
        self._shape1 = _as_shape(shape1)

    @property
    def shape0(self):
        return self._shape0

    @property
    def ndim0(self) -> int:
        return len(self._shape0)

    @property
    def size0(self) -> int:
        return int(np.prod(self._shape0, dtype=int))

    @property
    def shape1(self):
        return self._shape1

    @property
    def ndim1(self) -> int:
        return len(self._shape1)

    @property
    def size1(self) -> int:
        return int(np.prod(self._shape1, dtype=int))

    @property
    def array(self) -> np.ndarray:
        raise NotImplementedError

    @property
    def linop(self):
        raise NotImplementedError

    @property
    def matrix(self) -> np.ndarray:
        raise NotImplementedError

    def flatten0(self, event0, /) -> np.ndarray:
        event0 = np.asarray(event0)
        if event0.shape != self.shape0:
            raise ValueError(f"The shape of the event must be the same as `shape0`, but {event0.shape} != {self.shape0}.")
        return np.reshape(event0, (-1,), order="C")

    def flatten1(self, event1, /) -> np.ndarray:
        event1 = np.asarray(event1)
        if event1.shape != self.shape1:
            raise ValueError(f"The shape of the event must be the same as `shape1`, but {event1.shape} != {self.shape1}.")
        return np.reshape(event1, (-1,), order="C")


class ArrayCovariance(Covariance):
    @staticmethod
    def from_scalar(var) -> "ArrayCovariance":
        return ArrayCovariance(np.asarray(var, dtype=np.double), shape0=(), shape1=())

    def __init__(self, cov_array, shape0, shape1) -> None:
        super().__init__(shape0, shape1)
        self._cov_array = np.asarray(cov_array, dtype=np.double)
        if self._cov_array.shape != self.shape0 + self.shape1:
            raise ValueError(
                f"The shape of `cov_array` must be `shape0 + shape1`, but `{self._cov_array.shape} != "
                f"{self.shape0} + {self.shape1}`.")

    @property
    def array(self) -> np.ndarray:
        return self._cov_array

    @property
    def linop(self):
        return aslinop(self.matrix)

    @property
    def matrix(self) -> np.ndarray:
        return np.reshape(self._cov_array, (self.size0, self.size1), order="C")

    def __neg__(self):
        return -1.0 * self

    def __add__(self, other):
        if isinstance(other, ArrayCovariance) and self.shape0 == other.shape0 and self.shape1 == other.shape1:
            return ArrayCovariance(self.array + other.array, self.shape0, self.shape1)
        if isinstance(other, LinearOperatorCovariance):
            return other + self
        return NotImplemented

    def __sub__(self, other):
        return self + (-other)

    def __rmul__(self, other):
        if np.ndim(other) == 0:
            return ArrayCovariance(other * self.array, self.shape0, self.shape1)
        return NotImplemented


class LinearOperatorCovariance(Covariance):
    def __init__(self, cov_linop, shape0, shape1) -> None:
        super().__init__(shape0, shape1)
        self._cov_linop = aslinop(cov_linop)
        if tuple(self._cov_linop.shape) != (self.size0, self.size1):
            raise ValueError(
                f"The shape of `cov_linop` must be `(size0, size1)`, but `{tuple(self._cov_linop.shape)} != "
                f"({self.size0}, {self.size1})`.")
        self._dense = None

    @property
    def array(self) -> np.ndarray:
        return np.reshape(self.matrix, self.shape0 + self.shape1, order="C")

    @property
    def linop(self):
        return self._cov_linop

    @property
    def matrix(self) -> np.ndarray:
        if self._dense is None:                    # `todense(cache=True)` of the reference
            self._dense = np.asarray(self._cov_linop.todense())
        return self._dense

    def __neg__(self):
        return -1.0 * self

    def __add__(self, other):
        if isinstance(other, Covariance) and self.shape0 == other.shape0 and self.shape1 == other.shape1:
            return ArrayCovariance(self.array + other.array, self.shape0, self.shape1)
        return NotImplemented

    __radd__ = __add__

    def __sub__(self, other):
        return self + (-other)

    def __rmul__(self, other):
        if np.ndim(other) == 0:
            return ArrayCovariance(other * self.array, self.shape0, self.shape1)
        return NotImplemented
