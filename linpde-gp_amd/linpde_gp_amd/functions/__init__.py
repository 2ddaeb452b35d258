"""Deterministic functions used as prior means / right-hand sides.

Host-side mirror of `linpde_gp.functions.{Zero, Constant}`
(`functions/_constant.py:12-75` of the reference); only what the GP-posterior hot path
touches.  A function maps arrays of shape `batch + input_shape` to `batch + output_shape`.
"""

from __future__ import annotations

import numpy as np


def _as_shape(shape) -> tuple[int, ...]:
    if isinstance(shape, (int, np.integer)):
        return (int(shape),)
    return tuple(int(s) for s in shape)


class Function:
    def __init__(self, input_shape=(), output_shape=()):
        self._input_shape = _as_shape(input_shape)
        self._output_shape = _as_shape(output_shape)

    @property
    def input_shape(self):
        return self._input_shape

    @property
    def input_ndim(self):
        return len(self._input_shape)

    @property
    def output_shape(self):
        return self._output_shape

    @property
    def output_ndim(self):
        return len(self._output_shape)

    def __call__(self, x):
        x = np.asarray(x, dtype=np.double)
        if self.input_ndim and x.shape[x.ndim - self.input_ndim:] != self._input_shape:
            raise ValueError(
                f"The shape of the input {x.shape} is not compatible with the "
                f"input shape {self._input_shape} of the function."
            )
        return self._evaluate(x)

    def _evaluate(self, x):
        raise NotImplementedError


class Constant(Function):
    def __init__(self, input_shape, value):
        self._value = np.asarray(value, dtype=np.double)
        super().__init__(input_shape, output_shape=self._value.shape)

    @property
    def value(self):
        return self._value

    def _evaluate(self, x):
        batch_shape = x.shape[: x.ndim - self.input_ndim]
        return np.broadcast_to(self._value, batch_shape + self.output_shape).copy()


class Zero(Constant):
    def __init__(self, input_shape, output_shape=()):
        super().__init__(input_shape, value=np.zeros(_as_shape(output_shape)))


class LambdaFunction(Function):
    """Wraps a vectorised callable (right-hand sides, boundary values)."""

    def __init__(self, fn, input_shape, output_shape=()):
        super().__init__(input_shape, output_shape)
        self._fn = fn

    def _evaluate(self, x):
        return np.asarray(self._fn(x), dtype=np.double)
