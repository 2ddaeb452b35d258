"""Linear differential operators (API surface of `linpde_gp.linfuncops.diffops`).

Mirrors the public classes of the reference
  `_lindiffop.py:24-164`  LinearDifferentialOperator (coefficient-table representation)
  `_partial_derivative.py:17,131`  PartialDerivative / TimeDerivative
  `_derivative.py:11`  Derivative,  `_directional_derivative.py:15`  DirectionalDerivative
  `_laplacian.py:22,77,102`  WeightedLaplacian / Laplacian / SpatialLaplacian
  `_heat.py:14-39`  HeatOperator  (= TimeDerivative + WeightedLaplacian(-alpha on space))
  `_arithmetic.py:10-62`  ScaledLinearDifferentialOperator
No arithmetic happens here: an operator is only its coefficient table; the closed-form
differentiated kernels are evaluated by the HIP kernel from that table.  Where the
reference would fall back to JAX autodiff this raises NotImplementedError.
"""

from __future__ import annotations

import numpy as np

from .._linfuncop import LinearFunctionOperator, SumLinearFunctionOperator
from ._coefficients import MultiIndex, PartialDerivativeCoefficients


def _as_shape(shape):
    if isinstance(shape, (int, np.integer)):
        return (int(shape),)
    return tuple(int(s) for s in shape)


class LinearDifferentialOperator(LinearFunctionOperator):
    """Linear differential operator mapping to functions with codomain R."""

    def __init__(self, coefficients: PartialDerivativeCoefficients, input_shapes):
        input_shapes = (_as_shape(input_shapes[0]), _as_shape(input_shapes[1]))
        if coefficients.input_domain_shape != input_shapes[0]:
            raise ValueError("coefficients do not match the input domain shape")
        if coefficients.input_codomain_shape != input_shapes[1]:
            raise ValueError("coefficients do not match the input codomain shape")
        super().__init__(input_shapes=input_shapes, output_shapes=(input_shapes[0], ()))
        self._coefficients = coefficients

    @property
    def coefficients(self) -> PartialDerivativeCoefficients:
        return self._coefficients

    @property
    def has_mixed(self) -> bool:
        return self._coefficients.has_mixed

    def coefficients_dict(self):
        if self.input_codomain_shape != ():
            raise NotImplementedError("multi-output priors are out of scope of the MI355X path")
        if len(self.input_domain_shape) > 1:
            raise NotImplementedError("matrix-shaped inputs are not supported")
        out: dict = {}
        for mi, c in self._coefficients[()].items():
            key = mi.as_tuple() if mi.shape != () else (int(mi.array),)
            out[key] = out.get(key, 0.0) + float(c)
        return out

    def __rmul__(self, other):
        if np.ndim(other) == 0:
            return ScaledLinearDifferentialOperator(self, scalar=other)
        return NotImplemented

    def __add__(self, other):
        if isinstance(other, LinearDifferentialOperator):
            return LinearDifferentialOperator(self.coefficients + other.coefficients, self.input_shapes)
        return super().__add__(other)

    def __sub__(self, other):
        if isinstance(other, LinearDifferentialOperator):
            return LinearDifferentialOperator(self.coefficients - other.coefficients, self.input_shapes)
        return super().__sub__(other)


class ScaledLinearDifferentialOperator(LinearDifferentialOperator):
    def __init__(self, lindiffop: LinearDifferentialOperator, /, scalar):
        if np.ndim(scalar) != 0:
            raise ValueError("`scalar` must be a scalar.")
        self._lindiffop = lindiffop
        self._scalar = np.asarray(scalar, dtype=np.double)
        super().__init__(coefficients=float(self._scalar) * lindiffop.coefficients,
                         input_shapes=lindiffop.input_shapes)

    @property
    def lindiffop(self):
        return self._lindiffop

    @property
    def scalar(self):
        return self._scalar

    def __rmul__(self, other):
        if np.ndim(other) == 0:
            return ScaledLinearDifferentialOperator(self._lindiffop, scalar=np.asarray(other) * self._scalar)
        return NotImplemented

    def __repr__(self):
        return f"{self._scalar} * {self._lindiffop}"


class PartialDerivative(LinearDifferentialOperator):
    def __init__(self, multi_index: MultiIndex):
        self._multi_index = multi_index
        coeffs = PartialDerivativeCoefficients({(): {multi_index: 1.0}}, multi_index.shape, ())
        super().__init__(coeffs, input_shapes=(multi_index.shape, ()))

    @property
    def multi_index(self) -> MultiIndex:
        return self._multi_index

    @property
    def order(self) -> int:
        return self._multi_index.order

    @property
    def is_mixed(self) -> bool:
        return self._multi_index.is_mixed


class TimeDerivative(PartialDerivative):
    """d/dt with t = x[0]  (`_partial_derivative.py:131`)."""

    def __init__(self, domain_shape):
        domain_shape = _as_shape(domain_shape)
        if len(domain_shape) != 1:
            raise ValueError("`TimeDerivative` needs a one-dimensional domain shape.")
        super().__init__(MultiIndex.from_index((0,), domain_shape, 1))


class Derivative(PartialDerivative):
    """d^order/dx^order of a univariate function (`_derivative.py:11`)."""

    def __init__(self, order: int):
        if order < 0:
            raise ValueError(f"Order must be >= 0, but got {order}.")
        super().__init__(MultiIndex(order))


class DirectionalDerivative(LinearDifferentialOperator):
    """sum_i direction_i d/dx_i  (`_directional_derivative.py:15`)."""

    def __init__(self, direction):
        direction = np.asarray(direction, dtype=np.double)
        self._direction = direction
        entries = {
            MultiIndex.from_index(idx, direction.shape, 1): float(c)
            for idx, c in np.ndenumerate(direction) if c != 0.0
        }
        coeffs = PartialDerivativeCoefficients({(): entries}, direction.shape, ())
        super().__init__(coeffs, input_shapes=(direction.shape, ()))

    @property
    def direction(self):
        return self._direction


class WeightedLaplacian(LinearDifferentialOperator):
    r"""\sum_i w_i \partial^2 / \partial x_i^2  (`_laplacian.py:22-54`)."""

    def __init__(self, weights):
        weights = np.asarray(weights, dtype=np.double)
        self._weights = weights
        entries = {
            MultiIndex.from_index(idx, weights.shape, 2): float(c)
            for idx, c in np.ndenumerate(weights) if c != 0.0
        }
        coeffs = PartialDerivativeCoefficients({(): entries}, weights.shape, ())
        super().__init__(coeffs, input_shapes=(weights.shape, ()))

    @property
    def weights(self):
        return self._weights


class Laplacian(WeightedLaplacian):
    def __init__(self, domain_shape):
        super().__init__(np.ones(_as_shape(domain_shape) if domain_shape != () else (), dtype=np.double))


class SpatialLaplacian(WeightedLaplacian):
    """Laplacian over x[1:], leaving the time coordinate x[0] alone (`_laplacian.py:102`)."""

    def __init__(self, domain_shape):
        domain_shape = _as_shape(domain_shape)
        if len(domain_shape) != 1 or domain_shape[0] < 2:
            raise ValueError("`SpatialLaplacian` needs a domain shape (d,) with d >= 2.")
        w = np.ones(domain_shape, dtype=np.double)
        w[0] = 0.0
        super().__init__(w)


class HeatOperator(SumLinearFunctionOperator):
    """d/dt - alpha * Laplacian_x on (t, x) inputs (`_heat.py:14-39`)."""

    def __init__(self, domain_shape, alpha=1.0):
        domain_shape = _as_shape(domain_shape)
        if len(domain_shape) != 1:
            raise ValueError("The `HeatOperator` only applies to functions with `input_ndim == 1`.")
        self._alpha = float(alpha)
        w = np.zeros(domain_shape, dtype=np.double)
        w[1:] = -self._alpha
        super().__init__(TimeDerivative(domain_shape), WeightedLaplacian(w))

    @property
    def alpha(self):
        return self._alpha
