"""Linear functionals on the hot path: point evaluation, optionally composed with a
differential operator.

Host mirror of
  `linfunctls/_linfunctl.py:14-129`   LinearFunctional (+ `@` with a function operator)
  `linfunctls/_evaluation.py:10-64`   _EvaluationFunctional (output = codomain shape, then
                                      batch shape; scalar-valued priors only here)
  `linfunctls/_dirac.py:10-45`        DiracFunctional (output = batch shape, then codomain shape:
                                      the same functional for the scalar-valued priors here)
  `linfunctls/_arithmetic.py:13-174`  Scaled / Sum / CompositeLinearFunctional and the operators
                                      `-L`, `a * L`, `L1 + L2`, `L1 - L2`, `L @ D`
                                      (`_linfunctl.py:76-112`)
Every functional of this path has the canonical form "point evaluation of `sum_a c_a d^a f` at
X": scaling multiplies the coefficient map, a sum over the SAME points adds the maps (a sum
over different point sets is not one observation block and raises NotImplementedError).
Integrals, L2 projections and weak forms are out of scope.
"""

from __future__ import annotations

import numpy as np

from ..linfuncops import LinearFunctionOperator


def _as_shape(shape):
    if isinstance(shape, (int, np.integer)):
        return (int(shape),)
    return tuple(int(s) for s in shape)


class LinearFunctional:
    def __init__(self, input_shapes, output_shape):
        self._input_domain_shape = _as_shape(input_shapes[0])
        self._input_codomain_shape = _as_shape(input_shapes[1])
        self._output_shape = _as_shape(output_shape)

    @property
    def input_shapes(self):
        return (self._input_domain_shape, self._input_codomain_shape)

    @property
    def input_domain_shape(self):
        return self._input_domain_shape

    @property
    def input_codomain_shape(self):
        return self._input_codomain_shape

    @property
    def output_shape(self):
        return self._output_shape

    @property
    def output_size(self):
        return int(np.prod(self._output_shape, dtype=int))

    # canonical form used by the GPU path: evaluation points + operator coefficient map
    def points(self) -> np.ndarray:
        raise NotImplementedError

    def coefficients_dict(self) -> dict:
        raise NotImplementedError

    def device_points(self, ctx):
        """Device-resident evaluation points (reuses the handle of a `to_device` array)."""
        from .. import _engine

        return _engine.Points(ctx, self.points())

    def __call__(self, f, /, *, argnum: int = 0):
        from ..functions import Function
        from ..randprocs import _gaussian_process as gps
        from ..randprocs import covfuncs

        if isinstance(f, gps.ConditionalGaussianProcess):
            return gps.apply_linfunctl_to_conditional_gp(self, f)
        from ..randprocs._matrix_free import MatrixFreeConditionalGaussianProcess
        if isinstance(f, MatrixFreeConditionalGaussianProcess):
            # `L(posterior)` for a functional: the joint law of L[f] at the functional's points, through the matrix-free read-out
            from ..randvars import Normal
            view = f._with_test_operator(self)
            X = self.points()
            x = X if f.input_ndim else X[:, 0]
            return Normal(view.mean(x), view.cov.matrix(x))
        if isinstance(f, gps.GaussianProcess):
            return gps.apply_linfunctl_to_gp(self, f)
        if isinstance(f, covfuncs.CovarianceFunction):
            return covfuncs.ProcessVectorCrossCovariance(f, self, argnum=argnum)
        if isinstance(f, covfuncs.ProcessVectorCrossCovariance):
            return covfuncs.apply_linfunctl_to_pv_crosscov(self, f)
        if isinstance(f, Function):
            return self._apply_to_function(f)
        raise NotImplementedError(f"cannot apply {type(self).__name__} to {type(f).__name__}")

    def _apply_to_function(self, f):
        raise NotImplementedError

    def __matmul__(self, other):
        if isinstance(other, LinearFunctionOperator):
            return CompositeLinearFunctional(linfunctl=self, linfuncop=other)
        return NotImplemented

    # -- arithmetic (`_linfunctl.py:76-98`) --
    __array_ufunc__ = None

    def __neg__(self):
        return -1.0 * self

    def __add__(self, other):
        if isinstance(other, LinearFunctional):
            return SumLinearFunctional(self, other)
        return NotImplemented

    def __sub__(self, other):
        if isinstance(other, LinearFunctional):
            return self + (-other)
        return NotImplemented

    def __rmul__(self, other):
        if np.ndim(other) == 0:
            return ScaledLinearFunctional(linfunctl=self, scalar=other)
        return NotImplemented


class _EvaluationFunctional(LinearFunctional):
    def __init__(self, input_domain_shape, input_codomain_shape, X):
        input_domain_shape = _as_shape(input_domain_shape)
        input_codomain_shape = _as_shape(input_codomain_shape)
        self._X = np.asanyarray(X)
        nd = len(input_domain_shape)
        self._X_batch_shape = self._X.shape[: self._X.ndim - nd]
        if self._X.shape != self._X_batch_shape + input_domain_shape:
            raise ValueError(
                f"X has shape {self._X.shape}, expected batch shape + {input_domain_shape}")
        super().__init__(
            input_shapes=(input_domain_shape, input_codomain_shape),
            output_shape=input_codomain_shape + self._X_batch_shape,
        )

    @property
    def X(self):
        return self._X

    @property
    def X_batch_shape(self):
        return self._X_batch_shape

    def points(self) -> np.ndarray:
        d = self._input_domain_shape[0] if self._input_domain_shape else 1
        return np.ascontiguousarray(np.asarray(self._X, dtype=np.double).reshape(-1, d))

    def coefficients_dict(self):
        d = self._input_domain_shape[0] if self._input_domain_shape else 1
        return {(0,) * d: 1.0}

    def device_points(self, ctx):
        from .. import _engine

        return _engine.as_points(ctx, self._X, self.points())

    def _apply_to_function(self, f):
        res = np.asarray(f(self._X))
        if f.output_ndim > 0:
            res = np.moveaxis(res, res.ndim - f.output_ndim + np.arange(f.output_ndim), np.arange(f.output_ndim))
        return res


class DiracFunctional(_EvaluationFunctional):
    """Point evaluation with output layout batch shape + codomain shape (`_dirac.py:10-45`); for
    scalar-valued functions (the only ones on this path) the same functional as
    `_EvaluationFunctional`."""

    def __init__(self, input_domain_shape, input_codomain_shape, X):
        super().__init__(input_domain_shape, input_codomain_shape, X)
        self._output_shape = self._X_batch_shape + self._input_codomain_shape

    @property
    def X_batch_ndim(self):
        return len(self._X_batch_shape)

    def _apply_to_function(self, f):
        return np.asarray(f(self._X))


class ScaledLinearFunctional(LinearFunctional):
    """`scalar * linfunctl` (`_arithmetic.py:13-55`)."""

    def __init__(self, linfunctl: LinearFunctional, scalar):
        if np.ndim(scalar) != 0:
            raise ValueError("`scalar` must be a scalar")
        self._linfunctl = linfunctl
        self._scalar = float(scalar)
        super().__init__(input_shapes=linfunctl.input_shapes, output_shape=linfunctl.output_shape)

    @property
    def linfunctl(self):
        return self._linfunctl

    @property
    def scalar(self):
        return self._scalar

    def points(self):
        return self._linfunctl.points()

    def device_points(self, ctx):
        return self._linfunctl.device_points(ctx)

    def coefficients_dict(self):
        return {mi: self._scalar * c for mi, c in self._linfunctl.coefficients_dict().items()}

    def _apply_to_function(self, f):
        return self._scalar * self._linfunctl(f)

    def __rmul__(self, other):
        if np.ndim(other) == 0:
            return ScaledLinearFunctional(linfunctl=self._linfunctl, scalar=float(other) * self._scalar)
        return NotImplemented


class SumLinearFunctional(LinearFunctional):
    """`L1 + L2 + ...` (`_arithmetic.py:58-89`).  On the MI355X path the summands must evaluate at
    the same points (e.g. a Robin condition `a * Id + b * d/dn` on one boundary grid)."""

    def __init__(self, *summands: LinearFunctional):
        if len(summands) < 1:
            raise ValueError("at least one summand is required")
        s0 = summands[0]
        if not all(s.input_shapes == s0.input_shapes for s in summands):
            raise ValueError("all summands must have the same input shapes")
        if not all(s.output_shape == s0.output_shape for s in summands):
            raise ValueError("all summands must have the same output shape")
        self._summands = tuple(summands)
        super().__init__(input_shapes=s0.input_shapes, output_shape=s0.output_shape)

    @property
    def summands(self):
        return self._summands

    def points(self):
        P = self._summands[0].points()
        for s in self._summands[1:]:
            Q = s.points()
            if Q.shape != P.shape or not np.array_equal(P, Q):
                raise NotImplementedError(
                    "a sum of functionals over different point sets is not one observation block; "
                    "condition on the summands' blocks separately")
        return P

    def device_points(self, ctx):
        self.points()
        return self._summands[0].device_points(ctx)

    def coefficients_dict(self):
        out: dict = {}
        for s in self._summands:
            for mi, c in s.coefficients_dict().items():
                out[mi] = out.get(mi, 0.0) + c
        return out

    def _apply_to_function(self, f):
        res = self._summands[0](f)
        for s in self._summands[1:]:
            res = res + s(f)
        return res


class CompositeLinearFunctional(LinearFunctional):
    """`linfunctl @ linfuncop` -- here: point evaluation of `linfuncop[f]`."""

    def __init__(self, *, linfunctl: LinearFunctional, linfuncop: LinearFunctionOperator, linop=None):
        if linop is not None:
            raise NotImplementedError("a leading matrix factor is not supported on the MI355X path")
        if linfunctl.input_shapes != linfuncop.output_shapes:
            raise ValueError("shapes of the functional and the operator do not match")
        self._linfunctl = linfunctl
        self._linfuncop = linfuncop
        super().__init__(input_shapes=linfuncop.input_shapes, output_shape=linfunctl.output_shape)

    @property
    def linfunctl(self):
        return self._linfunctl

    @property
    def linfuncop(self):
        return self._linfuncop

    def points(self):
        return self._linfunctl.points()

    def device_points(self, ctx):
        return self._linfunctl.device_points(ctx)

    def coefficients_dict(self):
        inner = self._linfunctl.coefficients_dict()
        d = len(next(iter(inner)))
        if set(inner) != {(0,) * d}:
            raise NotImplementedError("only point evaluation may be composed with an operator")
        c0 = inner[(0,) * d]
        return {mi: c0 * c for mi, c in self._linfuncop.coefficients_dict().items()}

    def _apply_to_function(self, f):
        return self._linfunctl(self._linfuncop(f))


__all__ = ["LinearFunctional", "_EvaluationFunctional", "DiracFunctional", "ScaledLinearFunctional",
           "SumLinearFunctional", "CompositeLinearFunctional"]
