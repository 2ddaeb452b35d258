"""`LinearFunctionOperator` base -- host mirror of `linfuncops/_linfuncop.py:16-136`.

Everything on the hot path is a differential operator with constant coefficients, so the
whole symbolic layer reduces to ONE canonical form: a map
`{multi_index (tuple of per-dimension orders): coefficient}` (`coefficients_dict`).
Applying an operator to a covariance function (`L(k, argnum=...)`) therefore never
evaluates anything: it returns a `DifferentiatedCovarianceFunction` that remembers the
two coefficient maps and is lowered to the C-ABI descriptor when a Gram / cross block is
assembled on the GPU.
"""

from __future__ import annotations

import numpy as np


class LinearFunctionOperator:
    def __init__(self, input_shapes, output_shapes):
        self._input_domain_shape = tuple(input_shapes[0])
        self._input_codomain_shape = tuple(input_shapes[1])
        self._output_domain_shape = tuple(output_shapes[0])
        self._output_codomain_shape = tuple(output_shapes[1])

    @property
    def input_shapes(self):
        return (self._input_domain_shape, self._input_codomain_shape)

    @property
    def output_shapes(self):
        return (self._output_domain_shape, self._output_codomain_shape)

    @property
    def input_domain_shape(self):
        return self._input_domain_shape

    @property
    def input_codomain_shape(self):
        return self._input_codomain_shape

    # -- canonical form ---------------------------------------------------------------
    def coefficients_dict(self) -> dict[tuple[int, ...], float]:
        raise NotImplementedError(
            f"{type(self).__name__} has no constant-coefficient differential form; "
            "the reference would fall back to JAX autodiff here (out of scope)."
        )

    # -- application --------------------------------------------------------------------
    def __call__(self, f, /, *, argnum: int = 0):
        # local imports: the packages import each other like in the reference
        from ..functions import Function
        from ..randprocs import covfuncs
        from ..randprocs import _gaussian_process as gps

        if isinstance(f, covfuncs.CovarianceFunction):
            return covfuncs.apply_linfuncop(self, f, argnum=argnum)
        if isinstance(f, gps.ConditionalGaussianProcess):
            return gps.apply_linfuncop_to_conditional_gp(self, f)
        from ..randprocs._matrix_free import MatrixFreeConditionalGaussianProcess
        if isinstance(f, MatrixFreeConditionalGaussianProcess):
            return f._with_test_operator(self)
        if isinstance(f, gps.GaussianProcess):
            return gps.GaussianProcess(
                mean=self(f.mean),
                cov=self(self(f.cov, argnum=1), argnum=0),
            )
        if isinstance(f, Function):
            # sum_alpha c_alpha d^alpha f, term by term, for means that carry their derivatives in closed form
            # (Constant / Zero, Polynomial, Affine, LambdaFunction with `derivatives`); NotImplementedError where the
            # reference would differentiate by JAX autodiff (`diffops/_lindiffop.py:104-129`)
            from ..functions import apply_coefficients
            d = len(self._input_domain_shape) and self._input_domain_shape[0] or 1
            coeffs = self.coefficients_dict()
            if any(len(mi) != d for mi in coeffs):
                raise ValueError("the operator's multi-indices do not match the input dimension of the function")
            return apply_coefficients(coeffs, f)
        raise NotImplementedError(f"cannot apply {type(self).__name__} to {type(f).__name__}")

    def to_linfunctl(self, X):
        """`_EvaluationFunctional(X) @ self`  (`linfuncops/_linfuncop.py:93-105`)."""
        from ..linfunctls import CompositeLinearFunctional, _EvaluationFunctional

        return CompositeLinearFunctional(
            linfunctl=_EvaluationFunctional(
                input_domain_shape=self._output_domain_shape,
                input_codomain_shape=self._output_codomain_shape,
                X=X,
            ),
            linfuncop=self,
        )

    # -- algebra ------------------------------------------------------------------------
    def __rmul__(self, other):
        if np.ndim(other) == 0:
            return ScaledLinearFunctionOperator(self, scalar=other)
        return NotImplemented

    def __neg__(self):
        return -1.0 * self

    def __add__(self, other):
        if isinstance(other, LinearFunctionOperator):
            return SumLinearFunctionOperator(self, other)
        return NotImplemented

    def __sub__(self, other):
        if isinstance(other, LinearFunctionOperator):
            return SumLinearFunctionOperator(self, -other)
        return NotImplemented


class ScaledLinearFunctionOperator(LinearFunctionOperator):
    """`linfuncops/_arithmetic.py:12-52`."""

    def __init__(self, linfuncop, scalar):
        if np.ndim(scalar) != 0:
            raise ValueError("`scalar` must be a scalar.")
        self._linfuncop = linfuncop
        self._scalar = float(scalar)
        super().__init__(linfuncop.input_shapes, linfuncop.output_shapes)

    @property
    def linfuncop(self):
        return self._linfuncop

    @property
    def scalar(self):
        return self._scalar

    def coefficients_dict(self):
        return {mi: self._scalar * c for mi, c in self._linfuncop.coefficients_dict().items()}

    def __rmul__(self, other):
        if np.ndim(other) == 0:
            return ScaledLinearFunctionOperator(self._linfuncop, scalar=float(other) * self._scalar)
        return NotImplemented


class SumLinearFunctionOperator(LinearFunctionOperator):
    """`linfuncops/_arithmetic.py:55-110`: distributes over its summands."""

    def __init__(self, *summands):
        if not summands:
            raise ValueError("at least one summand is required")
        if not all(s.input_shapes == summands[0].input_shapes and s.output_shapes == summands[0].output_shapes
                   for s in summands):
            raise ValueError("all summands must have the same input and output shapes")
        self._summands = tuple(summands)
        super().__init__(summands[0].input_shapes, summands[0].output_shapes)

    @property
    def summands(self):
        return self._summands

    def coefficients_dict(self):
        out: dict = {}
        for s in self._summands:
            for mi, c in s.coefficients_dict().items():
                out[mi] = out.get(mi, 0.0) + c
        return out


class Identity(LinearFunctionOperator):
    """`linfuncops/_identity.py:9`."""

    def __init__(self, domain_shape, codomain_shape=()):
        if isinstance(domain_shape, (int, np.integer)):
            domain_shape = (int(domain_shape),)
        domain_shape = tuple(int(s) for s in domain_shape)
        codomain_shape = tuple(int(s) for s in codomain_shape)
        super().__init__((domain_shape, codomain_shape), (domain_shape, codomain_shape))

    def coefficients_dict(self):
        d = self._input_domain_shape[0] if self._input_domain_shape else 1
        return {(0,) * d: 1.0}
