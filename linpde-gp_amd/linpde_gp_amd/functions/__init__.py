"""Deterministic functions used as prior means / right-hand sides.

Host-side mirror of `linpde_gp.functions.{Zero, Constant, Polynomial, Monomial, Affine}`
(`functions/_constant.py:12-75`, `_polynomial.py:17-98`, `_affine.py:9-51` of the reference); only what the GP-posterior
hot path touches.  A function maps arrays of shape `batch + input_shape` to `batch + output_shape`.
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

    def __init__(self, fn, input_shape, output_shape=(), derivatives=None):
        super().__init__(input_shape, output_shape)
        self._fn = fn
        self._derivatives = {tuple(int(i) for i in mi): d for mi, d in (derivatives or {}).items()}

    def _evaluate(self, x):
        return np.asarray(self._fn(x), dtype=np.double)

    def partial_derivative(self, multi_index):
        """Analytic derivatives supplied by the caller: `derivatives = {multi_index: vectorised callable}`."""
        mi = tuple(int(i) for i in multi_index)
        if not any(mi):
            return self
        fn = self._derivatives.get(mi)
        if fn is None:
            raise NotImplementedError(
                f"no derivative {mi} was supplied for this LambdaFunction (the reference would use JAX autodiff: out of scope)")
        return LambdaFunction(fn, self.input_shape, self.output_shape)


# ---------------------------------------------------------------------------------------------
# Prior means with exact derivatives.  The reference differentiates non-constant means through its JAX
# fallback (`linfuncops/diffops/_lindiffop.py:104-129`, every `functions.*` class is a `JaxFunction`); there is
# no autodiff here, so the classes below carry their derivatives in closed form and
# `LinearFunctionOperator.__call__` (linfuncops/_linfuncop.py) applies  sum_alpha c_alpha d^alpha  term by term.
# Protocol: `f.partial_derivative(multi_index) -> Function`.
# ---------------------------------------------------------------------------------------------
class Polynomial(Function):
    """`p(x) = sum_k coeffs[k] x^k` on the real line (`functions/_polynomial.py:39-98`: same constructor, `coefficients`,
    `degree`, `differentiate`, `integrate`, unary minus, `+`, `-`, scalar `*`)."""

    def __init__(self, coeffs):
        self._coeffs = tuple(float(c) for c in coeffs)
        super().__init__(input_shape=(), output_shape=())

    @property
    def coefficients(self):
        return self._coeffs

    @property
    def degree(self) -> int:
        return len(self._coeffs) - 1

    def __repr__(self):
        return "Polynomial(" + " + ".join(f"{c} x^{k}" for k, c in enumerate(self._coeffs)) + ")"

    def _evaluate(self, x):
        out = np.zeros_like(x)
        for c in reversed(self._coeffs):          # Horner
            out = out * x + c
        return out

    def differentiate(self) -> "Polynomial":
        return Polynomial([k * c for k, c in enumerate(self._coeffs)][1:] or [0.0])

    def integrate(self) -> "Polynomial":
        return Polynomial([0.0] + [c / (k + 1) for k, c in enumerate(self._coeffs)])

    def partial_derivative(self, multi_index):
        (n,) = tuple(int(i) for i in multi_index)
        p = self
        for _ in range(n):
            p = p.differentiate()
        return p

    def __neg__(self):
        return Polynomial([-c for c in self._coeffs])

    def __add__(self, other):
        if isinstance(other, Constant) and other.input_shape == () and other.output_shape == ():
            other = Polynomial([float(other.value)])
        if not isinstance(other, Polynomial):
            return NotImplemented
        n = max(len(self._coeffs), len(other._coeffs))
        a = self._coeffs + (0.0,) * (n - len(self._coeffs))
        b = other._coeffs + (0.0,) * (n - len(other._coeffs))
        return Polynomial([x + y for x, y in zip(a, b)])

    def __sub__(self, other):
        return self + (-other)

    def __rmul__(self, other):
        if np.ndim(other) == 0:
            return Polynomial([float(other) * c for c in self._coeffs])
        return NotImplemented


class Monomial(Polynomial):
    """`x^degree` (`functions/_polynomial.py:17-36`)."""

    def __init__(self, degree: int):
        if degree < 0:
            raise ValueError("the degree of a monomial must be non-negative")
        super().__init__([0.0] * int(degree) + [1.0])


class Piecewise(Function):
    """Scalar function of the real line given piece by piece on the intervals between the break points `xs`
    (`functions/_piecewise.py:16-86`: same constructor, `xs`, `pieces`, `num_pieces`, scalar `*`): piece i on
    (xs[i], xs[i + 1]], the first one closed on the left as well; zero outside [xs[0], xs[-1]], as `np.piecewise` leaves it."""

    def __init__(self, xs, fns):
        xs = np.atleast_1d(np.asarray(xs, dtype=np.double))
        if xs.ndim != 1:
            raise ValueError("the break points must form a vector")
        fns = tuple(fns)
        if len(fns) != xs.size - 1:
            raise ValueError("one piece per interval between consecutive break points is required")
        if not all(f.input_shape == () and f.output_shape == () for f in fns):
            raise ValueError("the pieces must be scalar functions of the real line")
        self._xs, self._fns = xs, fns
        super().__init__(input_shape=(), output_shape=())

    @property
    def xs(self):
        return self._xs

    @property
    def pieces(self):
        return self._fns

    @property
    def num_pieces(self) -> int:
        return len(self._fns)

    def _evaluate(self, x):
        out = np.zeros_like(x)
        # interval index of every point: i with xs[i] < x <= xs[i + 1] (x == xs[0] belongs to the first piece)
        idx = np.searchsorted(self._xs, x, side="left") - 1
        idx = np.where(x == self._xs[0], 0, idx)
        inside = (x >= self._xs[0]) & (x <= self._xs[-1])
        for i, f in enumerate(self._fns):
            sel = inside & (idx == i)
            if np.any(sel):
                out[sel] = f(x[sel])
        return out

    def _like(self, fns):
        return Piecewise(self._xs, fns)

    def __rmul__(self, other):
        if np.ndim(other) == 0:
            return self._like([float(other) * f for f in self._fns])
        return NotImplemented

    def __neg__(self):
        return -1.0 * self

    def partial_derivative(self, multi_index):
        return Piecewise(self._xs, [differentiate(f, multi_index) for f in self._fns])


class PiecewiseLinear(Piecewise):
    """Continuous piecewise-linear interpolant (`functions/_piecewise.py:89-142`: `from_points`, `+` with a `Constant` or a
    `Polynomial` -- of degree <= 1: stays `PiecewiseLinear` --, scalar `*`): the heat-source profiles of the CPU-die
    experiment (`experiments/cpu.py:83-137`)."""

    @staticmethod
    def from_points(xs, ys) -> "PiecewiseLinear":
        xs, ys = np.asarray(xs, dtype=np.double), np.asarray(ys, dtype=np.double)
        pieces = []
        for lo, hi, y_lo, y_hi in zip(xs[:-1], xs[1:], ys[:-1], ys[1:]):
            slope = (y_hi - y_lo) / (hi - lo)
            pieces.append(Polynomial((y_lo - slope * lo, slope)))
        return PiecewiseLinear(xs, pieces)

    def _like(self, fns):
        return PiecewiseLinear(self._xs, fns)

    def __add__(self, other):
        if isinstance(other, Constant) and other.input_shape == () and other.output_shape == ():
            other = Polynomial([float(other.value)])
        if not isinstance(other, Polynomial):
            return NotImplemented
        fns = [f + other for f in self._fns]
        return PiecewiseLinear(self._xs, fns) if other.degree <= 1 else Piecewise(self._xs, fns)

    __radd__ = __add__


class Affine(Function):
    """`A x + b` with the shape rules of `functions/_affine.py:9-51` (0-d `A`: scalar map of the real line; 1-d `A`: real line
    -> R^m; 2-d `A`: R^d -> R^m).  Only the scalar form can be a prior mean here (single-output GPs)."""

    def __init__(self, A, b):
        self._A = np.asarray(A, dtype=np.double)
        self._b = np.asarray(b, dtype=np.double)
        if self._A.ndim == 0:
            input_shape, output_shape = (), ()
        elif self._A.ndim == 1:
            input_shape, output_shape = (), self._A.shape
        elif self._A.ndim == 2:
            input_shape, output_shape = (self._A.shape[1],), (self._A.shape[0],)
        else:
            raise ValueError("`A` must have at most two dimensions")
        if self._b.shape != output_shape:
            raise ValueError(f"`b` must have shape {output_shape}, got {self._b.shape}")
        super().__init__(input_shape, output_shape)

    @property
    def A(self):
        return self._A

    @property
    def b(self):
        return self._b

    def _evaluate(self, x):
        if self._input_shape == ():
            return self._A * x[..., None] + self._b if self._A.ndim == 1 else self._A * x + self._b
        return (self._A @ x[..., None])[..., 0] + self._b

    def partial_derivative(self, multi_index):
        mi = tuple(int(i) for i in multi_index)
        order = sum(mi)
        if order == 0:
            return self
        if order >= 2:
            return Zero(self.input_shape, self.output_shape)
        if self._input_shape == ():
            return Constant((), self._A)
        return Constant(self.input_shape, self._A[:, mi.index(1)])


class LinearCombination(Function):
    """`sum_i c_i f_i`: what a differential operator applied to a mean function returns."""

    def __init__(self, coeffs, fns):
        fns = tuple(fns)
        if not fns:
            raise ValueError("at least one function is required")
        if not all(f.input_shape == fns[0].input_shape and f.output_shape == fns[0].output_shape for f in fns):
            raise ValueError("all functions must have the same input and output shapes")
        self._coeffs = tuple(float(c) for c in coeffs)
        self._fns = fns
        super().__init__(fns[0].input_shape, fns[0].output_shape)

    def _evaluate(self, x):
        out = self._coeffs[0] * self._fns[0](x)
        for c, f in zip(self._coeffs[1:], self._fns[1:]):
            out = out + c * f(x)
        return out

    def partial_derivative(self, multi_index):
        return LinearCombination(self._coeffs, [differentiate(f, multi_index) for f in self._fns])


def differentiate(f: Function, multi_index) -> Function:
    """`d^alpha f` for a function that knows its derivatives; NotImplementedError otherwise (the reference's JAX fallback)."""
    mi = tuple(int(i) for i in multi_index)
    if isinstance(f, Constant):
        return f if not any(mi) else Zero(f.input_shape, f.output_shape)
    pd = getattr(f, "partial_derivative", None)
    if pd is None:
        raise NotImplementedError(
            f"{type(f).__name__} has no closed-form derivatives; applying a differential operator to it needs autodiff "
            "(JAX fallback of the reference, `diffops/_lindiffop.py:104-129`): out of scope.  Use `Polynomial`, `Affine`, "
            "`Constant`, or a `LambdaFunction(fn, ..., derivatives={multi_index: fn})`.")
    return pd(mi)


def apply_coefficients(coeffs: dict, f: Function) -> Function:
    """`(sum_alpha c_alpha d^alpha) f` for a constant-coefficient operator in canonical form (`coefficients_dict`)."""
    items = [(mi, c) for mi, c in coeffs.items() if c != 0.0]
    if not items:
        return Zero(f.input_shape, f.output_shape)
    if isinstance(f, Constant):
        c0 = sum(c for mi, c in items if not any(mi))
        return Constant(f.input_shape, c0 * f.value)
    return LinearCombination([c for _, c in items], [differentiate(f, mi) for mi, _ in items])
