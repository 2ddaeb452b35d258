"""Covariance functions of the hot path and their symbolic differentiation.

Host mirror of
  probnum `Matern`, `ExpQuad` (third party) and `covfuncs/_matern.py`, `_expquad.py`
  `covfuncs/_tensor_product.py:15-82`        TensorProduct
  `covfuncs/_jax_arithmetic.py:16-66`        scaled / sum kernels
  `covfuncs/linfuncops/_registry.py:14-31`   L(alpha*k) = alpha*L(k),  L(k1+k2) = L(k1)+L(k2)
  `covfuncs/linfuncops/diffops/_registry.py:31-72` + `_tensor_product.py:22-70`
       L0 (k_1 x ... x k_d) L1'^* = sum_{a in L0} sum_{b in L1} c_a c'_b prod_i d^{a_i} d'^{b_i} k_i
  `covfuncs/linfunctls/_registry.py:125-149` + `crosscov/_pv_crosscov.py:14-199`
       ProcessVectorCrossCovariance  x -> Cov(f(x), L[f])
The reference picks a closed-form Python class per (kernel, L0, L1) combination through a
singledispatch table; here every combination lowers to the SAME descriptor (term list of
per-dimension derivative orders, `lpgp_kdesc` of include/lpgp.h) and one HIP kernel
evaluates it.  Nothing is evaluated on the CPU.
"""

from __future__ import annotations

import numpy as np

from ..._lib import EXPQUAD, MATERN_HALFINT, MATERN_ISO, MAXD, MAXG, MAXT


def _as_shape(shape):
    if isinstance(shape, (int, np.integer)):
        return (int(shape),)
    return tuple(int(s) for s in shape)


class CovarianceFunction:
    """Scalar-valued covariance function k: R^d x R^d -> R (input_shape () or (d,))."""

    def __init__(self, input_shape=()):
        self._input_shape = _as_shape(input_shape)
        if len(self._input_shape) > 1:
            raise ValueError("only input shapes () and (d,) are supported")

    # -- shapes (probnum CovarianceFunction API) --
    @property
    def input_shape(self):
        return self._input_shape

    @property
    def input_ndim(self):
        return len(self._input_shape)

    @property
    def input_size(self):
        n = 1
        for s_ in self._input_shape:
            n *= int(s_)
        return n

    @property
    def output_shape_0(self):
        return ()

    @property
    def output_shape_1(self):
        return ()

    # -- canonical form --
    def _base_groups(self):
        """list of (scale, [(family, p, lengthscale), ...per dim]) -- a sum of scaled tensor products."""
        raise NotImplementedError

    def _operator_coeffs(self):
        d = max(self.input_size, 1)
        ident = {(0,) * d: 1.0}
        return ident, ident

    def lower(self):
        """C-ABI descriptor groups of this covariance function (see `_lib.make_kdesc_array`)."""
        L0, L1 = self._operator_coeffs()
        return lower_groups(self._base_groups(), L0, L1)

    # -- evaluation (GPU) --
    def _points(self, x):
        x = np.asarray(x, dtype=np.double)
        d = max(self.input_size, 1)
        if self.input_ndim == 0:
            batch = x.shape
        else:
            if x.shape[-1:] != self._input_shape:
                raise ValueError(
                    f"The shape of the input {x.shape} is not compatible with the input shape "
                    f"{self._input_shape} of the covariance function.")
            batch = x.shape[:-1]
        return np.ascontiguousarray(x.reshape(-1, d)), batch

    def matrix(self, x0, x1=None) -> np.ndarray:
        """Dense kernel matrix between two point sets (each of shape (N,) + input_shape)."""
        from ... import _engine

        X0, b0 = self._points(x0)
        if len(b0) > 1:
            raise ValueError("`matrix` needs inputs of shape (N,) + input_shape")
        ctx = _engine.default_context()
        P0 = _engine.Points(ctx, X0)
        if x1 is None:
            return _engine.kernel_matrix(ctx, self.lower(), P0, P0)
        X1, b1 = self._points(x1)
        if len(b1) > 1:
            raise ValueError("`matrix` needs inputs of shape (N,) + input_shape")
        return _engine.kernel_matrix(ctx, self.lower(), P0, _engine.Points(ctx, X1))

    def linop(self, x0, x1=None) -> "KernelLinearOperator":
        """Matrix-free `k(x0, x1)` as a linear operator (probnum `CovarianceFunction.linop`; the
        reference backs it with KeOps lazy tensors, `_keops_lazy_tensor`): `@` evaluates the
        kernel on the fly on the device, `todense()` materialises it."""
        return KernelLinearOperator(self, x0, x0 if x1 is None else x1)

    def __call__(self, x0, x1=None) -> np.ndarray:
        """Broadcasting evaluation k(x0, x1); `x1=None` gives the diagonal k(x0, x0)."""
        from ... import _engine

        X0, b0 = self._points(x0)
        if x1 is None:
            v = _engine.kernel_diag(_engine.default_context(), self.lower())
            return np.full(b0, v)
        X1, b1 = self._points(x1)
        out_shape = np.broadcast_shapes(b0, b1)
        K = self.matrix(X0 if self.input_ndim else X0[:, 0], X1 if self.input_ndim else X1[:, 0])
        i0 = np.broadcast_to(np.arange(X0.shape[0]).reshape(b0), out_shape)
        i1 = np.broadcast_to(np.arange(X1.shape[0]).reshape(b1), out_shape)
        return K[i0, i1]

    # -- algebra --
    def __rmul__(self, other):
        if np.ndim(other) == 0:
            return ScaledCovarianceFunction(self, scalar=other)
        return NotImplemented

    def __mul__(self, other):
        return self.__rmul__(other)

    def __add__(self, other):
        if isinstance(other, CovarianceFunction):
            return SumCovarianceFunction(self, other)
        return NotImplemented


class Matern(CovarianceFunction):
    """Half-integer Matérn, k = kappa_nu(|| sqrt(2 nu) (x - x') / lengthscales ||) (probnum `Matern`).

    input_shape () or (1,): univariate, every differential operator with a closed form
    (`UnivariateHalfIntegerMatern_*`, `diffops/_matern.py:267-611`).  input_shape (d,), d > 1: the
    ISOTROPIC kernel (scalar or per-dimension lengthscales); closed forms exist for identity and
    directional derivatives on either argument (`HalfIntegerMatern_Identity_DirectionalDerivative`,
    `HalfIntegerMatern_DirectionalDerivative_DirectionalDerivative`, `_matern.py:17-264`), anything
    of higher order raises NotImplementedError (the reference falls back to JAX autodiff there)."""

    def __init__(self, input_shape=(), nu=1.5, lengthscales=1.0):
        super().__init__(input_shape)
        self._nu = float(nu)
        p = self._nu - 0.5
        if p < 0 or abs(p - round(p)) > 1e-12:
            raise NotImplementedError("only half-integer nu has a closed form (`matern.p is None` otherwise)")
        self._p = int(round(p))
        d = max(self.input_size, 1)
        if d > MAXD:
            raise NotImplementedError(f"at most {MAXD} input dimensions are supported")
        try:
            ls = np.broadcast_to(np.asarray(lengthscales, dtype=np.double), (d,)).copy()
        except ValueError:
            raise ValueError(f"`lengthscales` must be a scalar or have shape ({d},)") from None
        if not (ls > 0).all():
            raise ValueError("`lengthscales` must be positive")
        self._ls = ls
        self._lengthscales = float(ls[0]) if d == 1 else ls

    @property
    def nu(self):
        return self._nu

    @property
    def p(self):
        return self._p

    @property
    def lengthscales(self):
        return self._lengthscales

    def _base_groups(self):
        if self.input_size > 1:
            return [(1.0, [(MATERN_ISO, self._p, float(l)) for l in self._ls])]
        return [(1.0, [(MATERN_HALFINT, self._p, float(self._ls[0]))])]


class ExpQuad(CovarianceFunction):
    """k = exp(-||(x - x') / lengthscales||^2 / 2); factorises over input dimensions."""

    def __init__(self, input_shape=(), lengthscales=1.0):
        super().__init__(input_shape)
        d = max(self.input_size, 1)
        ls = np.broadcast_to(np.asarray(lengthscales, dtype=np.double), (d,)).copy()
        if not (ls > 0).all():
            raise ValueError("`lengthscales` must be positive")
        self._lengthscales = ls

    @property
    def lengthscales(self):
        return self._lengthscales if self.input_ndim else float(self._lengthscales[0])

    def _base_groups(self):
        return [(1.0, [(EXPQUAD, 0, float(l)) for l in self._lengthscales])]


class TensorProduct(CovarianceFunction):
    """k(x, x') = prod_i k_i(x_i, x'_i) with univariate factors (`_tensor_product.py:15-48`)."""

    def __init__(self, *factors: CovarianceFunction):
        if len(factors) < 1:
            raise ValueError("At least one factor is required.")
        if not all(k.input_shape == () for k in factors):
            raise ValueError("The input shape of all factors must be `()`.")
        for k in factors:
            g = k._base_groups()
            if len(g) != 1 or len(g[0][1]) != 1:
                raise NotImplementedError("factors must be plain (optionally scaled) univariate kernels")
        self._factors = tuple(factors)
        super().__init__(input_shape=(len(factors),))

    @property
    def factors(self):
        return self._factors

    def _base_groups(self):
        scale = 1.0
        fs = []
        for k in self._factors:
            (s, (f,)), = k._base_groups()
            scale *= s
            fs.append(f)
        return [(scale, fs)]


class ScaledCovarianceFunction(CovarianceFunction):
    def __init__(self, covfunc: CovarianceFunction, scalar):
        if np.ndim(scalar) != 0:
            raise ValueError("`scalar` must be a scalar")
        super().__init__(covfunc.input_shape)
        self._covfunc = covfunc
        self._scalar = float(scalar)

    @property
    def covfunc(self):
        return self._covfunc

    @property
    def scalar(self):
        return self._scalar

    def _base_groups(self):
        return [(self._scalar * s, f) for s, f in self._covfunc._base_groups()]

    def _operator_coeffs(self):
        return self._covfunc._operator_coeffs()


class SumCovarianceFunction(CovarianceFunction):
    def __init__(self, *summands: CovarianceFunction):
        if not all(s.input_shape == summands[0].input_shape for s in summands):
            raise ValueError("all summands must have the same input shape")
        ops = [s._operator_coeffs() for s in summands]
        if any(o != ops[0] for o in ops[1:]):
            raise NotImplementedError("summands must carry the same differential operators")
        super().__init__(summands[0].input_shape)
        self._summands = tuple(summands)

    @property
    def summands(self):
        return self._summands

    def _base_groups(self):
        return [g for s in self._summands for g in s._base_groups()]

    def _operator_coeffs(self):
        return self._summands[0]._operator_coeffs()


class KernelLinearOperator:
    """`k(X0, X1)` as a matrix-free operator: products run `lpgp_kernel_matvec` (every entry
    evaluated on the fly in registers), the point sets stay resident on the device."""

    def __init__(self, covfunc: CovarianceFunction, x0, x1):
        from ... import _engine

        X0, b0 = covfunc._points(x0)
        X1, b1 = covfunc._points(x1)
        if len(b0) != 1 or len(b1) != 1:
            raise ValueError("`linop` needs inputs of shape (N,) + input_shape")
        self._ctx = _engine.default_context()
        self._k = covfunc
        self._desc = covfunc.lower()
        self._P0 = _engine.Points(self._ctx, X0)
        self._P1 = self._P0 if x1 is x0 else _engine.Points(self._ctx, X1)
        self.shape = (X0.shape[0], X1.shape[0])
        self.dtype = np.dtype(np.double)

    def __matmul__(self, V):
        from ... import _engine

        V = np.asarray(V, dtype=np.double)
        if V.shape[0] != self.shape[1]:
            raise ValueError(f"shape mismatch: {self.shape} @ {V.shape}")
        return _engine.kernel_matvec(self._ctx, self._desc, self._P0, self._P1, V)

    matmul = __matmul__

    @property
    def T(self):
        return _TransposedKernelOperator(self)

    def todense(self):
        from ... import _engine

        return _engine.kernel_matrix(self._ctx, self._desc, self._P0, self._P1)


class _TransposedKernelOperator:
    def __init__(self, op: KernelLinearOperator):
        self._op = op
        self.shape = op.shape[::-1]
        # (L0 k L1')(x, x') = (L1 k L0')(x', x) for the symmetric base kernels here: exchange
        # the operator maps of the two arguments
        L0, L1 = op._k._operator_coeffs()
        self._desc = lower_groups(op._k._base_groups(), L1, L0)

    def __matmul__(self, V):
        from ... import _engine

        V = np.asarray(V, dtype=np.double)
        if V.shape[0] != self.shape[1]:
            raise ValueError(f"shape mismatch: {self.shape} @ {V.shape}")
        return _engine.kernel_matvec(self._op._ctx, self._desc, self._op._P1, self._op._P0, V)

    matmul = __matmul__

    @property
    def T(self):
        return self._op

    def todense(self):
        return self._op.todense().T


class Zero(CovarianceFunction):
    def __init__(self, input_shape=()):
        super().__init__(input_shape)

    def _base_groups(self):
        d = max(self.input_size, 1)
        return [(0.0, [(EXPQUAD, 0, 1.0)] * d)]


class DifferentiatedCovarianceFunction(CovarianceFunction):
    """`L0 k L1'^*` -- the counterpart of the reference's per-combination classes
    (`TensorProduct_LinDiffOp_LinDiffOp`, `UnivariateHalfIntegerMatern_*`, `ExpQuad_*`)."""

    def __init__(self, covfunc: CovarianceFunction, L0: dict, L1: dict):
        super().__init__(covfunc.input_shape)
        self._covfunc = covfunc
        self._L0 = dict(L0)
        self._L1 = dict(L1)

    @property
    def covfunc(self):
        return self._covfunc

    def _base_groups(self):
        return self._covfunc._base_groups()

    def _operator_coeffs(self):
        return self._L0, self._L1


def _compose(outer: dict, inner: dict) -> dict:
    """Coefficient map of (outer o inner): orders add, coefficients multiply."""
    out: dict = {}
    for a, ca in outer.items():
        for b, cb in inner.items():
            key = tuple(x + y for x, y in zip(a, b))
            out[key] = out.get(key, 0.0) + ca * cb
    return out


def apply_linfuncop(L, k: CovarianceFunction, *, argnum: int = 0) -> CovarianceFunction:
    """`L(k, argnum=...)` (`LinearFunctionOperator.__call__` on covariance functions)."""
    if argnum not in (0, 1):
        raise ValueError("`argnum` must be 0 or 1")
    coeffs = L.coefficients_dict()
    d = max(k.input_size, 1)
    if any(len(mi) != d for mi in coeffs):
        raise ValueError(
            f"operator acts on inputs of shape {L.input_domain_shape}, kernel has input shape {k.input_shape}")
    L0, L1 = k._operator_coeffs()
    if argnum == 0:
        L0 = _compose(coeffs, L0)
    else:
        L1 = _compose(coeffs, L1)
    base = k._covfunc if isinstance(k, DifferentiatedCovarianceFunction) else k
    return DifferentiatedCovarianceFunction(base, L0, L1)


def lower_groups(base_groups, L0: dict, L1: dict):
    """Build the `lpgp_kdesc` groups: every (alpha, beta) pair of L0 x L1 becomes one term."""
    if len(base_groups) > MAXG:
        raise NotImplementedError(f"at most {MAXG} summands are supported")
    terms: dict = {}
    for a, ca in L0.items():
        for b, cb in L1.items():
            key = (tuple(a), tuple(b))
            terms[key] = terms.get(key, 0.0) + ca * cb
    term_list = [(c, a, b) for (a, b), c in terms.items() if c != 0.0]
    if not term_list:
        d = len(next(iter(L0)))
        term_list = [(0.0, (0,) * d, (0,) * d)]
    if len(term_list) > MAXT:
        raise NotImplementedError(f"at most {MAXT} terms are supported")
    groups = []
    for scale, factors in base_groups:
        d = len(factors)
        if d > MAXD:
            raise NotImplementedError(f"at most {MAXD} input dimensions are supported")
        for c, a, b in term_list:
            if len(a) != d or len(b) != d:
                raise ValueError("operator and kernel dimensions do not match")
            if factors[0][0] == MATERN_ISO:
                p = factors[0][1]
                if (sum(a) > 1 or sum(b) > 1) and c != 0.0:
                    raise NotImplementedError(
                        "the isotropic multivariate Matérn kernel has closed forms for identity and "
                        "directional derivatives only (no JAX autodiff fallback on the MI355X path); "
                        "use a `TensorProduct` prior for higher-order operators")
                if sum(a) + sum(b) > p and c != 0.0:
                    raise ValueError(
                        f"a multivariate Matérn-{p}+1/2 kernel does not admit {sum(a) + sum(b)} "
                        "derivative(s) in closed form (not enough differentiability)")
                continue
            for j, (fam, p, _) in enumerate(factors):
                if fam == MATERN_HALFINT and a[j] + b[j] > 2 * p:
                    raise ValueError(
                        f"a Matérn-{p}+1/2 factor is not {a[j] + b[j]} times differentiable "
                        "(mean-square sense); choose a smoother prior")
        groups.append({
            "d": d,
            "family": [f[0] for f in factors],
            "p": [f[1] for f in factors],
            "lengthscale": [f[2] for f in factors],
            "scale": scale,
            "terms": term_list,
        })
    return groups


class ProcessVectorCrossCovariance:
    """x -> Cov(f(x), L[f])  (`crosscov/_pv_crosscov.py:14-199`), L a point-evaluation
    functional, possibly composed with a differential operator."""

    def __init__(self, covfunc: CovarianceFunction, linfunctl, *, argnum: int = 1):
        if argnum not in (0, 1):
            raise ValueError("`argnum` must be 0 or 1")
        self._covfunc = covfunc
        self._linfunctl = linfunctl
        self._reverse = argnum == 0
        d = max(covfunc.input_size, 1)
        kL0, kL1 = covfunc._operator_coeffs()
        lc = linfunctl.coefficients_dict()
        if any(len(mi) != d for mi in lc):
            raise ValueError("functional and covariance function dimensions do not match")
        # the functional side always ends up on argument 1 of the stored kernel
        # (k(x, X) = k(X, x) for the symmetric priors handled here)
        if self._reverse:
            self._k = DifferentiatedCovarianceFunction(_base(covfunc), kL1, _compose(lc, kL0))
        else:
            self._k = DifferentiatedCovarianceFunction(_base(covfunc), kL0, _compose(lc, kL1))

    @property
    def covfunc(self):
        return self._covfunc

    @property
    def linfunctl(self):
        return self._linfunctl

    @property
    def reverse(self):
        return self._reverse

    @property
    def randproc_input_shape(self):
        return self._covfunc.input_shape

    @property
    def randvar_shape(self):
        return self._linfunctl.output_shape

    @property
    def randvar_size(self):
        return self._linfunctl.output_size

    def kernel(self) -> DifferentiatedCovarianceFunction:
        """(k L'^*) as a covariance function: argument 0 = x, argument 1 = the functional's points."""
        return self._k

    def __call__(self, x) -> np.ndarray:
        """Dense (batch..., N_obs) block, evaluated on the GPU."""
        X, batch = self._k._points(x)
        Xobs = self._linfunctl.points()
        K = self._k.matrix(X if self._k.input_ndim else X[:, 0], Xobs if self._k.input_ndim else Xobs[:, 0])
        return K.reshape(batch + (Xobs.shape[0],))


def apply_linfunctl_to_pv_crosscov(L, pv: ProcessVectorCrossCovariance):
    """`L(pv_crosscov)`: the covariance between two functionals of the process as a
    `LinearOperatorCovariance` (`crosscov/linfunctls/_evaluation.py:163-173`: `covfunc.linop(X)` when
    both functionals are the same evaluation, else `covfunc.linop(x, X)`).  The operator is the
    device-resident `KernelLinearOperator`: `.matrix` = `lpgp_kernel_matrix`, `.linop @ v` = matrix-free
    `lpgp_kernel_matvec`.  For `argnum=0` cross-covariances the new functional is the RIGHT variable."""
    from ... import randvars

    lc = L.coefficients_dict()
    kL0, kL1 = pv._k._operator_coeffs()
    k = DifferentiatedCovarianceFunction(_base(pv._k), _compose(lc, kL0), kL1)
    X_new, X_old = L.points(), pv.linfunctl.points()
    same = L is pv.linfunctl
    if k.input_ndim == 0:
        X_new, X_old = X_new[:, 0], X_old[:, 0]
    op = KernelLinearOperator(k, X_new, X_new if same else X_old)
    if pv.reverse:
        return randvars.LinearOperatorCovariance(op.T, pv.randvar_shape, L.output_shape)
    return randvars.LinearOperatorCovariance(op, L.output_shape, pv.randvar_shape)


def _base(k: CovarianceFunction) -> CovarianceFunction:
    return k._covfunc if isinstance(k, DifferentiatedCovarianceFunction) else k


__all__ = [
    "CovarianceFunction", "Matern", "ExpQuad", "TensorProduct", "ScaledCovarianceFunction",
    "SumCovarianceFunction", "Zero", "DifferentiatedCovarianceFunction",
    "ProcessVectorCrossCovariance", "apply_linfuncop", "apply_linfunctl_to_pv_crosscov", "lower_groups",
]
