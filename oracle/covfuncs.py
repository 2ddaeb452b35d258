"""NumPy restatement of the reference's differentiated covariance blocks (oracle; test-only).

What is restated (all paths relative to /root/reference/src/linpde_gp):

* 1-D half-integer Matérn factors  d^a/dx^a d^b/dx'^b k(x,x')
    `randprocs/covfuncs/linfuncops/diffops/_matern.py`
      :64-86   Identity x DirectionalDerivative   (a+b = 1)
      :300-318 DirectionalDerivative^2 (univariate) (a = b = 1)
      :403-410 Identity x WeightedLaplacian       (a+b = 2, one side)
      :476-483 WeightedLaplacian^2                (a = b = 2)
      :558-571 DirectionalDerivative x WeightedLaplacian (a+b = 3)
    all of which are instances of
        (-1)^b  a_s^{a+b}  sign(x-x')^{a+b}  P_{a+b}(s) e^{-s},   s = a_s |x-x'|,
    with a_s = sqrt(2 nu)/lengthscale (probnum `Matern._scale_factors`).
* 1-D ExpQuad factors `diffops/_expquad.py:45-57,106-122,187-201,280-312,390-410`
    k = exp(-(x-x')^2 / (2 l^2)); derivative = l^{-(a+b)} (-1)^a He_{a+b}(u) k, u=(x-x')/l.
* Tensor-product expansion `diffops/_tensor_product.py:22-70,84-119`:
    L0 k L1 = sum_{alpha in L0} sum_{beta in L1} c_alpha c'_beta prod_d d^{alpha_d} d'^{beta_d} k_d,
  evaluated exactly like `_compute_res`: every distinct 1-D factor once per dimension on
  broadcast (N0,1) x (1,N1) inputs, cached by (order0, order1), multiplied, accumulated.
* Multivariate ISOTROPIC half-integer Matérn k(x,x') = kappa(s), s = ||a_s * (x-x')||
  (probnum `Matern` with input_shape (d,)) with at most one directional derivative per
  argument, `diffops/_matern.py`
      :17-86    HalfIntegerMatern_Identity_DirectionalDerivative
                  (P_1 // s)(s) e^{-s} * <+-a_s*dir, a_s*(x-x')>
      :138-203  HalfIntegerMatern_DirectionalDerivative_DirectionalDerivative
                  [<a_s d0, a_s d1> (-P_1 // s)(s) - proj0 proj1 ((P_2 - P_1 // s) // s^2)(s)] e^{-s}
  (dispatch `diffops/_registry.py:141-187`).  Operators c + <v, grad> (identity part c, direction
  v) on either side are expanded bilinearly into those closed forms.
* scaled / sum kernels `covfuncs/_jax_arithmetic.py:16-66`,
  `covfuncs/linfuncops/_registry.py:14-31`.

A kernel is described by plain data (no classes shared with the product):

    kernel  = [(scale, [factor, ...]), ...]          # sum of scaled tensor products
    factor  = ("matern", nu, lengthscale) | ("expquad", lengthscale)
            | ("matern_iso", nu, lengthscales[d])   # then the ONLY entry of the factor list
    L       = {multi_index_tuple: coefficient}       # `PartialDerivativeCoefficients[()]`
                                                      # identity = {(0,)*d: 1.0}
"""

from __future__ import annotations

import numpy as np

from . import polynomials


def identity(d: int) -> dict:
    return {(0,) * d: 1.0}


def matern_factor(nu: float, lengthscale: float, n0: int, n1: int,
                  x0: np.ndarray, x1: np.ndarray) -> np.ndarray:
    """d^{n0}/dx0^{n0} d^{n1}/dx1^{n1} Matern_nu(x0, x1) on broadcastable 1-D inputs."""
    p = int(round(nu - 0.5))
    if abs(p + 0.5 - nu) > 1e-12 or p < 0:
        raise NotImplementedError("only half-integer Matérn has a closed form")
    n = n0 + n1
    a_s = np.sqrt(2.0 * nu) / lengthscale
    diffs = x0 - x1
    scaled_dists = a_s * np.abs(diffs)            # `_euclidean_distances(..., scale_factors)`
    poly = polynomials.matern_derivative_polynomial(p, n)
    res = polynomials.horner(poly, scaled_dists)  # polynomial part
    res *= np.exp(-scaled_dists)                  # exponential part
    if n % 2 == 1:                                # chain rule: a_s*(x-x') / s  == sign(x-x')
        res *= np.sign(diffs)
    res *= ((-1.0) ** n1) * a_s**n
    return res


def expquad_factor(lengthscale: float, n0: int, n1: int,
                   x0: np.ndarray, x1: np.ndarray) -> np.ndarray:
    """d^{n0}/dx0^{n0} d^{n1}/dx1^{n1} exp(-(x0-x1)^2/(2 l^2))."""
    n = n0 + n1
    u = (x0 - x1) / lengthscale
    he = polynomials.horner(polynomials.hermite_polynomial(n), u)
    return ((-1.0) ** n0) * lengthscale ** (-n) * he * np.exp(-0.5 * u * u)


def factor_eval(factor, n0: int, n1: int, x0: np.ndarray, x1: np.ndarray) -> np.ndarray:
    if factor[0] == "matern":
        return matern_factor(factor[1], factor[2], n0, n1, x0, x1)
    if factor[0] == "expquad":
        return expquad_factor(factor[1], n0, n1, x0, x1)
    raise ValueError(f"unknown factor {factor!r}")


def _floordiv_monomial(coeffs, k: int):
    """`RationalPolynomial // Monomial(k)` (`functions/_polynomial.py:311-323`): drop the k lowest
    coefficients (they must vanish for the quotient to be exact, as they do where the reference
    uses it)."""
    assert all(c == 0 for c in coeffs[:k]), "inexact monomial division"
    return tuple(coeffs[k:])


def _split_first_order(L: dict, d: int):
    """L = c + <v, grad>  ->  (c, v); anything of higher order has no closed form on the isotropic
    kernel (the reference falls back to JAX autodiff there)."""
    c, v = 0.0, np.zeros(d)
    for mi, coef in L.items():
        order = sum(mi)
        if order == 0:
            c += coef
        elif order == 1:
            v[mi.index(1)] += coef
        else:
            raise NotImplementedError("isotropic Matérn: only identity and directional derivatives")
    return c, v


def matern_iso_LkL(nu: float, lengthscales, L0: dict, L1: dict,
                   X0: np.ndarray, X1: np.ndarray | None) -> np.ndarray:
    """(L0 k L1)(X0, X1) for the isotropic Matérn; X1 None = diagonal (the `x1 is None` branches,
    `_matern.py:65-69,186-191`)."""
    p = int(round(nu - 0.5))
    X0 = np.asarray(X0, dtype=np.double)
    d = X0.shape[1]
    a_s = np.sqrt(2.0 * nu) / np.broadcast_to(np.asarray(lengthscales, dtype=np.double), (d,))
    c0, v0 = _split_first_order(L0, d)
    c1, v1 = _split_first_order(L1, d)
    P0 = polynomials.matern_derivative_polynomial(p, 0)
    first = bool(np.any(v0 != 0) or np.any(v1 != 0))
    second = bool(np.any(v0 != 0) and np.any(v1 != 0))
    if first:
        poly1 = _floordiv_monomial(polynomials.matern_derivative_polynomial(p, 1), 1)      # P_1 // s
    if second:
        neg_poly_deriv = tuple(-c for c in poly1)                                           # :164-166
        P2 = polynomials.matern_derivative_polynomial(p, 2)
        summed = tuple(a + (neg_poly_deriv[i] if i < len(neg_poly_deriv) else 0) for i, a in enumerate(P2))
        poly_diff = _floordiv_monomial(summed, 2)                                           # :168-171
        inprod = float(np.sum((a_s * v0) * (a_s * v1)))                                     # :185-187
    if X1 is None:
        val = c0 * c1 * float(P0[0])
        if second:
            val += inprod * float(neg_poly_deriv[0])                                        # :186-191
        return np.full(X0.shape[0], val)
    X1 = np.asarray(X1, dtype=np.double)
    scaled_diffs = (X0[:, None, :] - X1[None, :, :]) * a_s
    s = np.sqrt(np.sum(scaled_diffs**2, axis=-1))
    res = c0 * c1 * polynomials.horner(P0, s)
    if first:
        # argument 0 (`reverse=True`): +a_s*dir ; argument 1: -a_s*dir  (:58-65)
        w = c1 * (a_s * v0) - c0 * (a_s * v1)
        res = res + polynomials.horner(poly1, s) * np.sum(w * scaled_diffs, axis=-1)
    if second:
        proj0 = np.sum((a_s * v0) * scaled_diffs, axis=-1)
        proj1 = np.sum((a_s * v1) * scaled_diffs, axis=-1)
        res = res + inprod * polynomials.horner(neg_poly_deriv, s)
        res = res - proj0 * proj1 * polynomials.horner(poly_diff, s)
    return res * np.exp(-s)


def tensor_product_LkL(factors, L0: dict, L1: dict,
                       X0: np.ndarray, X1: np.ndarray) -> np.ndarray:
    """(L0 k L1)(X0, X1) for k = prod_d k_d, dense (N0, N1).

    Mirrors `TensorProduct_LinDiffOp_LinDiffOp._compute_res`
    (`diffops/_tensor_product.py:84-112`).
    """
    X0 = np.asarray(X0, dtype=np.double)
    X1 = np.asarray(X1, dtype=np.double)
    d = len(factors)
    assert X0.shape[1] == d and X1.shape[1] == d
    cache = [dict() for _ in range(d)]
    res = 0.0
    for mi0, c0 in L0.items():
        for mi1, c1 in L1.items():
            prod = None
            for dim in range(d):
                key = (mi0[dim], mi1[dim])
                if key not in cache[dim]:
                    cache[dim][key] = factor_eval(
                        factors[dim], key[0], key[1],
                        X0[:, None, dim], X1[None, :, dim],
                    )
                prod = cache[dim][key] if prod is None else prod * cache[dim][key]
            res = res + c0 * c1 * prod
    return res


def LkL(kernel, L0: dict, L1: dict, X0: np.ndarray, X1: np.ndarray | None = None) -> np.ndarray:
    """Dense block of (L0 k L1) for a sum of scaled tensor-product kernels."""
    if X1 is None:
        X1 = X0
    res = 0.0
    for scale, factors in kernel:
        if factors[0][0] == "matern_iso":
            res = res + scale * matern_iso_LkL(factors[0][1], factors[0][2], L0, L1, X0, X1)
        else:
            res = res + scale * tensor_product_LkL(factors, L0, L1, X0, X1)
    return res


def k_diag(kernel, L0: dict, L1: dict, X: np.ndarray) -> np.ndarray:
    """diag of (L0 k L1)(X, X) without forming the matrix (x1 is None shortcuts,
    `_matern.py:65-69,186-191,301-306`)."""
    X = np.asarray(X, dtype=np.double)
    out = np.zeros(X.shape[0])
    for scale, factors in kernel:
        if factors[0][0] == "matern_iso":
            out += scale * matern_iso_LkL(factors[0][1], factors[0][2], L0, L1, X, None)
            continue
        d = len(factors)
        for mi0, c0 in L0.items():
            for mi1, c1 in L1.items():
                prod = np.ones(X.shape[0])
                for dim in range(d):
                    prod = prod * factor_eval(factors[dim], mi0[dim], mi1[dim],
                                              X[:, dim], X[:, dim])
                out += scale * c0 * c1 * prod
    return out
