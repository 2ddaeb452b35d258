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
* scaled / sum kernels `covfuncs/_jax_arithmetic.py:16-66`,
  `covfuncs/linfuncops/_registry.py:14-31`.

A kernel is described by plain data (no classes shared with the product):

    kernel  = [(scale, [factor, ...]), ...]          # sum of scaled tensor products
    factor  = ("matern", nu, lengthscale) | ("expquad", lengthscale)
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
        res = res + scale * tensor_product_LkL(factors, L0, L1, X0, X1)
    return res


def k_diag(kernel, L0: dict, L1: dict, X: np.ndarray) -> np.ndarray:
    """diag of (L0 k L1)(X, X) without forming the matrix (x1 is None shortcuts,
    `_matern.py:65-69,186-191,301-306`)."""
    X = np.asarray(X, dtype=np.double)
    out = np.zeros(X.shape[0])
    for scale, factors in kernel:
        d = len(factors)
        for mi0, c0 in L0.items():
            for mi1, c1 in L1.items():
                prod = np.ones(X.shape[0])
                for dim in range(d):
                    prod = prod * factor_eval(factors[dim], mi0[dim], mi1[dim],
                                              X[:, dim], X[:, dim])
                out += scale * c0 * c1 * prod
    return out
