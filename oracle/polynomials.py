"""Exact polynomial tables of the differentiated 1-D kernel factors (oracle; test-only).

Follows
  * `covfuncs/linfuncops/diffops/_matern.py:613-639`
    (`half_integer_matern_polynomial`, `half_integer_matern_derivative_polynomial`:
    recursion P_n = P'_{n-1} - P_{n-1}, kappa^{(n)}(s) = P_n(s) e^{-s}),
  * `functions/_polynomial.py:166-323` (`RationalPolynomial`: exact `Fraction`
    coefficients, `differentiate` :201-204, floor-division by a monomial :311-323),
  * probnum `Matern.half_integer_coefficients(p)` (third party, not in the tree; its
    published closed form c_k = p!/(2p)! * (2p-k)!/((p-k)! k!) * 2^k is restated here).

All arithmetic is exact (`fractions.Fraction`); conversion to fp64 happens once, at
evaluation time, exactly like `RationalPolynomial.coefficients` -> `np.double`.
"""

from __future__ import annotations

from fractions import Fraction
import functools
from math import factorial

import numpy as np


@functools.lru_cache(maxsize=None)
def matern_half_integer_coefficients(p: int) -> tuple[Fraction, ...]:
    """Coefficients c_0..c_p with kappa_{p+1/2}(s) = (sum_k c_k s^k) e^{-s}."""
    if p < 0:
        raise ValueError("p must be a non-negative integer")
    return tuple(
        Fraction(factorial(p), factorial(2 * p))
        * Fraction(factorial(2 * p - k), factorial(p - k) * factorial(k))
        * 2**k
        for k in range(p + 1)
    )


def _differentiate(c: tuple[Fraction, ...]) -> tuple[Fraction, ...]:
    return tuple(k * c[k] for k in range(1, len(c))) + (Fraction(0),)


@functools.lru_cache(maxsize=None)
def matern_derivative_polynomial(p: int, n: int) -> tuple[Fraction, ...]:
    """Coefficients (ascending) of P_n with d^n/ds^n [kappa(s)] = P_n(s) e^{-s}.

    `_matern.py:634-639`: P_0 = base polynomial, P_n = P_{n-1}' - P_{n-1}.
    The tuple always has p+1 entries (degree never grows).
    """
    if n == 0:
        return matern_half_integer_coefficients(p)
    prev = matern_derivative_polynomial(p, n - 1)
    d = _differentiate(prev)
    return tuple(dk - ck for dk, ck in zip(d, prev))


@functools.lru_cache(maxsize=None)
def hermite_polynomial(n: int) -> tuple[Fraction, ...]:
    """Probabilists' Hermite He_n (ascending coefficients): He_{n+1} = u He_n - He_n'.

    d^n/du^n exp(-u^2/2) = (-1)^n He_n(u) exp(-u^2/2); this is the closed form behind
    `diffops/_expquad.py:45-57,106-122,187-201,280-312,390-410`.
    """
    if n == 0:
        return (Fraction(1),)
    prev = hermite_polynomial(n - 1)
    shifted = (Fraction(0),) + prev  # u * He_{n-1}
    d = _differentiate(prev) + (Fraction(0),)
    d = d[: len(shifted)]
    d = d + (Fraction(0),) * (len(shifted) - len(d))
    return tuple(s - dk for s, dk in zip(shifted, d))


def horner(coeffs, x: np.ndarray) -> np.ndarray:
    """Horner evaluation in fp64, same loop as `functions/_polynomial.py:61-68`."""
    c = [float(ck) for ck in coeffs]
    res = np.full_like(x, c[-1], dtype=np.double)
    for ck in reversed(c[:-1]):
        res *= x
        res += ck
    return res
