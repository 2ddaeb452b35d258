"""Pin the oracle's kernel formulas against independent symbolic differentiation.

Mirrors `tests/linpde_gp/randprocs/kernels/linfuncops/diffops/test_diffops.py:30-42`
of the reference (closed form vs autodiff of the base kernel), with SymPy + 50-digit
mpmath in place of JAX (not installed here).
"""
from fractions import Fraction

import mpmath
import numpy as np
import pytest
import sympy as sp

from oracle import covfuncs, polynomials

mpmath.mp.dps = 50


def test_matern_base_coefficients():
    # SURVEY.md §8(a) A1 / probnum Matern.half_integer_coefficients
    assert polynomials.matern_half_integer_coefficients(0) == (1,)
    assert polynomials.matern_half_integer_coefficients(1) == (1, 1)
    assert polynomials.matern_half_integer_coefficients(2) == (1, 1, Fraction(1, 3))
    assert polynomials.matern_half_integer_coefficients(3) == (
        1, 1, Fraction(2, 5), Fraction(1, 15))


def test_matern_derivative_polynomials_known_values():
    P = polynomials.matern_derivative_polynomial
    F = Fraction
    # p = 1
    assert [P(1, n) for n in range(5)] == [(1, 1), (0, -1), (-1, 1), (2, -1), (-3, 1)]
    # p = 2 (nu = 5/2), the table behind the Poisson configs
    assert P(2, 1) == (0, F(-1, 3), F(-1, 3))
    assert P(2, 2) == (F(-1, 3), F(-1, 3), F(1, 3))
    assert P(2, 3) == (0, 1, F(-1, 3))
    assert P(2, 4) == (1, F(-5, 3), F(1, 3))
    # p = 3
    assert P(3, 2) == (F(-1, 5), F(-1, 5), 0, F(1, 15))
    assert P(3, 4) == (F(1, 5), F(1, 5), F(-2, 5), F(1, 15))


def test_hermite():
    H = polynomials.hermite_polynomial
    assert H(0) == (1,)
    assert H(1) == (0, 1)
    assert H(2) == (-1, 0, 1)
    assert H(3) == (0, -3, 0, 1)
    assert H(4) == (3, 0, -6, 0, 1)


def _sympy_matern(p, a):
    """(expr for x>y, expr for x<y) of kappa_p(a|x-y|)."""
    x, y = sp.symbols("x y", real=True)
    c = polynomials.matern_half_integer_coefficients(p)
    out = []
    for s in (a * (x - y), a * (y - x)):
        out.append(sum(sp.Rational(ck.numerator, ck.denominator) * s**k
                       for k, ck in enumerate(c)) * sp.exp(-s))
    return x, y, out


@pytest.mark.parametrize("p", [1, 2, 3, 4])
def test_matern_factor_vs_sympy(p):
    nu = p + 0.5
    ell = 0.7
    a_exact = sp.sqrt(2 * sp.Rational(2 * p + 1, 2)) / sp.Rational(7, 10)
    x, y, (e_gt, e_lt) = _sympy_matern(p, a_exact)
    rng = np.random.default_rng(390852098 + p)
    x0 = rng.uniform(-3, 3, size=24)
    x1 = rng.uniform(-3, 3, size=24)
    # mean-square differentiability: total order <= 2p
    orders = [(n0, n1) for n0 in range(0, 3) for n1 in range(0, 3) if n0 + n1 <= 2 * p]
    for n0, n1 in orders:
        got = covfuncs.matern_factor(nu, ell, n0, n1, x0, x1)
        f_gt = sp.lambdify((x, y), sp.diff(e_gt, x, n0, y, n1), "mpmath")
        f_lt = sp.lambdify((x, y), sp.diff(e_lt, x, n0, y, n1), "mpmath")
        ref = np.array([
            float((f_gt if xi > yi else f_lt)(mpmath.mpf(float(xi)), mpmath.mpf(float(yi))))
            for xi, yi in zip(x0, x1)
        ])
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-14,
                                   err_msg=f"p={p} orders=({n0},{n1})")


def test_expquad_factor_vs_sympy():
    ell = 0.25
    x, y = sp.symbols("x y", real=True)
    e = sp.exp(-(x - y) ** 2 / (2 * sp.Rational(1, 4) ** 2))
    rng = np.random.default_rng(4158)
    x0 = rng.uniform(-1, 1, size=24)
    x1 = rng.uniform(-1, 1, size=24)
    for n0 in range(3):
        for n1 in range(3):
            f = sp.lambdify((x, y), sp.diff(e, x, n0, y, n1), "mpmath")
            ref = np.array([float(f(mpmath.mpf(float(a)), mpmath.mpf(float(b))))
                            for a, b in zip(x0, x1)])
            got = covfuncs.expquad_factor(ell, n0, n1, x0, x1)
            np.testing.assert_allclose(got, ref, rtol=1e-11, atol=1e-12 * max(1, np.abs(ref).max()))


def test_poisson2d_LkL_closed_form():
    """sigma^2 alpha^2 e^{-(s1+s2)}[a^4 P4 P0 + 2 a^4 P2 P2 + a^4 P0 P4]."""
    rng = np.random.default_rng(24)
    X0 = rng.uniform(-1, 1, size=(17, 2))
    X1 = rng.uniform(-1, 1, size=(13, 2))
    kernel = [(4.0, [("matern", 2.5, 1.0), ("matern", 2.5, 1.0)])]
    lap = {(2, 0): -1.0, (0, 2): -1.0}
    got = covfuncs.LkL(kernel, lap, lap, X0, X1)
    a = np.sqrt(5.0)
    s1 = a * np.abs(X0[:, None, 0] - X1[None, :, 0])
    s2 = a * np.abs(X0[:, None, 1] - X1[None, :, 1])
    P = lambda n, s: polynomials.horner(polynomials.matern_derivative_polynomial(2, n), s)
    ref = 4.0 * np.exp(-(s1 + s2)) * (
        a**4 * P(4, s1) * P(0, s2) + 2 * a**4 * P(2, s1) * P(2, s2) + a**4 * P(0, s1) * P(4, s2))
    np.testing.assert_allclose(got, ref, rtol=1e-13, atol=1e-13)
    cross = covfuncs.LkL(kernel, covfuncs.identity(2), lap, X0, X1)
    ref_c = -4.0 * np.exp(-(s1 + s2)) * (a**2 * P(2, s1) * P(0, s2) + a**2 * P(0, s1) * P(2, s2))
    np.testing.assert_allclose(cross, ref_c, rtol=1e-13, atol=1e-13)


def test_tensor_product_expquad_equals_multivariate():
    # tests/linpde_gp/randprocs/kernels/test_tensor_product.py:39-47
    rng = np.random.default_rng(3)
    X = rng.normal(size=(20, 3))
    ls = np.array([0.4, 1.3, 0.9])
    kernel = [(1.0, [("expquad", l) for l in ls])]
    got = covfuncs.LkL(kernel, covfuncs.identity(3), covfuncs.identity(3), X)
    D = (X[:, None, :] - X[None, :, :]) / ls
    np.testing.assert_allclose(got, np.exp(-0.5 * np.sum(D * D, axis=-1)), rtol=1e-13)


# ---- isotropic multivariate Matérn with directional derivatives -------------------------------
# the reference's cases `cases_matern.py:19-89` (input_shape (3,), its seeds for the directions)
# checked the way `test_diffops.py:30-42` does (closed form vs derivative of the base kernel)
def _sympy_matern_iso(p, a, d):
    xs = sp.symbols(f"x0:{d}", real=True)
    ys = sp.symbols(f"y0:{d}", real=True)
    s = sp.sqrt(sum((a * (xi - yi)) ** 2 for xi, yi in zip(xs, ys)))
    c = polynomials.matern_half_integer_coefficients(p)
    k = sum(sp.Rational(ck.numerator, ck.denominator) * s**i for i, ck in enumerate(c)) * sp.exp(-s)
    return xs, ys, k


@pytest.mark.parametrize("p", [1, 2, 3, 4])
def test_matern_iso_vs_sympy(p):
    d = 3
    nu = p + 0.5
    a_exact = sp.sqrt(2 * sp.Rational(2 * p + 1, 2))          # lengthscale 1, as in the reference's cases
    xs, ys, k = _sympy_matern_iso(p, a_exact, d)
    X0 = np.random.default_rng(109134809 + d).uniform(-3, 3, size=(12, d))
    X1 = np.random.default_rng(1 + d).uniform(-3, 3, size=(9, d))
    dir_a1 = 2.0 * np.random.default_rng(390852098).standard_normal(size=(d,))     # cases_matern.py:25-27
    dir_a0 = 2.0 * np.random.default_rng(4158976).standard_normal(size=(d,))       # :44-46
    rng = np.random.default_rng(413598)                                            # :68-71
    dir0, dir1 = rng.standard_normal(size=(d,)), rng.standard_normal(size=(d,))
    ident = covfuncs.identity(d)

    def dd(v):
        return {tuple(int(i == j) for i in range(d)): float(v[j]) for j in range(d)}

    def sym_apply(expr, v, vars_):
        return sum(sp.Float(float(vj), 60) * sp.diff(expr, xj) for vj, xj in zip(v, vars_))

    cases = [("id x dd", ident, dd(dir_a1), sym_apply(k, dir_a1, ys)),
             ("dd x id", dd(dir_a0), ident, sym_apply(k, dir_a0, xs))]
    if p >= 2:
        cases.append(("dd x dd", dd(dir0), dd(dir1), sym_apply(sym_apply(k, dir0, xs), dir1, ys)))
        # identity + direction on both sides (what a Robin-type functional gives)
        L0 = {**dd(dir0), (0,) * d: 0.7}
        L1 = {**dd(dir1), (0,) * d: -1.3}
        e0 = sp.Float(0.7, 60) * k + sym_apply(k, dir0, xs)
        cases.append(("robin", L0, L1, sp.Float(-1.3, 60) * e0 + sym_apply(e0, dir1, ys)))
    kernel = [(1.0, [("matern_iso", nu, np.ones(d))])]
    for name, L0, L1, expr in cases:
        got = covfuncs.LkL(kernel, L0, L1, X0, X1)
        f = sp.lambdify((*xs, *ys), expr, "mpmath")
        ref = np.array([[float(f(*[mpmath.mpf(float(v)) for v in (*x, *y)])) for y in X1] for x in X0])
        np.testing.assert_allclose(got, ref, rtol=1e-11, atol=1e-13, err_msg=f"p={p} {name}")


def test_matern_iso_diagonal_and_lengthscales():
    # x1 is None branches (`_matern.py:65-69,186-191`) == limit of the dense block; ARD lengthscales
    d, nu = 3, 2.5
    ls = np.array([0.6, 1.1, 2.0])
    kernel = [(1.7, [("matern_iso", nu, ls)])]
    rng = np.random.default_rng(5)
    X = rng.uniform(-1, 1, size=(7, d))
    v0, v1 = rng.standard_normal(d), rng.standard_normal(d)
    L0 = {tuple(int(i == j) for i in range(d)): float(v0[j]) for j in range(d)}
    L1 = {tuple(int(i == j) for i in range(d)): float(v1[j]) for j in range(d)}
    for A, B in [(covfuncs.identity(d), covfuncs.identity(d)), (L0, covfuncs.identity(d)), (L0, L1)]:
        dense = covfuncs.LkL(kernel, A, B, X, X)
        np.testing.assert_allclose(covfuncs.k_diag(kernel, A, B, X), np.diag(dense), rtol=1e-13, atol=1e-15)
    # d = 1 isotropic == the univariate factor
    X1 = rng.uniform(-1, 1, size=(9, 1))
    got = covfuncs.LkL([(1.0, [("matern_iso", nu, np.array([0.8]))])], {(1,): 1.0}, {(1,): 1.0}, X1, X1)
    ref = covfuncs.LkL([(1.0, [("matern", nu, 0.8)])], {(1,): 1.0}, {(1,): 1.0}, X1, X1)
    np.testing.assert_allclose(got, ref, rtol=1e-13, atol=1e-14)


@pytest.mark.parametrize("p", [0, 1, 2, 3, 4, 5])
def test_matern_base_coefficients_vs_bessel_definition(p):
    """`matern_half_integer_coefficients` restates probnum's `Matern.half_integer_coefficients` (third
    party, absent from the tree) from its published closed form.  Pin it to the DEFINITION of the Matern
    covariance, k_nu(s) = 2^{1-nu} / Gamma(nu) * s^nu * K_nu(s) with s = sqrt(2 nu) r / l (modified Bessel
    function of the second kind), in 50-digit arithmetic: removes the assumption shared by `oracle/` and
    `tests/golden/make_golden.py`."""
    nu = mpmath.mpf(2 * p + 1) / 2
    c = [mpmath.mpf(ck.numerator) / mpmath.mpf(ck.denominator)
         for ck in polynomials.matern_half_integer_coefficients(p)]
    for s in [mpmath.mpf(v) for v in ("0.001", "0.01", "0.1", "0.37", "1", "2.5", "7", "19", "40")]:
        closed = sum(ck * s**k for k, ck in enumerate(c)) * mpmath.exp(-s)
        bessel = mpmath.mpf(2) ** (1 - nu) / mpmath.gamma(nu) * s**nu * mpmath.besselk(nu, s)
        assert abs(closed - bessel) <= mpmath.mpf(10) ** (-40) * abs(bessel), (p, s)
    # k_nu(0) = 1
    assert c[0] == 1
