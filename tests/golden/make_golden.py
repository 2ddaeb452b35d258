#!/usr/bin/env python3
"""Generate the golden vectors in this directory.

The reference holds NO golden fixtures for this path and cannot be imported here
(probnum / jax / pykeops absent) -- SURVEY.md §8(c).  These vectors are therefore produced
by independent high-precision evaluation, not by the oracle under test:

* `kernel_blocks.npz`: differentiated 1-D Matern / ExpQuad factors and the 2-D Poisson and
  heat kernel blocks, by SymPy symbolic differentiation evaluated in 50-digit mpmath on
  seeded points (the reference's own `test_diffops.py:15-42` checks the same identities
  against JAX autodiff);
* `posterior_small.npz`: posterior mean / variance of a 1-D Poisson-Dirichlet problem
  (analytic solution sin(pi x)), solved in 50-digit mpmath (mpmath.cholesky_solve) from
  the mpmath kernel blocks.

* `matern_iso_blocks.npz`: the ISOTROPIC 3-D Matern (nu = 5/2, 7/2; per-dimension lengthscales)
  with identity / directional derivatives on either argument, again SymPy derivatives of the
  base kernel in 50-digit mpmath (the reference's cases `cases_matern.py:19-89` with its seeds
  for the directions).

* `posterior_noisy.npz`: the same posterior with observation noise (cond ~1e6): held to the plain 1e-8.

Run:  python tests/golden/make_golden.py      (about a minute; `... iso` / `... noisy` write that one file only)
"""
import os
import sys

import mpmath
import numpy as np
import sympy as sp

mpmath.mp.dps = 50
HERE = os.path.dirname(os.path.abspath(__file__))


def matern_exprs(p, a):
    """kappa_{p+1/2}(a |x - y|) on the two branches x>y, x<y (closed form of half-integer Matern)."""
    x, y = sp.symbols("x y", real=True)
    from math import factorial
    c = [sp.Rational(factorial(p), factorial(2 * p)) * sp.Rational(factorial(2 * p - k), factorial(p - k) * factorial(k)) * 2**k
         for k in range(p + 1)]
    out = []
    for s in (a * (x - y), a * (y - x)):
        out.append(sum(ck * s**k for k, ck in enumerate(c)) * sp.exp(-s))
    return x, y, out


def factor_fn(kind, param, n0, n1):
    """Returns f(x0, x1) -> mpf for d^{n0}/dx^{n0} d^{n1}/dy^{n1} of the 1-D factor."""
    if kind == "matern":
        nu, ell = param
        p = int(round(nu - 0.5))
        a = sp.sqrt(sp.Rational(2 * p + 1)) / sp.nsimplify(ell)
        x, y, (e_gt, e_lt) = matern_exprs(p, a)
        f_gt = sp.lambdify((x, y), sp.diff(e_gt, x, n0, y, n1), "mpmath")
        f_lt = sp.lambdify((x, y), sp.diff(e_lt, x, n0, y, n1), "mpmath")
        return lambda u, v: (f_gt if u > v else f_lt)(mpmath.mpf(float(u)), mpmath.mpf(float(v)))
    x, y = sp.symbols("x y", real=True)
    ell = sp.nsimplify(param)
    e = sp.exp(-(x - y) ** 2 / (2 * ell**2))
    f = sp.lambdify((x, y), sp.diff(e, x, n0, y, n1), "mpmath")
    return lambda u, v: f(mpmath.mpf(float(u)), mpmath.mpf(float(v)))


def block(factors, scale, L0, L1, X0, X1):
    """mpmath matrix of scale * sum_{a,b} c_a c_b prod_d d^a d^b k_d."""
    d = len(factors)
    fns = {}
    out = mpmath.zeros(X0.shape[0], X1.shape[0])
    for a, ca in L0.items():
        for b, cb in L1.items():
            for i in range(X0.shape[0]):
                for j in range(X1.shape[0]):
                    prod = mpmath.mpf(1)
                    for dim in range(d):
                        key = (dim, a[dim], b[dim])
                        if key not in fns:
                            fns[key] = factor_fn(*factors[dim], a[dim], b[dim])
                        prod *= fns[key](X0[i, dim], X1[j, dim])
                    out[i, j] += mpmath.mpf(scale) * ca * cb * prod
    return out


def make_iso():
    from math import factorial
    d = 3
    rng = np.random.default_rng(20240613)
    X0 = rng.uniform(-3, 3, size=(7, d))
    X1 = rng.uniform(-3, 3, size=(6, d))
    ls = [sp.Rational(7, 10), sp.Rational(1), sp.Rational(13, 10)]
    dir_a1 = 2.0 * np.random.default_rng(390852098).standard_normal(size=(d,))      # cases_matern.py:25-27
    dir_a0 = 2.0 * np.random.default_rng(4158976).standard_normal(size=(d,))        # :44-46
    r2 = np.random.default_rng(413598)                                              # :68-71
    dir0, dir1 = r2.standard_normal(size=(d,)), r2.standard_normal(size=(d,))
    out = {"X0": X0, "X1": X1, "lengthscales": np.array([float(v) for v in ls]),
           "dir_arg1": dir_a1, "dir_arg0": dir_a0, "dir0": dir0, "dir1": dir1}
    xs = sp.symbols(f"x0:{d}", real=True)
    ys = sp.symbols(f"y0:{d}", real=True)

    def dd(expr, v, vars_):
        return sum(sp.Float(float(vj), 60) * sp.diff(expr, xj) for vj, xj in zip(v, vars_))

    for nu2 in (5, 7):
        p = (nu2 - 1) // 2
        c = [sp.Rational(factorial(p), factorial(2 * p)) * sp.Rational(factorial(2 * p - k), factorial(p - k) * factorial(k)) * 2**k
             for k in range(p + 1)]
        s_ = sp.sqrt(sum((sp.sqrt(nu2) * (xi - yi) / l) ** 2 for xi, yi, l in zip(xs, ys, ls)))
        k = sum(ck * s_**i for i, ck in enumerate(c)) * sp.exp(-s_)
        exprs = {"k": k, "k_dd": dd(k, dir_a1, ys), "dd_k": dd(k, dir_a0, xs), "dd_k_dd": dd(dd(k, dir0, xs), dir1, ys)}
        for name, e in exprs.items():
            f = sp.lambdify((*xs, *ys), e, "mpmath")
            out[f"matern{nu2}2_{name}"] = np.array(
                [[float(f(*[mpmath.mpf(float(v)) for v in (*a, *b)])) for b in X1] for a in X0])
    np.savez(os.path.join(HERE, "matern_iso_blocks.npz"), **out)
    print("wrote matern_iso_blocks.npz")


def to_np(M):
    return np.array([[float(M[i, j]) for j in range(M.cols)] for i in range(M.rows)])


def main():
    if sys.argv[1:] == ["iso"]:
        return make_iso()
    make_iso()
    rng = np.random.default_rng(20240612)
    out = {}
    # ---- 1-D factors, off-grid points (x != y everywhere) ----
    x0 = rng.uniform(-3, 3, size=(9, 1))
    x1 = rng.uniform(-3, 3, size=(7, 1))
    out["x0_1d"], out["x1_1d"] = x0, x1
    for nu in (1.5, 2.5, 3.5):
        p = int(nu - 0.5)
        for n0 in range(3):
            for n1 in range(3):
                if n0 + n1 > 2 * p:
                    continue
                M = block([("matern", (nu, 0.7))], 1.0, {(n0,): 1}, {(n1,): 1}, x0, x1)
                out[f"matern{int(2*nu)}2_l0.7_{n0}{n1}"] = to_np(M)
    for n0 in range(3):
        for n1 in range(3):
            M = block([("expquad", 0.25)], 1.0, {(n0,): 1}, {(n1,): 1}, x0 / 3, x1 / 3)
            out[f"expquad_l0.25_{n0}{n1}"] = to_np(M)
    # ---- 2-D Poisson (c3 kernel) and heat (c5 kernel) blocks ----
    X0 = rng.uniform(-1, 1, size=(6, 2))
    X1 = rng.uniform(-1, 1, size=(5, 2))
    out["X0_2d"], out["X1_2d"] = X0, X1
    pois = [("matern", (2.5, 1.0)), ("matern", (2.5, 1.0))]
    lap = {(2, 0): -1, (0, 2): -1}
    ident = {(0, 0): 1}
    out["poisson_LkL"] = to_np(block(pois, 4.0, lap, lap, X0, X1))
    out["poisson_kL"] = to_np(block(pois, 4.0, ident, lap, X0, X1))
    out["poisson_k"] = to_np(block(pois, 4.0, ident, ident, X0, X1))
    heat_k = [("matern", (1.5, 2.5)), ("matern", (2.5, 2.0))]
    Xh0 = np.column_stack([rng.uniform(0, 5, 6), rng.uniform(-1, 1, 6)])
    Xh1 = np.column_stack([rng.uniform(0, 5, 5), rng.uniform(-1, 1, 5)])
    out["Xh0"], out["Xh1"] = Xh0, Xh1
    heat = {(1, 0): 1, (0, 2): sp.Rational(-1, 10)}
    out["heat_LkL"] = to_np(block(heat_k, 1.0, heat, heat, Xh0, Xh1))
    out["heat_Lk"] = to_np(block(heat_k, 1.0, heat, ident, Xh0, Xh1))
    np.savez(os.path.join(HERE, "kernel_blocks.npz"), **out)

    # ---- small 1-D Poisson-Dirichlet posterior in 50-digit arithmetic ----
    n = 14
    Xp = np.linspace(-0.95, 0.95, n)[:, None]
    Xb = np.array([[-1.0], [1.0]])
    Xt = np.linspace(-0.9, 0.9, 7)[:, None] + 0.0123
    fac = [("matern", (2.5, 1.0))]
    lap1 = {(2,): -1}
    id1 = {(0,): 1}
    Gbb = block(fac, 4.0, id1, id1, Xb, Xb)
    Gpb = block(fac, 4.0, lap1, id1, Xp, Xb)
    Gpp = block(fac, 4.0, lap1, lap1, Xp, Xp)
    N = 2 + n
    G = mpmath.zeros(N, N)
    for i in range(2):
        for j in range(2):
            G[i, j] = Gbb[i, j]
    for i in range(n):
        for j in range(2):
            G[2 + i, j] = Gpb[i, j]
            G[j, 2 + i] = Gpb[i, j]
        for j in range(n):
            G[2 + i, 2 + j] = Gpp[i, j]
    y = mpmath.matrix([0, 0] + [mpmath.pi**2 * mpmath.sin(mpmath.pi * mpmath.mpf(float(v))) for v in Xp[:, 0]])
    w = mpmath.cholesky_solve(G, y)
    Ktb = block(fac, 4.0, id1, id1, Xt, Xb)
    Ktp = block(fac, 4.0, id1, lap1, Xt, Xp)
    mean, var = [], []
    for i in range(Xt.shape[0]):
        krow = mpmath.matrix([Ktb[i, 0], Ktb[i, 1]] + [Ktp[i, j] for j in range(n)])
        mean.append(float((krow.T * w)[0]))
        z = mpmath.cholesky_solve(G, krow)
        var.append(float(mpmath.mpf(4) - (krow.T * z)[0]))
    np.savez(os.path.join(HERE, "posterior_small.npz"), Xp=Xp, Xb=Xb, Xt=Xt, Yp=np.array([float(v) for v in y[2:]]),
             mean=np.array(mean), var=np.array(var), weights=np.array([float(v) for v in w]))
    print("wrote kernel_blocks.npz, posterior_small.npz")


def posterior_noisy():
    """`posterior_noisy.npz` (round 4): the 1-D Poisson-Dirichlet problem of `posterior_small.npz` with observation NOISE --
    variance 1e-6 on the two boundary values, 1e-3 on the 14 PDE observations -- again solved in 50-digit mpmath.  With
    the noise cond(G) drops from ~1e9 to ~1e6, so an fp64 path must reproduce these vectors to the plain 1e-8 criterion
    (posterior_small is held to 1e-7: cond(G) eps, see tests/test_gpu_golden.py)."""
    n = 14
    Xp = np.linspace(-0.95, 0.95, n)[:, None]
    Xb = np.array([[-1.0], [1.0]])
    Xt = np.linspace(-0.9, 0.9, 7)[:, None] + 0.0123
    fac = [("matern", (2.5, 1.0))]
    lap1, id1 = {(2,): -1}, {(0,): 1}
    nb, npde = mpmath.mpf("1e-6"), mpmath.mpf("1e-3")
    Gbb = block(fac, 4.0, id1, id1, Xb, Xb)
    Gpb = block(fac, 4.0, lap1, id1, Xp, Xb)
    Gpp = block(fac, 4.0, lap1, lap1, Xp, Xp)
    N = 2 + n
    G = mpmath.zeros(N, N)
    for i in range(2):
        for j in range(2):
            G[i, j] = Gbb[i, j] + (nb if i == j else 0)
    for i in range(n):
        for j in range(2):
            G[2 + i, j] = Gpb[i, j]
            G[j, 2 + i] = Gpb[i, j]
        for j in range(n):
            G[2 + i, 2 + j] = Gpp[i, j] + (npde if i == j else 0)
    y = mpmath.matrix([0, 0] + [mpmath.pi**2 * mpmath.sin(mpmath.pi * mpmath.mpf(float(v))) for v in Xp[:, 0]])
    w = mpmath.cholesky_solve(G, y)
    Ktb = block(fac, 4.0, id1, id1, Xt, Xb)
    Ktp = block(fac, 4.0, id1, lap1, Xt, Xp)
    mean, var = [], []
    for i in range(Xt.shape[0]):
        krow = mpmath.matrix([Ktb[i, 0], Ktb[i, 1]] + [Ktp[i, j] for j in range(n)])
        mean.append(float((krow.T * w)[0]))
        z = mpmath.cholesky_solve(G, krow)
        var.append(float(mpmath.mpf(4) - (krow.T * z)[0]))
    np.savez(os.path.join(HERE, "posterior_noisy.npz"), Xp=Xp, Xb=Xb, Xt=Xt, Yp=np.array([float(v) for v in y[2:]]),
             noise_b=float(nb), noise_p=float(npde), mean=np.array(mean), var=np.array(var), weights=np.array([float(v) for v in w]))
    print("wrote posterior_noisy.npz")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "noisy":
        sys.exit(posterior_noisy())
    sys.exit(main())
