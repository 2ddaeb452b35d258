#!/usr/bin/env python3
"""Golden vectors for the MULTI-BLOCK conditioning algebra (round 6).

`posterior_small.npz` / `posterior_noisy.npz` (make_golden.py) pin a 1-D problem with two observation blocks.  The blocks the
BASELINE configurations are made of -- four boundary blocks with a nugget plus a PDE block (c3), an initial condition, two
boundary conditions, heat-operator collocation and noisy interior values (c5), Neumann blocks on an isotropic Matern prior
(`experiments/cpu.py:214-229`) -- were pinned only by the oracle agreeing with the device.  Here each of them is solved
INDEPENDENTLY of both: kernel entries by SymPy differentiation of the base kernel evaluated in 50-digit mpmath (make_golden.py's
`block`; the isotropic kernel from its closed form in the distance), the Gram matrix assembled block by block in mpmath
(`_conditional.py:253-294`: rows L_i k L_j'*, noise on the diagonal blocks, `:392-394`), representer weights and posterior by
`mpmath.cholesky_solve` at 50 digits (`_conditional.py:44,193-197,223-231`).

  posterior_multiblock.npz
    poisson2d_*   prior 4 M52(l=1) x M52(l=1), L = -Laplace; blocks: 4 edges x 4 value observations (nugget 1e-8, values 0),
                  then 4 x 4 collocation points with f = 2  (c3's algebra at N = 32)
    heat_*        prior M32(l_t=2.5) x M52(l_x=2), L = d_t - 0.1 d_xx; blocks: initial condition (5 values, nugget 1e-6), two
                  boundary conditions (4 values each, noise 1e-5), 4 x 3 collocation points (rhs 0), 4 noisy interior VALUES
                  (noise 1e-4)  (c5's algebra at N = 29)
    neumann_*     prior 1.5^2 isotropic Matern-5/2 (l = 0.9, 0.7) in 2-D; blocks: 9 noisy values (1e-4), then 5 Neumann
                  observations -0.8 <n, grad u> at boundary points (noise 1e-4)
    expquad2d_*   prior 1.7 ExpQuad(l=0.6) x ExpQuad(l=0.8), L = -Laplace: c3's block structure on the OTHER kernel family of the
                  path (`diffops/_expquad.py:12-433`): 4 edges x 4 values (noise 1e-6), then 4 x 4 collocation points, f = 2 (noise 1e-6)
  For every problem: the point sets, right-hand sides, noise levels, cond_2 of the fp64 Gram matrix, representer weights,
  posterior mean and variance at the test points.

Run:  python tests/golden/make_golden_multiblock.py [problem ...]     (a few minutes: ~1e5 SymPy-lambdified kernel entries at 50 digits)
"""
import os
import sys

import mpmath
import numpy as np
import sympy as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import block  # noqa: E402  (50-digit kernel blocks of product kernels)

mpmath.mp.dps = 50


def solve_posterior(blocks_LkL, noises, ys, cross_fn, kxx_fn, Xt):
    """blocks_LkL(i, j) -> mpmath matrix of the Gram block (i >= j); noises[i]: scalar variance of block i; ys[i]: values.
    cross_fn(i) -> mpmath matrix (n_test x n_i) of (k L_i'*)(x, X_i); kxx_fn(t) -> prior variance at test point t."""
    sizes = [len(y) for y in ys]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    N = int(offs[-1])
    G = mpmath.zeros(N, N)
    for i in range(len(ys)):
        for j in range(i + 1):
            B = blocks_LkL(i, j)
            for a in range(sizes[i]):
                for b in range(sizes[j]):
                    G[int(offs[i]) + a, int(offs[j]) + b] = B[a, b]
                    G[int(offs[j]) + b, int(offs[i]) + a] = B[a, b]
        for a in range(sizes[i]):
            G[int(offs[i]) + a, int(offs[i]) + a] += mpmath.mpf(noises[i])
    y = mpmath.matrix([mpmath.mpf(float(v)) for yy in ys for v in yy])
    w = mpmath.cholesky_solve(G, y)
    K = [cross_fn(i) for i in range(len(ys))]
    mean, var = [], []
    for t in range(Xt.shape[0]):
        krow = mpmath.matrix([K[i][t, a] for i in range(len(ys)) for a in range(sizes[i])])
        mean.append(float((krow.T * w)[0]))
        z = mpmath.cholesky_solve(G, krow)
        var.append(float(kxx_fn(t) - (krow.T * z)[0]))
    Gf = np.array([[float(G[i, j]) for j in range(N)] for i in range(N)])
    return np.array([float(v) for v in w]), np.array(mean), np.array(var), float(np.linalg.cond(Gf))


def poisson2d():
    fac = [("matern", (2.5, 1.0)), ("matern", (2.5, 1.0))]
    scale = 4.0
    ident, lap = {(0, 0): 1}, {(2, 0): -1, (0, 2): -1}
    s = np.linspace(-1.0 + 1e-6, 1.0 - 1e-6, 4)               # points along an edge, inset as notebook 0001 cell 13
    edges = [np.column_stack([np.full(4, -1.0), s]), np.column_stack([np.full(4, 1.0), s]),
             np.column_stack([s, np.full(4, -1.0)]), np.column_stack([s, np.full(4, 1.0)])]
    g = np.linspace(-0.75, 0.75, 4)
    Xp = np.array([[a, b] for a in g for b in g])               # meshgrid(indexing="ij") flattened C-order
    Xs = edges + [Xp]
    Ls = [ident] * 4 + [lap]
    ys = [np.zeros(4)] * 4 + [np.full(16, 2.0)]
    noises = ["1e-8"] * 4 + ["0"]
    Xt = np.array([[0.0, 0.0], [0.31, -0.42], [-0.55, 0.6], [0.8, 0.15], [-0.2, -0.85], [0.47, 0.53]])
    w, mean, var, cond = solve_posterior(
        lambda i, j: block(fac, scale, Ls[i], Ls[j], Xs[i], Xs[j]), noises, ys,
        lambda i: block(fac, scale, ident, Ls[i], Xt, Xs[i]), lambda t: mpmath.mpf(scale), Xt)
    out = {"poisson2d_Xt": Xt, "poisson2d_weights": w, "poisson2d_mean": mean, "poisson2d_var": var, "poisson2d_cond": cond,
           "poisson2d_noise": np.array([float(v) for v in noises])}
    for i, X in enumerate(Xs):
        out[f"poisson2d_X{i}"] = X
        out[f"poisson2d_Y{i}"] = ys[i]
    print(f"poisson2d: N = {sum(len(y) for y in ys)}, cond_2(G) = {cond:.2e}")
    return out


def expquad2d():
    fac = [("expquad", 0.6), ("expquad", 0.8)]
    scale = 1.7
    ident, lap = {(0, 0): 1}, {(2, 0): -1, (0, 2): -1}
    s = np.linspace(-1.0 + 1e-6, 1.0 - 1e-6, 4)
    edges = [np.column_stack([np.full(4, -1.0), s]), np.column_stack([np.full(4, 1.0), s]),
             np.column_stack([s, np.full(4, -1.0)]), np.column_stack([s, np.full(4, 1.0)])]
    g = np.linspace(-0.7, 0.7, 4)
    Xp = np.array([[a, b] for a in g for b in g])
    Xs = edges + [Xp]
    Ls = [ident] * 4 + [lap]
    ys = [np.zeros(4)] * 4 + [np.full(16, 2.0)]
    noises = ["1e-6"] * 5
    Xt = np.array([[0.0, 0.0], [0.31, -0.42], [-0.55, 0.6], [0.8, 0.15], [-0.2, -0.85], [0.47, 0.53]])
    w, mean, var, cond = solve_posterior(
        lambda i, j: block(fac, scale, Ls[i], Ls[j], Xs[i], Xs[j]), noises, ys,
        lambda i: block(fac, scale, ident, Ls[i], Xt, Xs[i]), lambda t: mpmath.mpf(scale), Xt)
    out = {"expquad2d_Xt": Xt, "expquad2d_weights": w, "expquad2d_mean": mean, "expquad2d_var": var, "expquad2d_cond": cond,
           "expquad2d_noise": np.array([float(v) for v in noises])}
    for i, X in enumerate(Xs):
        out[f"expquad2d_X{i}"] = X
        out[f"expquad2d_Y{i}"] = ys[i]
    print(f"expquad2d: N = {sum(len(y) for y in ys)}, cond_2(G) = {cond:.2e}")
    return out


def heat():
    fac = [("matern", (1.5, 2.5)), ("matern", (2.5, 2.0))]
    scale = 1.0
    ident, op = {(0, 0): 1}, {(1, 0): 1, (0, 2): sp.Rational(-1, 10)}
    xi = np.linspace(-0.8, 0.8, 5)
    Xic = np.column_stack([np.zeros(5), xi])
    tb = np.array([0.5, 1.7, 3.1, 4.4])
    Xb0, Xb1 = np.column_stack([tb, np.full(4, -1.0)]), np.column_stack([tb, np.full(4, 1.0)])
    Xc = np.array([[t, x] for t in (0.6, 1.9, 3.3, 4.6) for x in (-0.5, 0.05, 0.55)])
    Xv = np.array([[1.1, -0.3], [2.4, 0.4], [3.7, -0.6], [4.2, 0.2]])
    Xs = [Xic, Xb0, Xb1, Xc, Xv]
    Ls = [ident, ident, ident, op, ident]
    ys = [np.sin(np.pi * xi), np.zeros(4), np.zeros(4), np.zeros(12),
          np.array([np.exp(-0.1 * np.pi**2 * t) * np.sin(np.pi * x) for t, x in Xv]) + np.array([0.004, -0.007, 0.002, 0.005])]
    noises = ["1e-6", "1e-5", "1e-5", "0", "1e-4"]
    Xt = np.array([[0.3, 0.1], [1.4, -0.45], [2.8, 0.7], [3.9, -0.1], [4.8, 0.33]])
    w, mean, var, cond = solve_posterior(
        lambda i, j: block(fac, scale, Ls[i], Ls[j], Xs[i], Xs[j]), noises, ys,
        lambda i: block(fac, scale, ident, Ls[i], Xt, Xs[i]), lambda t: mpmath.mpf(scale), Xt)
    out = {"heat_Xt": Xt, "heat_weights": w, "heat_mean": mean, "heat_var": var, "heat_cond": cond,
           "heat_noise": np.array([float(v) for v in noises])}
    for i, X in enumerate(Xs):
        out[f"heat_X{i}"] = X
        out[f"heat_Y{i}"] = ys[i]
    print(f"heat: N = {sum(len(y) for y in ys)}, cond_2(G) = {cond:.2e}")
    return out


def neumann():
    """Isotropic Matern-5/2, kappa(s) = (1 + s + s^2 / 3) e^{-s}, s = sqrt(5) ||(x - y) / l||: values and -0.8 <n, grad u>."""
    d = 2
    ls = [sp.Rational(9, 10), sp.Rational(7, 10)]
    scale = sp.Rational(9, 4)
    xs = sp.symbols(f"x0:{d}", real=True)
    yv = sp.symbols(f"y0:{d}", real=True)
    s_ = sp.sqrt(sum((sp.sqrt(5) * (a - b) / l) ** 2 for a, b, l in zip(xs, yv, ls)))
    k = scale * (1 + s_ + s_**2 / 3) * sp.exp(-s_)
    rng = np.random.default_rng(20240701)
    Xv = rng.uniform(-1, 1, size=(9, d))
    Xn = np.array([[-1.0, -0.6], [-1.0, 0.5], [1.0, -0.2], [1.0, 0.7], [0.3, 1.0]])
    normals = np.array([[-1.0, 0.0], [-1.0, 0.0], [1.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
    kappa = sp.Rational(4, 5)

    def dn(expr, vars_, nrm):       # -kappa <n, grad>
        return -kappa * sum(sp.nsimplify(float(c)) * sp.diff(expr, v) for c, v in zip(nrm, vars_))

    def entry_fn(expr):
        f = sp.lambdify((*xs, *yv), expr, "mpmath")
        return lambda a, b: f(*[mpmath.mpf(float(v)) for v in (*a, *b)])

    # the Neumann functional depends on the POINT (its normal): one lambdified expression per distinct normal
    uniq = sorted({tuple(n) for n in normals})
    kvv = entry_fn(k)
    k_nv = {n: entry_fn(dn(k, xs, n)) for n in uniq}                 # (L_n k)(x, y)
    k_vn = {n: entry_fn(dn(k, yv, n)) for n in uniq}                 # (k L_n'*)(x, y)
    k_nn = {(n0, n1): entry_fn(dn(dn(k, xs, n0), yv, n1)) for n0 in uniq for n1 in uniq}

    def mat(rows, cols, f):
        M = mpmath.zeros(len(rows), len(cols))
        for i, r in enumerate(rows):
            for j, c in enumerate(cols):
                M[i, j] = f(i, j, r, c)
        return M

    def gram_block(i, j):
        if (i, j) == (0, 0):
            return mat(Xv, Xv, lambda a, b, r, c: kvv(r, c))
        if (i, j) == (1, 0):
            return mat(Xn, Xv, lambda a, b, r, c: k_nv[tuple(normals[a])](r, c))
        # (coinciding points: the closed form in the distance has a removable singularity there; from kappa(s) = 1 - s^2 / 6 + O(s^4),
        #  d/dx_a d/dy_b k (x, x) = scale * (5 / 3) / l_a^2 * delta_ab)
        def nn(a, b, r, c):
            if a == b:
                return mpmath.mpf(kappa) ** 2 * sum(mpmath.mpf(float(normals[a][q])) ** 2 * mpmath.mpf(scale) * mpmath.mpf(5) / 3 / mpmath.mpf(ls[q]) ** 2
                                                    for q in range(d))
            return k_nn[(tuple(normals[a]), tuple(normals[b]))](r, c)
        return mat(Xn, Xn, nn)

    Xt = np.array([[0.0, 0.0], [0.6, -0.3], [-0.7, 0.8], [0.9, 0.9]])

    def cross(i):
        if i == 0:
            return mat(Xt, Xv, lambda a, b, r, c: kvv(r, c))
        return mat(Xt, Xn, lambda a, b, r, c: k_vn[tuple(normals[b])](r, c))

    yvals = np.cos(2 * Xv[:, 0]) * np.exp(0.5 * Xv[:, 1])
    yn = np.array([0.3, -0.1, 0.25, 0.4, -0.2])
    w, mean, var, cond = solve_posterior(gram_block, ["1e-4", "1e-4"], [yvals, yn], cross, lambda t: mpmath.mpf(scale), Xt)
    print(f"neumann: N = 14, cond_2(G) = {cond:.2e}")
    return {"neumann_Xv": Xv, "neumann_Xn": Xn, "neumann_normals": normals, "neumann_kappa": float(kappa), "neumann_Yv": yvals, "neumann_Yn": yn,
            "neumann_lengthscales": np.array([float(v) for v in ls]), "neumann_scale": float(scale), "neumann_noise": np.array([1e-4, 1e-4]),
            "neumann_Xt": Xt, "neumann_weights": w, "neumann_mean": mean, "neumann_var": var, "neumann_cond": cond}


def main():
    path = os.path.join(HERE, "posterior_multiblock.npz")
    makers = {"poisson2d": poisson2d, "heat": heat, "neumann": neumann, "expquad2d": expquad2d}
    only = sys.argv[1:]                      # `... expquad2d`: recompute that problem only, keep the others' committed vectors
    out = dict(np.load(path)) if only and os.path.exists(path) else {}
    for name, fn in makers.items():
        if not only or name in only:
            out.update(fn())
    np.savez(path, **out)
    print("wrote posterior_multiblock.npz")


if __name__ == "__main__":
    sys.exit(main())
