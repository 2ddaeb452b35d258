"""HIP path vs CPU oracle on the same seeded inputs, through the C ABI / host mirror.

Tolerances (fp64):
  * Gram / cross-covariance entries:  rtol 1e-12 relative to max|block| (the reference's own
    kernel tests use atol 1e-14 + rtol 1e-7, test_diffops.py:42)
  * posterior mean / marginal variance: 1e-8 relative to max|.| (north_star)
"""
import numpy as np
import pytest
import scipy.linalg

from oracle import covfuncs as ocf
from oracle import gp as ogp

pytestmark = pytest.mark.gpu
# Gram / kernel entries against the oracle, relative to the largest entry of the block (SURVEY section 8d asks for <= 1e-13 absolute on
# O(1) entries; the bar here is the one tests/test_gpu_random.py has carried since round 4, `ENTRY_RTOL`)
ENTRY_ATOL = float(__import__("os").environ.get("LPGP_TEST_ENTRY_ATOL", "4e-15"))


@pytest.fixture(scope="module")
def lp():
    import linpde_gp_amd
    return linpde_gp_amd


def _rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def _sobol_like(rng, n, d, lo=-3.0, hi=3.0):
    return rng.uniform(lo, hi, size=(n, d))


# ---- the reference's own tensor-product kernel cases (cases_tensor_product.py:10-186), on its own
#      inputs: 128 Sobol points in [-3, 3]^2, seed 109134809 + d (test_diffops.py:15-20) ----------
def _ref_tensor_product_cases():
    def unit(rng):
        v = rng.standard_normal(size=(2,))
        return v / np.sqrt(np.sum(v**2))
    cases = []
    rng = np.random.default_rng(390852098); d = unit(rng)
    cases.append(("ID x DD", 1.5, 1.5, None, ("dd", d)))
    cases.append(("DD x ID", 1.5, 1.5, ("dd", d), None))
    rng = np.random.default_rng(390852098); d0 = unit(rng); d1 = unit(rng)
    cases.append(("DD x DD", 1.5, 1.5, ("dd", d0), ("dd", d1)))
    rng = np.random.default_rng(67835487); w = 2.0 * rng.standard_normal(size=(2,))
    cases.append(("ID x WL", 2.5, 2.5, None, ("wl", w)))
    rng = np.random.default_rng(89012645); w = 2.0 * rng.standard_normal(size=(2,))
    cases.append(("WL x ID", 2.5, 2.5, ("wl", w), None))
    rng = np.random.default_rng(89012645); w0 = 2.0 * rng.standard_normal(size=(2,)); w1 = 2.0 * rng.standard_normal(size=(2,))
    cases.append(("WL x WL", 2.5, 2.5, ("wl", w0), ("wl", w1)))
    rng = np.random.default_rng(390852098); d = unit(rng); w = 2.0 * rng.standard_normal(size=(2,))
    cases.append(("DD x WL", 3.5, 3.5, ("dd", d), ("wl", w)))
    cases.append(("WL x DD", 2.5, 2.5, ("wl", w), ("dd", d)))
    cases.append(("ID x heat", 1.5, 2.5, None, ("heat", 0.1)))
    cases.append(("heat x heat", 1.5, 2.5, ("heat", 0.2), ("heat", 0.1)))
    return cases


@pytest.mark.parametrize("name,nu0,nu1,L0,L1", _ref_tensor_product_cases(), ids=lambda v: v if isinstance(v, str) else None)
def test_reference_tensor_product_cases(lp, name, nu0, nu1, L0, L1):
    import scipy.stats
    from linpde_gp_amd.linfuncops import diffops
    cf = lp.randprocs.covfuncs
    X = scipy.stats.qmc.scale(scipy.stats.qmc.Sobol(2, seed=109134809 + 2).random_base2(7), -3.0, 3.0)

    def host_op(spec):
        if spec is None:
            return None
        kind, par = spec
        return {"dd": lambda: diffops.DirectionalDerivative(par), "wl": lambda: diffops.WeightedLaplacian(par),
                "heat": lambda: diffops.HeatOperator((2,), alpha=par)}[kind]()

    def coeffs(spec):
        if spec is None:
            return ocf.identity(2)
        kind, par = spec
        if kind == "dd":
            return {(1, 0): float(par[0]), (0, 1): float(par[1])}
        if kind == "wl":
            return {(2, 0): float(par[0]), (0, 2): float(par[1])}
        return {(1, 0): 1.0, (0, 2): -float(par)}          # d/dt - alpha d^2/dx^2

    k = cf.TensorProduct(cf.Matern((), nu=nu0), cf.Matern((), nu=nu1))
    kk = k
    if L1 is not None:
        kk = host_op(L1)(kk, argnum=1)
    if L0 is not None:
        kk = host_op(L0)(kk, argnum=0)
    got = kk.matrix(X, X)
    ref = ocf.LkL([(1.0, [("matern", nu0, 1.0), ("matern", nu1, 1.0)])], coeffs(L0), coeffs(L1), X, X)
    # the reference's own bar is atol 1e-14 + rtol 1e-7 against JAX autodiff (test_diffops.py:42)
    np.testing.assert_allclose(got, ref, rtol=0, atol=ENTRY_ATOL * np.abs(ref).max())


# ---- (1) kernel blocks -------------------------------------------------------------------
MATERN_CASES = [(nu, a, b) for nu in (1.5, 2.5, 3.5, 4.5) for a in range(3) for b in range(3)
                if a + b <= 2 * int(nu - 0.5)]


@pytest.mark.parametrize("nu,a,b", MATERN_CASES)
def test_matern_1d_blocks(lp, nu, a, b):
    cf = lp.randprocs.covfuncs
    from linpde_gp_amd.linfuncops import diffops
    rng = np.random.default_rng(390852098 + int(10 * nu) + 3 * a + b)
    X0 = _sobol_like(rng, 150, 1)
    X1 = _sobol_like(rng, 77, 1)
    ell = 0.9
    k = cf.Matern((), nu=nu, lengthscales=ell)
    kk = diffops.Derivative(a)(diffops.Derivative(b)(k, argnum=1), argnum=0)
    got = kk.matrix(X0[:, 0], X1[:, 0])
    ref = ocf.LkL([(1.0, [("matern", nu, ell)])], {(a,): 1.0}, {(b,): 1.0}, X0, X1)
    assert _rel(got, ref) < 1e-12


@pytest.mark.parametrize("a,b", [(0, 0), (1, 0), (0, 1), (1, 1), (2, 0), (0, 2), (2, 2), (1, 2), (2, 1)])
def test_expquad_1d_blocks(lp, a, b):
    cf = lp.randprocs.covfuncs
    from linpde_gp_amd.linfuncops import diffops
    rng = np.random.default_rng(4158 + 3 * a + b)
    X0 = _sobol_like(rng, 130, 1, -1, 1)
    X1 = _sobol_like(rng, 64, 1, -1, 1)
    k = 4.0 * cf.ExpQuad((), lengthscales=0.25)
    kk = diffops.Derivative(a)(diffops.Derivative(b)(k, argnum=1), argnum=0)
    got = kk.matrix(X0[:, 0], X1[:, 0])
    ref = ocf.LkL([(4.0, [("expquad", 0.25)])], {(a,): 1.0}, {(b,): 1.0}, X0, X1)
    assert _rel(got, ref) < 1e-12


def test_poisson2d_blocks(lp):
    cf = lp.randprocs.covfuncs
    from linpde_gp_amd.linfuncops import diffops
    rng = np.random.default_rng(24)
    X0 = _sobol_like(rng, 200, 2, -1, 1)
    X1 = _sobol_like(rng, 131, 2, -1, 1)
    k = 2.0**2 * cf.TensorProduct(cf.Matern((), nu=2.5, lengthscales=1.0), cf.Matern((), nu=2.5, lengthscales=0.7))
    okern = [(4.0, [("matern", 2.5, 1.0), ("matern", 2.5, 0.7)])]
    D = -1.0 * diffops.Laplacian((2,))
    lap = {(2, 0): -1.0, (0, 2): -1.0}
    ident = ocf.identity(2)
    assert _rel(k.matrix(X0, X1), ocf.LkL(okern, ident, ident, X0, X1)) < 1e-13
    assert _rel(D(k, argnum=1).matrix(X0, X1), ocf.LkL(okern, ident, lap, X0, X1)) < 1e-12
    assert _rel(D(k, argnum=0).matrix(X0, X1), ocf.LkL(okern, lap, ident, X0, X1)) < 1e-12
    assert _rel(D(D(k, argnum=1), argnum=0).matrix(X0, X1), ocf.LkL(okern, lap, lap, X0, X1)) < 1e-12
    # symmetric (lower-only) assembly path
    G = D(D(k, argnum=1), argnum=0).matrix(X0)
    assert _rel(G, ocf.LkL(okern, lap, lap, X0, X0)) < 1e-12


def test_heat_blocks(lp):
    # cases_tensor_product.py:167-186 (heat x heat) of the reference
    cf = lp.randprocs.covfuncs
    from linpde_gp_amd.linfuncops import diffops
    rng = np.random.default_rng(128)
    X0 = np.column_stack([rng.uniform(0, 5, 90), rng.uniform(-1, 1, 90)])
    X1 = np.column_stack([rng.uniform(0, 5, 70), rng.uniform(-1, 1, 70)])
    k = cf.TensorProduct(cf.Matern((), nu=1.5, lengthscales=2.5), cf.Matern((), nu=2.5, lengthscales=2.0))
    okern = [(1.0, [("matern", 1.5, 2.5), ("matern", 2.5, 2.0)])]
    H = diffops.HeatOperator((2,), alpha=0.1)
    heat = {(1, 0): 1.0, (0, 2): -0.1}
    ident = ocf.identity(2)
    assert _rel(H(H(k, argnum=1), argnum=0).matrix(X0, X1), ocf.LkL(okern, heat, heat, X0, X1)) < 1e-12
    assert _rel(H(k, argnum=0).matrix(X0, X1), ocf.LkL(okern, heat, ident, X0, X1)) < 1e-12
    dd = diffops.DirectionalDerivative([0.3, -1.2])
    ddc = {(1, 0): 0.3, (0, 1): -1.2}
    assert _rel(dd(H(k, argnum=1), argnum=0).matrix(X0, X1), ocf.LkL(okern, ddc, heat, X0, X1)) < 1e-12


def test_sum_kernel_and_3d(lp):
    cf = lp.randprocs.covfuncs
    from linpde_gp_amd.linfuncops import diffops
    rng = np.random.default_rng(5)
    X0 = rng.normal(size=(70, 3))
    X1 = rng.normal(size=(65, 3))
    k1 = 1.5 * cf.ExpQuad((3,), lengthscales=[0.4, 1.3, 0.9])
    k2 = 0.5 * cf.TensorProduct(*(cf.Matern((), nu=2.5, lengthscales=l) for l in (1.0, 2.0, 0.5)))
    k = k1 + k2
    okern = [(1.5, [("expquad", 0.4), ("expquad", 1.3), ("expquad", 0.9)]),
             (0.5, [("matern", 2.5, 1.0), ("matern", 2.5, 2.0), ("matern", 2.5, 0.5)])]
    L = diffops.Laplacian((3,))
    lap = {(2, 0, 0): 1.0, (0, 2, 0): 1.0, (0, 0, 2): 1.0}
    assert _rel(L(L(k, argnum=1), argnum=0).matrix(X0, X1), ocf.LkL(okern, lap, lap, X0, X1)) < 1e-12


def _general_operator(d, rng, max_order=2):
    """A general operator  sum_{|a| <= max_order} c_a d^a  (mixed derivatives included) and its oracle dictionary."""
    import itertools
    from linpde_gp_amd.linfuncops import diffops
    from linpde_gp_amd.linfuncops.diffops._coefficients import MultiIndex, PartialDerivativeCoefficients
    idx = [a for a in itertools.product(range(max_order + 1), repeat=d) if sum(a) <= max_order]
    coef = {a: float(c) for a, c in zip(idx, rng.uniform(0.5, 1.5, len(idx)) * rng.choice([-1.0, 1.0], len(idx)))}
    pdc = PartialDerivativeCoefficients({(): {MultiIndex(a): c for a, c in coef.items()}}, (d,), ())
    return diffops.LinearDifferentialOperator(pdc, ((d,), ())), coef


def test_long_sums_and_many_terms(lp):
    """The reference's sums and operators are unbounded (`_jax_arithmetic.py:16-66`, `_tensor_product.py:38-67`); the
    descriptor tables hold 16 summands and 256 terms per summand (round 3: 4 and 64): seven summands under a pair of
    Laplacians, and a pair of GENERAL second-order operators in three dimensions (10 x 10 = 100 terms)."""
    cf = lp.randprocs.covfuncs
    from linpde_gp_amd.linfuncops import diffops
    rng = np.random.default_rng(77)
    X0, X1 = rng.uniform(-1, 1, size=(75, 2)), rng.uniform(-1, 1, size=(66, 2))
    k, okern = None, []
    for g in range(7):
        s, l0, l1 = 0.3 + 0.2 * g, 0.6 + 0.15 * g, 1.4 - 0.1 * g
        if g % 2:
            kg = s * cf.TensorProduct(cf.ExpQuad((), lengthscales=l0), cf.Matern((), nu=2.5, lengthscales=l1))
            okern.append((s, [("expquad", l0), ("matern", 2.5, l1)]))
        else:
            kg = s * cf.TensorProduct(cf.Matern((), nu=3.5, lengthscales=l0), cf.Matern((), nu=2.5, lengthscales=l1))
            okern.append((s, [("matern", 3.5, l0), ("matern", 2.5, l1)]))
        k = kg if k is None else k + kg
    L = diffops.Laplacian((2,))
    lap = {(2, 0): 1.0, (0, 2): 1.0}
    assert _rel(L(L(k, argnum=1), argnum=0).matrix(X0, X1), ocf.LkL(okern, lap, lap, X0, X1)) < 1e-12
    G = L(L(k, argnum=1), argnum=0).matrix(X0)
    assert _rel(G, ocf.LkL(okern, lap, lap, X0, X0)) < 1e-12
    # 100 terms
    X0, X1 = rng.uniform(-1, 1, size=(70, 3)), rng.uniform(-1, 1, size=(40, 3))
    k3 = 1.3 * cf.TensorProduct(*(cf.Matern((), nu=2.5, lengthscales=l) for l in (1.0, 1.7, 0.8)))
    okern3 = [(1.3, [("matern", 2.5, 1.0), ("matern", 2.5, 1.7), ("matern", 2.5, 0.8)])]
    A, ca = _general_operator(3, rng)
    B, cb = _general_operator(3, rng)
    assert len(ca) * len(cb) == 100
    assert _rel(A(B(k3, argnum=1), argnum=0).matrix(X0, X1), ocf.LkL(okern3, ca, cb, X0, X1)) < 1e-12


def test_grid_blocks_beyond_the_kronecker_tables(lp, kronecker_everywhere):
    """A sum on tensor grids that does not fit the fixed-size tables of the Kronecker path (`lpgp_kron_fits`: 48 terms,
    16 distinct 1-D matrices per dimension) is assembled entry-wise from the flattened grids -- round 3 raised an error --
    in `matrix`-free conditioning and prediction alike: same Gram matrix as scattered copies of the same points."""
    from linpde_gp_amd import domains
    from linpde_gp_amd._lib import lib, make_kdesc_array
    cf = lp.randprocs.covfuncs
    rng = np.random.default_rng(78)
    k = (1.1 * cf.TensorProduct(cf.Matern((), nu=2.5, lengthscales=1.3), cf.Matern((), nu=3.5, lengthscales=0.9))
         + 0.4 * cf.TensorProduct(cf.ExpQuad((), lengthscales=1.0), cf.Matern((), nu=2.5, lengthscales=1.6))
         + 0.2 * cf.TensorProduct(cf.Matern((), nu=3.5, lengthscales=2.0), cf.ExpQuad((), lengthscales=0.7)))
    A, ca = _general_operator(2, rng)          # 6 x 6 = 36 terms per summand, 108 in all
    prior = lp.GaussianProcess(lp.functions.Zero((2,)), k)
    Xg = domains.TensorProductGrid(np.linspace(-1.0, 1.0, 19), np.linspace(-0.5, 0.7, 23))
    Xv = domains.TensorProductGrid(np.linspace(-0.9, 0.9, 7), np.linspace(-0.4, 0.6, 9))
    Yg, Yv = rng.standard_normal(Xg.shape[:-1]), rng.standard_normal(Xv.shape[:-1])
    Xt = rng.uniform(-0.5, 0.5, size=(30, 2))

    def run(Xv_, Yv_, Xg_, Yg_):
        u = prior.condition_on_observations(Yv_, X=Xv_, b=lp.randvars.Normal(np.zeros(Yv_.shape), np.full(Yv_.size, 1e-4)))
        u = u.condition_on_observations(Yg_, X=Xg_, L=A, b=lp.randvars.Normal(np.zeros(Yg_.shape), np.full(Yg_.size, 1e-2)))
        return (u.gram.todense(), *u.predict(Xt))
    G1, m1, v1 = run(Xv, Yv, Xg, Yg)                                                           # grids
    G0, m0, v0 = run(np.asarray(Xv).reshape(-1, 2), Yv.reshape(-1), np.asarray(Xg).reshape(-1, 2), Yg.reshape(-1))
    np.testing.assert_allclose(G1, G0, rtol=0, atol=1e-13 * np.abs(G0).max())
    assert _rel(m1, m0) < 1e-9 and _rel(v1, v0) < 1e-9
    okern = [(1.1, [("matern", 2.5, 1.3), ("matern", 3.5, 0.9)]), (0.4, [("expquad", 1.0), ("matern", 2.5, 1.6)]),
             (0.2, [("matern", 3.5, 2.0), ("expquad", 0.7)])]
    Xgf = np.asarray(Xg).reshape(-1, 2)
    n0 = Yv.size
    Gpde = ocf.LkL(okern, ca, ca, Xgf, Xgf) + 1e-2 * np.eye(Xgf.shape[0])
    assert _rel(G1[n0:, n0:], Gpde) < 1e-12


# ---- (2) factor + solve --------------------------------------------------------------------
def _poisson_blocks(nb, npde, rhs=2.0, noise=1e-8):
    """Small version of BASELINE config c3."""
    g = np.linspace(-1, 1, npde)
    Xp = np.stack(np.meshgrid(g, g, indexing="ij"), axis=-1).reshape(-1, 2)
    e = np.linspace(-1 + 1e-6, 1 - 1e-6, nb)
    edges = [np.column_stack([np.full(nb, -1.0), e]), np.column_stack([np.full(nb, 1.0), e]),
             np.column_stack([e, np.full(nb, -1.0)]), np.column_stack([e, np.full(nb, 1.0)])]
    ident = ocf.identity(2)
    lap = {(2, 0): -1.0, (0, 2): -1.0}
    blocks = [ogp.ObsBlock(X, ident, np.zeros(nb), 0.0, noise) for X in edges]
    blocks.append(ogp.ObsBlock(Xp, lap, np.full(Xp.shape[0], rhs)))
    return blocks


def _condition_host(lp, prior, blocks):
    from linpde_gp_amd.linfuncops import diffops
    u = prior
    for b in blocks:
        n = b.n
        noise = None if b.noise_cov is None else lp.randvars.Normal(np.zeros(n), b.noise_cov * np.eye(n))
        if set(b.L) == {(0, 0)}:
            u = u.condition_on_observations(b.Y, X=b.X, b=noise)
        else:
            u = u.condition_on_observations(b.Y, X=b.X, L=-1.0 * diffops.Laplacian((2,)), b=noise)
    return u


def test_poisson2d_posterior_small(lp):
    cf = lp.randprocs.covfuncs
    blocks = _poisson_blocks(nb=24, npde=20)       # 4*24 + 400 = 496 observations, ragged vs 128
    okern = [(4.0, [("matern", 2.5, 1.0), ("matern", 2.5, 1.0)])]
    prior = lp.GaussianProcess(
        lp.functions.Zero((2,)),
        2.0**2 * cf.TensorProduct(cf.Matern((), nu=2.5, lengthscales=1.0), cf.Matern((), nu=2.5, lengthscales=1.0)))
    u = _condition_host(lp, prior, blocks)
    post = ogp.condition(okern, blocks)
    t = np.linspace(-1 + 1 / 16, 1 - 1 / 16, 16)
    Xt = np.stack(np.meshgrid(t, t, indexing="ij"), axis=-1).reshape(-1, 2)
    mean, var = u.predict(Xt)
    assert _rel(mean, post.mean(Xt)) < 1e-8
    ref_var = post.var(Xt)
    assert np.max(np.abs(var - ref_var)) / np.max(np.abs(ref_var)) < 1e-8
    # factor against LAPACK: two backward-stable factorisations of a Gram matrix with
    # cond ~ 1e10 (boundary nugget 1e-8) agree to cond*eps in the factor, and exactly
    # (to rounding) in the product L L^T
    Lf = u.gram.cholesky(True)
    assert _rel(Lf, post.chol) < 1e-5
    np.testing.assert_allclose(u.gram.todense(), post.G, rtol=0, atol=1e-12 * np.max(np.abs(post.G)))
    # analytic solution of -Lap u = 2 with zero boundary is not polynomial; sanity only: u > 0 inside
    assert mean.min() > 0.0


def test_mean_without_weights_matches_weight_path(lp):
    """predict(mean + variance) forms the mean as V^T (L^{-1} r) (lpgp_mat_set_residual, no
    representer weights); mean-only prediction uses K_xX w.  Both against the oracle and
    against each other, on a point count that is / is not a multiple of the 128-tile."""
    cf = lp.randprocs.covfuncs
    blocks = _poisson_blocks(nb=20, npde=18)
    okern = [(4.0, [("matern", 2.5, 1.0), ("matern", 2.5, 1.0)])]
    prior = lp.GaussianProcess(
        lp.functions.Zero((2,)),
        2.0**2 * cf.TensorProduct(cf.Matern((), nu=2.5, lengthscales=1.0), cf.Matern((), nu=2.5, lengthscales=1.0)))
    post = ogp.condition(okern, blocks)
    rng = np.random.default_rng(77)
    for m in (128, 201):
        u = _condition_host(lp, prior, blocks)
        Xt = rng.uniform(-0.95, 0.95, size=(m, 2))
        assert u._representer_weights is None
        mean_z, var = u.predict(Xt)                    # residual path
        assert u._representer_weights is None          # ... really did not solve for the weights
        assert u.predict(Xt, return_var=False) is not None and u._representer_weights is None      # (served from the posterior's last prediction)
        u._pred_cache = None
        mean_w = u.predict(Xt, return_var=False)       # weights path
        assert u._representer_weights is not None
        u._pred_cache = None
        mean_w2, var2 = u.predict(Xt)                  # weights now resident: K w, then the solve
        ref = post.mean(Xt)
        assert _rel(mean_z, ref) < 1e-8 and _rel(mean_w, ref) < 1e-8
        assert _rel(mean_z, mean_w) < 1e-8
        np.testing.assert_array_equal(mean_w, mean_w2)
        # two reduction kernels; var = k(x,x) - sum v^2 cancels against k(x,x) = 4
        np.testing.assert_allclose(var, var2, rtol=0, atol=1e-13 * 4.0)
        assert _rel(var, post.var(Xt)) < 1e-8


def test_posterior_is_freed_without_cyclic_gc(lp):
    """A dropped posterior must release its device matrix by reference counting alone (no
    posterior -> mean -> posterior cycle): at c4 the matrix is 35 GB."""
    import gc
    import weakref
    cf = lp.randprocs.covfuncs
    prior = lp.GaussianProcess(lp.functions.Zero((2,)),
                               cf.TensorProduct(cf.Matern((), nu=2.5), cf.Matern((), nu=2.5)))
    blocks = _poisson_blocks(nb=8, npde=6)
    gc.collect()
    gc.disable()
    try:
        u = _condition_host(lp, prior, blocks)
        m = u.mean
        _ = u.predict(np.zeros((3, 2)))
        state = weakref.ref(u._state)
        del u
        assert state() is not None          # the mean function keeps its posterior alive ...
        del m, _
        assert state() is None              # ... and nothing else does
    finally:
        gc.enable()


def test_iterative_equals_oneshot_and_linop_readout(lp):
    """Reference `tests/linpde_gp/randprocs/test_posterior_gp.py:152-178`: 4 batches (2,3,2,4),
    two with Normal noise, prior 4*ExpQuad(l=0.25); also the Laplacian read-out."""
    cf = lp.randprocs.covfuncs
    from linpde_gp_amd.linfuncops import diffops
    prior = lp.GaussianProcess(lp.functions.Zero((1,)), 2.0**2 * cf.ExpQuad((1,), lengthscales=0.25))
    okern = [(4.0, [("expquad", 0.25)])]
    sizes = (2, 3, 2, 4)
    Xs = np.linspace(-1.0, 1.0, sum(sizes))[:, None]
    Ys = 2.0 * np.sin(np.pi * Xs[:, 0])
    Xb = np.array_split(Xs, np.cumsum(sizes)[:-1])
    Yb = np.array_split(Ys, np.cumsum(sizes)[:-1])
    noise = [(np.ones(2), 0.6**2), None, None, (np.zeros(4), 0.3**2)]
    u = prior
    oblocks = []
    for X, Y, nz in zip(Xb, Yb, noise):
        b = None if nz is None else lp.randvars.Normal(nz[0], nz[1] * np.eye(len(Y)))
        u = u.condition_on_observations(Y, X, b=b)
        oblocks.append(ogp.ObsBlock(X, ocf.identity(1), Y, None if nz is None else nz[0],
                                    None if nz is None else nz[1]))
    post = ogp.condition(okern, oblocks)
    post_it = ogp.condition_iteratively(okern, oblocks)
    Xt = np.linspace(-1.0, 1.0, 50)[:, None]
    rv = u(Xt)
    np.testing.assert_allclose(rv.mean, post.mean(Xt), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(rv.cov, post.cov(Xt), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(rv.var, post.var(Xt), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(post_it.mean(Xt), post.mean(Xt), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(u.representer_weights, post.weights, rtol=1e-7, atol=1e-9)
    # Laplacian of the posterior (`test_posterior_gp_linop`)
    Lu = diffops.Laplacian((1,))(u)
    lap = {(2,): 1.0}
    rvL = Lu(Xt)
    np.testing.assert_allclose(rvL.mean, post.mean(Xt, lap), rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(rvL.cov, post.cov(Xt, Ltest=lap), rtol=1e-7, atol=1e-6)


def test_tensor_grid_assembly_matches_generic(lp, kronecker_everywhere):
    """Observations on a `TensorProductGrid` are assembled as sums of Kronecker products of 1-D
    kernel matrices (`lpgp_gram_assemble_grid`; the reference's Kronecker `linop`,
    covfuncs/_tensor_product.py:64-82, diffops/_tensor_product.py:140-156): same Gram matrix
    and posterior as the per-entry path; ragged grids, heat operator x value observations on a
    second grid, sum kernel.  Sizes (37 x 70, 9 x 40) take the register-resident 2-D kernel with
    partial tiles, (21 x 23, 5 x 11) the general expansion kernel."""
    from linpde_gp_amd import config, domains
    from linpde_gp_amd.linfuncops import diffops
    cf = lp.randprocs.covfuncs
    k = (1.5 * cf.TensorProduct(cf.Matern((), nu=1.5, lengthscales=2.5), cf.Matern((), nu=2.5, lengthscales=2.0))
         + 0.3 * cf.TensorProduct(cf.ExpQuad((), lengthscales=1.1), cf.Matern((), nu=3.5, lengthscales=0.9)))
    prior = lp.GaussianProcess(lp.functions.Zero((2,)), k)
    rng = np.random.default_rng(3)
    for (ng, nv) in (((37, 70), (9, 40)), ((21, 23), (5, 11))):
        _grid_case(lp, prior, rng, ng, nv)


def _grid_case(lp, prior, rng, ng, nv):
    from linpde_gp_amd import config, domains
    from linpde_gp_amd.linfuncops import diffops
    Xg = domains.TensorProductGrid(np.linspace(0.0, 5.0, ng[0]), np.linspace(-1.0, 1.0, ng[1]))
    Xv = domains.TensorProductGrid(np.linspace(0.3, 4.7, nv[0]), np.linspace(-0.8, 0.8, nv[1]))
    Yg, Yv = rng.standard_normal(Xg.shape[:-1]), rng.standard_normal(Xv.shape[:-1])
    Xe = rng.uniform(-1, 1, size=(50, 2)) * np.array([2.5, 1.0]) + np.array([2.5, 0.0])
    Ye = rng.standard_normal(50)
    Xt = rng.uniform(0.0, 1.0, size=(40, 2))
    res = {}
    for flag in (True, False):
        config.use_grid_assembly = flag
        try:
            u = prior.condition_on_observations(Yv, X=Xv, b=lp.randvars.Normal(np.zeros(Yv.shape), np.full(Yv.size, 1e-4)))
            u = u.condition_on_observations(Ye, X=Xe, b=lp.randvars.Normal(np.zeros(50), 1e-3))     # not a grid
            u = u.condition_on_observations(Yg, X=Xg, L=diffops.HeatOperator((2,), alpha=0.1),
                                            b=lp.randvars.Normal(np.zeros(Yg.shape), np.full(Yg.size, 1e-6)))
            res[flag] = (u.gram.todense(), *u.predict(Xt))
        finally:
            config.use_grid_assembly = True
    G1, m1, v1 = res[True]
    G0, m0, v0 = res[False]
    np.testing.assert_allclose(G1, G0, rtol=0, atol=1e-12 * np.abs(G0).max())
    # (the two Gram matrices differ by rounding, ~1e-13; the posterior amplifies that by cond(G))
    assert _rel(m1, m0) < 1e-6 and _rel(v1, v0) < 1e-6
    # a sliced grid no longer is the grid of its factors: silently takes the generic path
    Xs = Xg[::2]
    u = prior.condition_on_observations(Yg[::2], X=Xs, L=diffops.HeatOperator((2,), alpha=0.1),
                                        b=lp.randvars.Normal(np.zeros(Yg[::2].shape), np.full(Yg[::2].size, 1e-6)))
    assert np.all(np.isfinite(u.predict(Xt)[0]))


def test_condition_normal_on_observations(lp):
    """Finite-dimensional Gaussian conditioning (`randvars/_normal.py:8-71`) vs its dense NumPy form."""
    rng = np.random.default_rng(31)
    n, m = 37, 11
    B = rng.standard_normal((n, n))
    prior = lp.randvars.Normal(rng.standard_normal(n), B @ B.T + n * np.eye(n))
    A = rng.standard_normal((m, n))
    Cn = rng.standard_normal((m, m))
    noise = lp.randvars.Normal(rng.standard_normal(m), Cn @ Cn.T + 0.1 * np.eye(m))
    y = rng.standard_normal(m)
    for tr, nz in ((A, noise), (A, None), (None, lp.randvars.Normal(np.zeros(n), 0.5 * np.eye(n)))):
        obs = y if tr is not None else rng.standard_normal(n)
        post = prior.condition_on_observations(obs, nz, tr)
        At = np.eye(n) if tr is None else tr
        S = At @ prior.cov @ At.T + (0 if nz is None else nz.cov)
        K = np.linalg.solve(S, At @ prior.cov).T
        ref_mean = prior.mean + K @ (obs - At @ prior.mean - (0 if nz is None else nz.mean))
        ref_cov = prior.cov - K @ At @ prior.cov
        np.testing.assert_allclose(post.mean, ref_mean, rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(post.cov, ref_cov, rtol=1e-9, atol=1e-9)
    # one scalar observation through a row vector (`np.ndim(A) == 1` branch of the reference)
    post = prior.condition_on_observations(np.array(0.3), lp.randvars.Normal(np.zeros(()), np.array(0.01)), A[0])
    s = A[0] @ prior.cov @ A[0] + 0.01
    np.testing.assert_allclose(post.mean, prior.mean + prior.cov @ A[0] * (0.3 - A[0] @ prior.mean) / s, rtol=1e-10)


def test_potrs_multiple_rhs_and_dense_noise(lp):
    cf = lp.randprocs.covfuncs
    rng = np.random.default_rng(0)
    X = np.sort(rng.uniform(-1, 1, 300))[:, None]
    Y = np.sin(3 * X[:, 0])
    A = rng.normal(size=(300, 300))
    noise_cov = 0.05 * (A @ A.T) / 300 + 0.01 * np.eye(300)
    prior = lp.GaussianProcess(lp.functions.Constant((1,), 0.5), 1.3 * cf.Matern((1,), nu=1.5, lengthscales=0.5))
    u = prior.condition_on_observations(Y, X, b=lp.randvars.Normal(np.zeros(300), noise_cov))
    okern = [(1.3, [("matern", 1.5, 0.5)])]
    post = ogp.condition(okern, [ogp.ObsBlock(X, ocf.identity(1), Y, None, noise_cov)], mean_const=0.5)
    B = rng.normal(size=(300, 5))
    np.testing.assert_allclose(u.gram.solve(B), np.linalg.solve(post.G, B), rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(u.gram.solve(B[:, 0]), np.linalg.solve(post.G, B[:, 0]), rtol=1e-8, atol=1e-8)
    Xt = np.linspace(-1, 1, 33)[:, None]
    m, v = u.predict(Xt)
    assert _rel(m, post.mean(Xt)) < 1e-9 and _rel(v, post.var(Xt)) < 1e-9
    # the rest of the `LinearOperator` protocol the reference's callers use on `gram`
    g = u.gram
    assert g.shape == (300, 300) and g.T is g
    np.testing.assert_allclose(g @ B, post.G @ B, rtol=1e-12, atol=1e-12)          # matrix-free re-evaluation + dense noise
    np.testing.assert_allclose(g @ B[:, 0], post.G @ B[:, 0], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(g.inv() @ B, np.linalg.solve(post.G, B), rtol=1e-8, atol=1e-8)
    assert g.inv().inv() is g
    sign, logdet = np.linalg.slogdet(post.G)
    assert sign == 1.0 and abs(g.logabsdet() - logdet) < 1e-9 * abs(logdet)
    assert abs(g.trace() - np.trace(post.G)) < 1e-12 * np.trace(post.G)
    with pytest.raises(ValueError):
        g @ np.zeros(7)


@pytest.fixture(params=["lazy", "eager"])
def factorization_check(lp, request):
    """`lp.config.lazy_factorization`: False (default, the reference's order of events: its constructor evaluates the
    representer weights, `_conditional.py:44,83,280-282`, so `condition_on_observations` raises itself) / True (opt-in: the
    factorisation is enqueued and a Gram matrix that is not positive definite is reported at the first use of the factor)."""
    saved = lp.config.lazy_factorization
    lp.config.lazy_factorization = request.param == "lazy"
    yield request.param
    lp.config.lazy_factorization = saved


def _condition_expecting_failure(lazy, parent, *args, **kwargs):
    """A conditioning whose Gram matrix is not positive definite: eager -> `condition_on_observations` raises; lazy -> it
    returns an object whose first use raises (and keeps raising)."""
    if not lazy:
        with pytest.raises(np.linalg.LinAlgError):
            parent.condition_on_observations(*args, **kwargs)
        return None
    child = parent.condition_on_observations(*args, **kwargs)
    with pytest.raises(np.linalg.LinAlgError):
        child.predict(np.array([[0.1], [0.2]]))
    with pytest.raises(np.linalg.LinAlgError):
        child.representer_weights
    with pytest.raises(np.linalg.LinAlgError):
        child.condition_on_observations(np.zeros(1), np.array([[0.77]]))          # known dead: raises at once
    return child


def test_not_positive_definite_raises(lp, factorization_check):
    cf = lp.randprocs.covfuncs
    prior = lp.GaussianProcess(lp.functions.Zero((1,)), cf.ExpQuad((1,), lengthscales=1.0))
    X = np.array([[0.0], [0.0], [0.5]])             # duplicated point, no noise => singular Gram
    _condition_expecting_failure(factorization_check == "lazy", prior, np.zeros(3), X, b=lp.randvars.Normal(np.zeros(3), -1e-3 * np.eye(3)))


@pytest.fixture
def lazy_mode(lp):
    saved = lp.config.lazy_factorization
    lp.config.lazy_factorization = True
    yield
    lp.config.lazy_factorization = saved


def test_eager_status_is_the_default(lp):
    """the reference reports a Gram matrix that is not positive definite INSIDE `condition_on_observations`
    (`_conditional.py:44,83`); so does the default configuration here (a try/except jitter retry keeps working)."""
    assert lp.config.lazy_factorization is False
    cf = lp.randprocs.covfuncs
    prior = lp.GaussianProcess(lp.functions.Zero((1,)), cf.ExpQuad((1,), lengthscales=1.0))
    X = np.array([[0.0], [0.0], [0.5]])
    u = None
    tried = []
    for noise in (-1e-3, 1e-3):                     # the retry idiom (first attempt: surely not positive definite)
        tried.append(noise)
        try:
            u = prior.condition_on_observations(np.zeros(3), X, b=lp.randvars.Normal(np.zeros(3), noise * np.eye(3)))
            break
        except np.linalg.LinAlgError:
            continue
    assert tried == [-1e-3, 1e-3] and u is not None and np.all(np.isfinite(u.predict(np.array([[0.1]]))[0]))


def test_lazy_failure_in_the_middle_of_a_chain(lp, lazy_mode):
    """Lazy status (opt-in): a block that is not positive definite in the MIDDLE of a chain of conditionings.  Nothing
    raises while the chain is built (the host runs ahead of the device); the first use of any object from the failed
    block on raises `LinAlgError`, the objects before it stay exact, and the chain can be continued from them."""
    cf = lp.randprocs.covfuncs
    okern = [(1.0, [("expquad", 1.0)])]
    ident = ocf.identity(1)
    prior = lp.GaussianProcess(lp.functions.Zero((1,)), cf.ExpQuad((1,), lengthscales=1.0))
    rng = np.random.default_rng(5)
    X1, Y1 = rng.uniform(-1, 1, (140, 1)), rng.normal(size=140)
    noise = lp.randvars.Normal(np.zeros(140), 1e-2 * np.eye(140))
    Xbad = np.array([[0.2], [0.2], [0.5]])
    X3, Y3 = np.array([[0.31], [-0.62]]), np.array([0.1, 0.2])
    u1 = prior.condition_on_observations(Y1, X1, b=noise)
    u2 = u1.condition_on_observations(np.zeros(3), Xbad, b=lp.randvars.Normal(np.zeros(3), -1e-3 * np.eye(3)))     # not PD
    u3 = u2.condition_on_observations(Y3, X3)                                                                      # built on it
    Xt = np.linspace(-1, 1, 7)[:, None]
    with pytest.raises(np.linalg.LinAlgError):
        u3.predict(Xt)
    with pytest.raises(np.linalg.LinAlgError):
        u2.mean(Xt)
    post1 = ogp.condition(okern, [ogp.ObsBlock(X1, ident, Y1, 0.0, 1e-2)])
    m1, v1 = u1.predict(Xt)
    assert _rel(m1, post1.mean(Xt)) < 1e-8 and np.max(np.abs(v1 - post1.var(Xt))) < 1e-9
    u4 = u1.condition_on_observations(Y3, X3)
    post4 = ogp.condition(okern, [ogp.ObsBlock(X1, ident, Y1, 0.0, 1e-2), ogp.ObsBlock(X3, ident, Y3)])
    m4, v4 = u4.predict(Xt)
    assert _rel(m4, post4.mean(Xt)) < 1e-8 and np.max(np.abs(v4 - post4.var(Xt))) < 1e-9
    with pytest.raises(np.linalg.LinAlgError):
        u3.predict(Xt)                               # still dead: its blocks are not the ones in the matrix
    np.testing.assert_allclose(u1.representer_weights, post1.weights, rtol=1e-7, atol=1e-9)


def test_dead_child_conditioned_while_a_sibling_is_pending(lp, lazy_mode):
    """u2 = u1.cond(bad); u1.predict() finds the failure and truncates to [A]; u2b = u1.cond(B') leaves a
    factorisation pending on [A, B']; conditioning the dead u2 now must raise -- not append rows lowered against u2's
    points onto u2b's blocks (same sizes: it would be silently wrong)."""
    cf = lp.randprocs.covfuncs
    okern = [(1.0, [("expquad", 1.0)])]
    ident = ocf.identity(1)
    prior = lp.GaussianProcess(lp.functions.Zero((1,)), cf.ExpQuad((1,), lengthscales=1.0))
    rng = np.random.default_rng(11)
    X1, Y1 = rng.uniform(-1, 1, (40, 1)), rng.normal(size=40)
    noise = lp.randvars.Normal(np.zeros(40), 1e-2 * np.eye(40))
    Xbad = np.array([[0.2], [0.2], [0.5]])
    Xgood, Ygood = np.array([[0.21], [-0.33], [0.52]]), np.array([0.3, -0.1, 0.2])      # same size as the bad block
    Xc, Yc = np.array([[0.77]]), np.array([0.4])
    Xt = np.linspace(-1, 1, 5)[:, None]
    u1 = prior.condition_on_observations(Y1, X1, b=noise)
    u2 = u1.condition_on_observations(np.zeros(3), Xbad, b=lp.randvars.Normal(np.zeros(3), -1e-3 * np.eye(3)))
    u1.predict(Xt)                                   # verifies: the bad block is dropped, u2 is dead
    u2b = u1.condition_on_observations(Ygood, Xgood)                 # pending on [A, B']
    assert u2b._state.pending
    with pytest.raises(np.linalg.LinAlgError):
        u2.condition_on_observations(Yc, Xc)
    u3 = u2b.condition_on_observations(Yc, Xc)
    post = ogp.condition(okern, [ogp.ObsBlock(X1, ident, Y1, 0.0, 1e-2), ogp.ObsBlock(Xgood, ident, Ygood), ogp.ObsBlock(Xc, ident, Yc)])
    m, v = u3.predict(Xt)
    assert _rel(m, post.mean(Xt)) < 1e-8 and np.max(np.abs(v - post.var(Xt))) < 1e-9


def test_earlier_posteriors_stay_usable_and_branching(lp):
    """The reference's posteriors are immutable values (`_conditional.py:253-294`): an earlier one keeps
    working after a later conditioning, and two conditionings of the same object are independent.  Here
    the chain shares ONE device matrix (views on its leading blocks); the second child gets a copy."""
    cf = lp.randprocs.covfuncs
    okern = [(1.0, [("expquad", 0.5)])]
    ident = ocf.identity(1)
    prior = lp.GaussianProcess(lp.functions.Zero((1,)), cf.ExpQuad((1,), lengthscales=0.5))
    X1, Y1 = np.array([[0.0], [1.0]]), np.array([0.3, -0.2])
    X2, Y2 = np.array([[0.3], [0.6]]), np.ones(2)
    X3, Y3 = np.array([[-0.4], [1.4], [0.45]]), np.array([0.5, 0.1, -0.3])
    Xt = np.linspace(-0.5, 1.5, 9)[:, None]
    b1, b2, b3 = (ogp.ObsBlock(X, ident, Y) for X, Y in ((X1, Y1), (X2, Y2), (X3, Y3)))
    u1 = prior.condition_on_observations(Y1, X1)
    u2 = u1.condition_on_observations(Y2, X2)
    p1, p12, p13 = ogp.condition(okern, [b1]), ogp.condition(okern, [b1, b2]), ogp.condition(okern, [b1, b3])

    def same(u, post):
        m, v = u.predict(Xt)
        assert _rel(m, post.mean(Xt)) < 1e-9 and np.max(np.abs(v - post.var(Xt))) < 1e-9
        np.testing.assert_allclose(u.representer_weights, post.weights, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(u.cov.matrix(Xt[:4]), post.cov(Xt[:4]), rtol=0, atol=1e-9)
        np.testing.assert_allclose(u.mean(Xt), post.mean(Xt), rtol=0, atol=1e-9)

    same(u2, p12)
    same(u1, p1)                  # superseded object: served from the leading part of the shared factor
    same(u2, p12)                 # ... and back
    u3 = u1.condition_on_observations(Y3, X3)      # branching: u1 was already extended by u2
    assert u3._state is not u2._state
    same(u3, p13)
    same(u2, p12)
    same(u1, p1)
    assert u1.gram.shape == (2, 2) and u2.gram.shape == (4, 4) and u3.gram.shape == (5, 5)


def test_failed_conditioning_leaves_the_parent_intact(lp, factorization_check):
    """a failed `condition_on_observations` (Gram not positive definite) must not corrupt the
    device state it shares with the object it was called on (the new block is rolled back)."""
    cf = lp.randprocs.covfuncs
    okern = [(1.0, [("expquad", 1.0)])]
    ident = ocf.identity(1)
    prior = lp.GaussianProcess(lp.functions.Zero((1,)), cf.ExpQuad((1,), lengthscales=1.0))
    rng = np.random.default_rng(3)
    X1, Y1 = rng.uniform(-1, 1, (150, 1)), rng.normal(size=150)      # two tile columns: phase A of the append runs
    noise = lp.randvars.Normal(np.zeros(150), 1e-2 * np.eye(150))
    u1 = prior.condition_on_observations(Y1, X1, b=noise)
    post1 = ogp.condition(okern, [ogp.ObsBlock(X1, ident, Y1, 0.0, 1e-2)])
    Xt = np.linspace(-1, 1, 7)[:, None]
    # the parent has predicted (weights and residual are resident and keyed to it) BEFORE the failure (the
    # rollback clears them on the device, so the host must forget them as well)
    m0, v0 = u1.predict(Xt)
    assert _rel(m0, post1.mean(Xt)) < 1e-8
    assert _rel(u1.mean(Xt), post1.mean(Xt)) < 1e-8
    Xbad = np.array([[0.2], [0.2], [0.5]])           # duplicated point and negative noise: not PD
    _condition_expecting_failure(factorization_check == "lazy", u1, np.zeros(3), Xbad, b=lp.randvars.Normal(np.zeros(3), -1e-3 * np.eye(3)))
    with pytest.raises(ValueError):                 # host-side validation error: nothing reached the device
        u1.condition_on_observations(np.zeros(4), Xbad)
    m, v = u1.predict(Xt)
    assert _rel(m, post1.mean(Xt)) < 1e-8 and np.max(np.abs(v - post1.var(Xt))) < 1e-9
    assert _rel(u1.mean(Xt), post1.mean(Xt)) < 1e-8          # the mean-only path (representer weights)
    np.testing.assert_allclose(u1.representer_weights, post1.weights, rtol=1e-7, atol=1e-9)
    # and the parent can still be extended
    X2, Y2 = np.array([[0.31], [-0.62]]), np.array([0.1, 0.2])
    u2 = u1.condition_on_observations(Y2, X2)
    post2 = ogp.condition(okern, [ogp.ObsBlock(X1, ident, Y1, 0.0, 1e-2), ogp.ObsBlock(X2, ident, Y2)])
    m2, v2 = u2.predict(Xt)
    assert _rel(m2, post2.mean(Xt)) < 1e-8 and np.max(np.abs(v2 - post2.var(Xt))) < 1e-9


def test_api_errors(lp):
    cf = lp.randprocs.covfuncs
    from linpde_gp_amd.linfuncops import diffops
    prior = lp.GaussianProcess(lp.functions.Zero((1,)), cf.Matern((1,), nu=2.5))
    with pytest.raises(ValueError):
        prior.condition_on_observations(np.zeros(3))                      # X and L omitted
    with pytest.raises(ValueError):
        prior.condition_on_observations(np.zeros(3), L=diffops.Laplacian((1,)))   # operator without X
    with pytest.raises(ValueError):
        prior.condition_on_observations(np.zeros(4), np.zeros((3, 1)))    # Y shape
    with pytest.raises(ValueError):
        prior.condition_on_observations(np.zeros(3), np.zeros((3, 1)), b=lp.randvars.Normal(np.zeros(2), np.eye(2)))
    L = diffops.Laplacian((1,)).to_linfunctl(np.zeros((3, 1)))
    with pytest.raises(TypeError):
        prior.condition_on_observations(np.zeros(3), np.zeros((3, 1)), L=L)   # functional + X


def test_outer_update_schedule_matches_single_level(lp):
    """The large-matrix schedule of the factorisation (far columns updated once per nb_outer columns,
    pieces (a0) / (a1) / (b); on by default only beyond 24 576 rows) forced onto a small problem:
    same posterior as the single-level schedule and as the oracle, block append included."""
    from linpde_gp_amd import _engine
    cf = lp.randprocs.covfuncs
    ctx = _engine.default_context()
    blocks = _poisson_blocks(nb=40, npde=50)       # 160 + 2500 observations: 22 tile rows
    okern = [(4.0, [("matern", 2.5, 1.0), ("matern", 2.5, 1.0)])]
    prior = lp.GaussianProcess(
        lp.functions.Zero((2,)),
        2.0**2 * cf.TensorProduct(cf.Matern((), nu=2.5, lengthscales=1.0), cf.Matern((), nu=2.5, lengthscales=1.0)))
    t = np.linspace(-1 + 1 / 16, 1 - 1 / 16, 16)
    Xt = np.stack(np.meshgrid(t, t, indexing="ij"), axis=-1).reshape(-1, 2)
    post = ogp.condition(okern, blocks)
    res = {}
    try:
        for name, nbo, mn in [("single", 0, 1 << 20), ("outer1024", 1024, 2), ("outer1536", 1536, 5)]:
            ctx.set_option("nb_outer", nbo)
            ctx.set_option("nb_outer_min_tiles", mn)
            u = _condition_host(lp, prior, blocks)
            res[name] = u.predict(Xt)
    finally:
        ctx.set_option("nb_outer", 2048)
        ctx.set_option("nb_outer_min_tiles", 192)
    for name, (mean, var) in res.items():
        assert _rel(mean, post.mean(Xt)) < 1e-8, name
        assert np.max(np.abs(var - post.var(Xt))) / np.max(np.abs(post.var(Xt))) < 1e-8, name
    assert _rel(res["outer1024"][0], res["single"][0]) < 1e-10


def test_empty_observation_and_prediction_sets(lp):
    """Empty inputs behave like the NumPy reference's: conditioning on zero observations leaves the
    process unchanged (prior, or the posterior it extends), predicting at zero points returns empty
    arrays, and the posterior covariance of empty point sets is an empty matrix."""
    cf = lp.randprocs.covfuncs
    rng = np.random.default_rng(0)
    prior = lp.GaussianProcess(lp.functions.Constant((2,), 0.3),
                               1.7 * cf.TensorProduct(cf.Matern((), nu=2.5), cf.Matern((), nu=2.5)))
    X, Y = rng.uniform(-1, 1, (5, 2)), rng.standard_normal(5)
    Xt = rng.uniform(-1, 1, (7, 2))
    e = prior.condition_on_observations(np.zeros(0), X=np.zeros((0, 2)))
    m, v = e.predict(Xt)
    np.testing.assert_allclose(m, 0.3)
    np.testing.assert_allclose(v, 1.7)
    assert e.representer_weights.shape == (0,)
    np.testing.assert_allclose(e.cov.matrix(Xt), prior.cov.matrix(Xt), rtol=1e-14)
    u = prior.condition_on_observations(Y, X=X)
    m0, v0 = u.predict(Xt)
    u2 = u.condition_on_observations(np.zeros(0), X=np.zeros((0, 2)))
    m2, v2 = u2.predict(Xt)
    assert np.array_equal(m0, m2) and np.array_equal(v0, v2)
    # the empty posterior can still be extended
    u3 = e.condition_on_observations(Y, X=X)
    np.testing.assert_allclose(u3.predict(Xt)[0], m0, rtol=1e-12)
    me, ve = u.predict(np.zeros((0, 2)))
    assert me.shape == (0,) and ve.shape == (0,)
    assert u.mean(np.zeros((0, 2))).shape == (0,)
    assert u.cov.matrix(np.zeros((0, 2))).shape == (0, 0)
    assert u.cov.matrix(Xt, np.zeros((0, 2))).shape == (7, 0)
    assert prior.cov.matrix(np.zeros((0, 2)), X[:3]).shape == (0, 3)


def test_functional_of_crosscovariance_is_a_linear_operator_covariance(lp):
    """`L0(L1(k, argnum=1))` -> `LinearOperatorCovariance` over the device kernel operator
    (`crosscov/linfunctls/_evaluation.py:163-173`, `randvars/_covariance.py:197-224`)."""
    from linpde_gp_amd.linfuncops import diffops
    cf = lp.randprocs.covfuncs
    rng = np.random.default_rng(77)
    X0, X1 = rng.uniform(-1, 1, (5, 7, 2)), rng.uniform(-1, 1, (9, 2))
    k = 4.0 * cf.TensorProduct(cf.Matern((), nu=2.5, lengthscales=1.0), cf.Matern((), nu=2.5, lengthscales=0.7))
    okern = [(4.0, [("matern", 2.5, 1.0), ("matern", 2.5, 0.7)])]
    lap, ident = {(2, 0): 1.0, (0, 2): 1.0}, ocf.identity(2)
    L0 = diffops.Laplacian((2,)).to_linfunctl(X0)
    L1 = lp.linfunctls._EvaluationFunctional((2,), (), X1)
    cov = L0(L1(k, argnum=1))
    assert isinstance(cov, lp.randvars.LinearOperatorCovariance)
    assert cov.shape0 == (5, 7) and cov.shape1 == (9,) and cov.array.shape == (5, 7, 9)
    ref = ocf.LkL(okern, lap, ident, X0.reshape(-1, 2), X1)
    assert _rel(cov.matrix, ref) < 1e-12
    V = rng.normal(size=(9, 3))
    np.testing.assert_allclose(cov.linop @ V, ref @ V, rtol=0, atol=1e-11 * np.abs(ref @ V).max())
    # argnum = 0: the functional applied first is the LEFT variable
    cov_r = L0(L1(k, argnum=0))
    assert cov_r.shape0 == (9,) and cov_r.shape1 == (5, 7)
    assert _rel(cov_r.matrix, ref.T) < 1e-12
    # the same functional twice: symmetric Gram of the observations (what `from_observations` factors)
    G = L0(L0(k, argnum=1))
    assert _rel(G.matrix, ocf.LkL(okern, lap, lap, X0.reshape(-1, 2), X0.reshape(-1, 2))) < 1e-12
    assert isinstance(G + lp.randvars.ArrayCovariance(np.zeros((5, 7, 5, 7)), (5, 7), (5, 7)), lp.randvars.ArrayCovariance)


def test_polynomial_prior_mean_under_operators(lp):
    """A non-constant prior mean: the posterior given (Y, L) under mean m is m + the zero-mean posterior given
    Y - L[m](X) (`_conditional.py:96-110,193-197`); `L[m]` by closed-form derivatives (`functions.Polynomial`), the
    zero-mean side against the oracle.  Also the derivative read-out of the posterior, whose mean needs L[m](x)."""
    cf = lp.randprocs.covfuncs
    from linpde_gp_amd.linfuncops import diffops
    coeffs = [0.3, -1.0, 0.25, 0.5]
    m = lp.functions.Polynomial(coeffs)
    k = 1.7**2 * cf.Matern((), nu=2.5, lengthscales=0.8)
    okern = [(1.7**2, [("matern", 2.5, 0.8)])]
    X, Xb = np.linspace(-1.0, 1.0, 40), np.array([-1.0, 1.0])
    Y, Yb = np.sin(2.0 * X), np.array([0.2, -0.1])
    L = -1.0 * diffops.Laplacian(())
    noise = lp.randvars.Normal(np.zeros(2), 1e-6 * np.eye(2))
    u = lp.GaussianProcess(m, k).condition_on_observations(Yb, Xb, b=noise).condition_on_observations(Y, X, L=L)
    Lm = -np.polyval(np.polyder(coeffs[::-1], 2), X)                      # -m''(X)
    Y0, Yb0 = Y - Lm, Yb - m(Xb)
    u0 = lp.GaussianProcess(lp.functions.Zero(()), k).condition_on_observations(Yb0, Xb, b=noise) \
        .condition_on_observations(Y0, X, L=L)
    post = ogp.condition(okern, [ogp.ObsBlock(Xb[:, None], ocf.identity(1), Yb0, np.zeros(2), 1e-6),
                                 ogp.ObsBlock(X[:, None], {(2,): -1.0}, Y0)])
    xt = np.linspace(-0.95, 0.95, 33)
    mean, var = u.predict(xt)
    assert _rel(mean, post.mean(xt[:, None]) + m(xt)) < 1e-8
    assert _rel(var, post.var(xt[:, None])) < 1e-8
    np.testing.assert_allclose(u.representer_weights, u0.representer_weights, rtol=1e-12, atol=1e-12)
    # first derivative of the posterior: m'(x) + d/dx of the zero-mean posterior mean
    D = diffops.PartialDerivative(diffops.MultiIndex((1,)))
    dmean = D(u).mean(xt)
    ref = np.polyval(np.polyder(coeffs[::-1], 1), xt) + post.mean(xt[:, None], {(1,): 1.0})
    assert _rel(dmean, ref) < 1e-8
    # a mean without derivatives under an operator: the reference's JAX fallback, NotImplementedError here
    g = lp.GaussianProcess(lp.functions.LambdaFunction(np.cos, ()), k)
    with pytest.raises(NotImplementedError):
        g.condition_on_observations(Y, X, L=L)
    assert g.condition_on_observations(Y, X).predict(xt)[0].shape == (33,)        # plain values are fine
