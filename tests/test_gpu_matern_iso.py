"""Isotropic multivariate Matérn with directional derivatives: HIP path vs oracle.

The reference's own cases `cases_matern.py:19-89` for input_shape (3,) (its seeds for the
directions, nu in 1.5 ... 4.5, lengthscale 1) on its own inputs (128 Sobol points in [-3, 3]^3,
seed 109134809 + 3, `test_diffops.py:15-20`), through `CovarianceFunction.matrix`, `__call__`
(x1 = None diagonal), the matrix-free `linop`, and a GP posterior with value + normal-derivative
(Neumann-type) observation blocks.  Tolerances: entries 1e-10 relative (the reference: atol 1e-14
+ rtol 1e-7 against JAX autodiff), posterior mean / variance 1e-8 (north_star).
"""
import numpy as np
import pytest
import scipy.stats

from oracle import covfuncs as ocf
from oracle import gp as ogp

pytestmark = pytest.mark.gpu

D = 3


@pytest.fixture(scope="module")
def lp():
    import linpde_gp_amd
    return linpde_gp_amd


def _dd(v):
    return {tuple(int(i == j) for i in range(len(v))): float(v[j]) for j in range(len(v))}


def _cases():
    out = []
    for nu in (1.5, 2.5, 3.5, 4.5):
        out.append((f"id x dd nu={nu}", nu, None, 2.0 * np.random.default_rng(390852098).standard_normal(size=(D,))))
        out.append((f"dd x id nu={nu}", nu, 2.0 * np.random.default_rng(4158976).standard_normal(size=(D,)), None))
        if nu > 1.5:        # "Not enough differentiability" (cases_matern.py:64-65)
            rng = np.random.default_rng(413598)
            out.append((f"dd x dd nu={nu}", nu, rng.standard_normal(size=(D,)), rng.standard_normal(size=(D,))))
    return out


@pytest.mark.parametrize("name,nu,d0,d1", _cases(), ids=lambda v: v if isinstance(v, str) else None)
def test_reference_matern_cases_3d(lp, name, nu, d0, d1):
    from linpde_gp_amd.linfuncops import diffops
    cf = lp.randprocs.covfuncs
    X = scipy.stats.qmc.scale(scipy.stats.qmc.Sobol(D, seed=109134809 + D).random_base2(7), -3.0, 3.0)
    k = cf.Matern((D,), nu=nu)
    kk = k
    if d1 is not None:
        kk = diffops.DirectionalDerivative(d1)(kk, argnum=1)
    if d0 is not None:
        kk = diffops.DirectionalDerivative(d0)(kk, argnum=0)
    okern = [(1.0, [("matern_iso", nu, np.ones(D))])]
    L0 = ocf.identity(D) if d0 is None else _dd(d0)
    L1 = ocf.identity(D) if d1 is None else _dd(d1)
    ref = ocf.LkL(okern, L0, L1, X, X)
    got = kk.matrix(X, X)
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-12 * np.abs(ref).max())
    # x1 = None: the diagonal shortcut (`_matern.py:65-69,186-191`)
    np.testing.assert_allclose(kk(X, None), ocf.k_diag(okern, L0, L1, X), rtol=1e-13, atol=1e-15)
    # broadcasting call == dense block (test_diffops.py compares both entry points)
    np.testing.assert_allclose(kk(X[:, None, :], X[None, :5, :]), ref[:, :5], rtol=1e-10, atol=1e-12 * np.abs(ref).max())
    # matrix-free product (the `_keops_lazy_tensor` slot, `_matern.py:112-135,231-264`)
    V = np.random.default_rng(24).standard_normal(size=(X.shape[0], 3))
    np.testing.assert_allclose(kk.linop(X, X) @ V, ref @ V, rtol=1e-10, atol=1e-11 * np.abs(ref @ V).max())


def test_iso_lengthscales_sum_and_ragged(lp):
    """Per-dimension lengthscales, a scaled sum with a product kernel, ragged block sizes, d = 2."""
    from linpde_gp_amd.linfuncops import diffops
    cf = lp.randprocs.covfuncs
    rng = np.random.default_rng(7)
    X0, X1 = rng.uniform(-1, 1, size=(150, 2)), rng.uniform(-1, 1, size=(77, 2))
    ls = np.array([0.6, 1.4])
    v0, v1 = rng.standard_normal(2), rng.standard_normal(2)
    k = 1.7 * cf.Matern((2,), nu=2.5, lengthscales=ls) + 0.3 * cf.ExpQuad((2,), lengthscales=0.8)
    kk = diffops.DirectionalDerivative(v0)(diffops.DirectionalDerivative(v1)(k, argnum=1), argnum=0)
    okern = [(1.7, [("matern_iso", 2.5, ls)]), (0.3, [("expquad", 0.8), ("expquad", 0.8)])]
    ref = ocf.LkL(okern, _dd(v0), _dd(v1), X0, X1)
    np.testing.assert_allclose(kk.matrix(X0, X1), ref, rtol=1e-10, atol=1e-12 * np.abs(ref).max())
    # d = 1 with input_shape (1,) is the univariate kernel
    x = rng.uniform(-1, 1, size=(40, 1))
    k1 = diffops.DirectionalDerivative(np.array([1.3]))(cf.Matern((1,), nu=1.5, lengthscales=0.7), argnum=1)
    ref1 = ocf.LkL([(1.0, [("matern", 1.5, 0.7)])], ocf.identity(1), {(1,): 1.3}, x, x)
    np.testing.assert_allclose(k1.matrix(x, x), ref1, rtol=1e-10, atol=1e-13)


def test_iso_errors(lp):
    from linpde_gp_amd.linfuncops import diffops
    cf = lp.randprocs.covfuncs
    k = cf.Matern((2,), nu=2.5)
    X = np.zeros((4, 2))
    with pytest.raises(NotImplementedError):          # the reference would fall back to JAX autodiff
        diffops.Laplacian((2,))(k, argnum=0).matrix(X, X)
    k15 = cf.Matern((2,), nu=1.5)
    dd = diffops.DirectionalDerivative(np.array([1.0, 0.5]))
    with pytest.raises(ValueError):                   # "Not enough differentiability"
        dd(dd(k15, argnum=1), argnum=0).matrix(X, X)
    with pytest.raises(ValueError):
        cf.Matern((3,), nu=2.5, lengthscales=np.ones(2))


def test_iso_posterior_with_neumann_blocks(lp):
    """GP regression on the unit square: noisy values inside, outward normal derivatives on two
    edges (Neumann-type `DirectionalDerivative` blocks), iterative conditioning;
    posterior mean / variance vs the oracle's dense conditioning."""
    from linpde_gp_amd.linfuncops import diffops
    cf = lp.randprocs.covfuncs
    rng = np.random.default_rng(11)
    Xv = rng.uniform(0, 1, size=(300, 2))
    f = lambda X: np.sin(2.0 * X[:, 0]) * np.cos(1.5 * X[:, 1])
    e = np.linspace(0.02, 0.98, 90)
    Xl = np.column_stack([np.zeros_like(e), e])          # x = 0, outward normal (-1, 0)
    Xt = np.column_stack([e, np.ones_like(e)])           # y = 1, outward normal (0, 1)
    dfl = -(2.0 * np.cos(2.0 * Xl[:, 0]) * np.cos(1.5 * Xl[:, 1]))
    dft = -1.5 * np.sin(2.0 * Xt[:, 0]) * np.sin(1.5 * Xt[:, 1])
    ls = np.array([0.7, 0.9])
    prior = lp.GaussianProcess(lp.functions.Zero((2,)), 1.5 * cf.Matern((2,), nu=3.5, lengthscales=ls))
    okern = [(1.5, [("matern_iso", 3.5, ls)])]
    nl, nt = np.array([-1.0, 0.0]), np.array([0.0, 1.0])
    blocks = [ogp.ObsBlock(Xv, ocf.identity(2), f(Xv), 0.0, 1e-6),
              ogp.ObsBlock(Xl, _dd(nl), dfl, 0.0, 1e-6),
              ogp.ObsBlock(Xt, _dd(nt), dft, 0.0, 1e-6)]
    u = prior.condition_on_observations(f(Xv), X=Xv, b=lp.randvars.Normal(np.zeros(300), 1e-6 * np.eye(300)))
    u = u.condition_on_observations(dfl, X=Xl, L=diffops.DirectionalDerivative(nl),
                                    b=lp.randvars.Normal(np.zeros(90), 1e-6 * np.eye(90)))
    u = u.condition_on_observations(dft, X=Xt, L=diffops.DirectionalDerivative(nt),
                                    b=lp.randvars.Normal(np.zeros(90), 1e-6 * np.eye(90)))
    post = ogp.condition(okern, blocks)
    g = np.linspace(0.05, 0.95, 12)
    Xp = np.stack(np.meshgrid(g, g, indexing="ij"), axis=-1).reshape(-1, 2)
    mean, var = u.predict(Xp)
    assert np.max(np.abs(mean - post.mean(Xp))) / np.max(np.abs(post.mean(Xp))) < 1e-8
    assert np.max(np.abs(var - post.var(Xp))) / np.max(np.abs(post.var(Xp))) < 1e-8
    assert np.max(np.abs(mean - f(Xp))) < 5e-2           # and it does regress the function


def test_robin_and_dirac_functionals(lp):
    """Conditioning through functional arithmetic: a Robin condition `2 u - 0.5 du/dy` on one edge as
    `2 * Ev + 0.5 * (Ev @ DirectionalDerivative)` (SumLinearFunctional over one point set), values
    through a `DiracFunctional`, a scaled functional -- against the oracle's dense conditioning with the
    same coefficient maps; isotropic Matérn and tensor-product priors."""
    from linpde_gp_amd.linfuncops import diffops
    cf, lf = lp.randprocs.covfuncs, lp.linfunctls
    rng = np.random.default_rng(3)
    Xv = rng.uniform(0, 1, size=(140, 2))
    e = np.linspace(0.03, 0.97, 70)
    Xe = np.column_stack([e, np.zeros_like(e)])
    f = lambda X: np.cos(1.3 * X[:, 0]) * np.exp(-0.7 * X[:, 1])
    dfdy = lambda X: -0.7 * f(X)
    yr = 2.0 * f(Xe) + 0.5 * (-dfdy(Xe))                 # outward normal (0, -1)
    robin_coeffs = {(0, 0): 2.0, (0, 1): -0.5}
    priors = [
        (1.2 * cf.Matern((2,), nu=2.5, lengthscales=0.8), [(1.2, [("matern_iso", 2.5, np.full(2, 0.8))])]),
        (1.2 * cf.TensorProduct(cf.Matern((), nu=2.5, lengthscales=0.8), cf.Matern((), nu=3.5, lengthscales=1.1)),
         [(1.2, [("matern", 2.5, 0.8), ("matern", 3.5, 1.1)])]),
    ]
    g = np.linspace(0.1, 0.9, 9)
    Xp = np.stack(np.meshgrid(g, g, indexing="ij"), axis=-1).reshape(-1, 2)
    for k, okern in priors:
        prior = lp.GaussianProcess(lp.functions.Zero((2,)), k)
        ev = lf._EvaluationFunctional((2,), (), Xe)
        robin = 2.0 * ev + 0.5 * (ev @ diffops.DirectionalDerivative(np.array([0.0, -1.0])))
        u = prior.condition_on_observations(-3.0 * f(Xv), L=-3.0 * lf.DiracFunctional((2,), (), Xv),
                                            b=lp.randvars.Normal(np.zeros(140), 1e-6 * np.eye(140)))
        u = u.condition_on_observations(yr, L=robin, b=lp.randvars.Normal(np.zeros(70), 1e-6 * np.eye(70)))
        blocks = [ogp.ObsBlock(Xv, {(0, 0): -3.0}, -3.0 * f(Xv), 0.0, 1e-6), ogp.ObsBlock(Xe, robin_coeffs, yr, 0.0, 1e-6)]
        post = ogp.condition(okern, blocks)
        mean, var = u.predict(Xp)
        assert np.max(np.abs(mean - post.mean(Xp))) / np.max(np.abs(post.mean(Xp))) < 1e-8
        assert np.max(np.abs(var - post.var(Xp))) / np.max(np.abs(post.var(Xp))) < 1e-8
        assert np.max(np.abs(mean - f(Xp))) < 5e-2
