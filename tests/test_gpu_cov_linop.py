"""Round 6: the posterior covariance AS A LINEAR OPERATOR (`u.cov.linop(x0, x1)`, `_conditional.py:245-251`: `k_xx - kLas_x0 @
gram.solve(kLas_x1.T)` as probnum linear operators) -- products on the device, the n0 x n1 matrix never formed -- and the dense
host-array product the reference leaves to NumPy around the path (`lpgp_gemm_host`: `randvars/_normal.py:8-71`, `gram.todense()`).
Oracle: `oracle.gp.Posterior.cov` (LAPACK `cho_solve` on the dense Gram matrix)."""
import numpy as np
import pytest

from oracle import covfuncs as ocf
from oracle import gp as ogp

pytestmark = pytest.mark.gpu


@pytest.fixture
def lp():
    import linpde_gp_amd as lp
    return lp


def _chain_1d(lp):
    """The reference's integration case (`test_posterior_gp.py:152-178`): four batches, two with noise, 4 * ExpQuad(l = 0.25)."""
    cf = lp.randprocs.covfuncs
    prior = lp.GaussianProcess(lp.functions.Zero((1,)), 2.0**2 * cf.ExpQuad((1,), lengthscales=0.25))
    okern = [(4.0, [("expquad", 0.25)])]
    sizes = (2, 3, 2, 4)
    Xs = np.linspace(-1.0, 1.0, sum(sizes))[:, None]
    Ys = 2.0 * np.sin(np.pi * Xs[:, 0])
    noise = [(np.ones(2), 0.6**2), None, None, (np.zeros(4), 0.3**2)]
    u, oblocks = prior, []
    for X, Y, nz in zip(np.array_split(Xs, np.cumsum(sizes)[:-1]), np.array_split(Ys, np.cumsum(sizes)[:-1]), noise):
        u = u.condition_on_observations(Y, X, b=None if nz is None else lp.randvars.Normal(nz[0], nz[1] * np.eye(len(Y))))
        oblocks.append(ogp.ObsBlock(X, ocf.identity(1), Y, None if nz is None else nz[0], None if nz is None else nz[1]))
    return u, ogp.condition(okern, oblocks)


def _scattered_2d(lp, n=700):
    cf = lp.randprocs.covfuncs
    rng = np.random.default_rng(8)
    X = rng.uniform(-1, 1, (n, 2))
    Y = np.sin(3 * X[:, 0]) * np.cos(2 * X[:, 1])
    prior = lp.GaussianProcess(lp.functions.Zero((2,)), 1.3**2 * cf.TensorProduct(cf.Matern((), nu=2.5, lengthscales=0.4), cf.Matern((), nu=2.5, lengthscales=0.4)))
    okern = [(1.69, [("matern", 2.5, 0.4), ("matern", 2.5, 0.4)])]
    u = prior.condition_on_observations(Y, X, b=lp.randvars.Normal(np.zeros(n), 1e-3 * np.eye(n)))
    return u, ogp.condition(okern, [ogp.ObsBlock(X, ocf.identity(2), Y, 0.0, 1e-3)])


def _check_operator(op, ref, rng):
    scale = np.max(np.abs(ref))
    n0, n1 = ref.shape
    assert op.shape == (n0, n1) and op.dtype == np.double
    V = rng.standard_normal((n1, 5))
    np.testing.assert_allclose(op @ V, ref @ V, rtol=0, atol=1e-9 * scale * np.sqrt(n1))
    np.testing.assert_allclose(op @ V[:, 0], ref @ V[:, 0], rtol=0, atol=1e-9 * scale * np.sqrt(n1))
    assert (op @ V[:, 0]).shape == (n0,)
    W = rng.standard_normal((n0, 3))
    np.testing.assert_allclose(op.T @ W, ref.T @ W, rtol=0, atol=1e-9 * scale * np.sqrt(n0))
    assert op.T.shape == (n1, n0)
    np.testing.assert_allclose(op.T.T @ V, ref @ V, rtol=0, atol=1e-9 * scale * np.sqrt(n1))
    np.testing.assert_allclose(op.todense(), ref, rtol=0, atol=1e-9 * scale)
    np.testing.assert_allclose(op.T.todense(), ref.T, rtol=0, atol=1e-9 * scale)
    with pytest.raises(ValueError):
        op @ np.zeros(n1 + 1)


def test_posterior_covariance_operator_1d_chain(lp):
    u, post = _chain_1d(lp)
    rng = np.random.default_rng(0)
    X0 = np.linspace(-1.0, 1.0, 50)[:, None]
    X1 = rng.uniform(-1.2, 1.2, (31, 1))
    _check_operator(u.cov.linop(X0, X1), post.cov(X0, X1), rng)
    _check_operator(u.cov.linop(X0), post.cov(X0), rng)
    # the same object through the reference's generic entry point
    assert type(u.cov.linop(X0)).__name__ == "PosteriorCovarianceOperator"


def test_posterior_covariance_operator_2d_wide(lp):
    """700 observations, 900 x 400 points: three tile columns of right-hand sides, ragged everywhere."""
    u, post = _scattered_2d(lp)
    rng = np.random.default_rng(1)
    X0, X1 = rng.uniform(-1, 1, (900, 2)), rng.uniform(-1, 1, (400, 2))
    _check_operator(u.cov.linop(X0, X1), post.cov(X0, X1), rng)


def test_posterior_covariance_operator_through_a_read_out(lp):
    """`L(posterior)` (`_conditional.py:432-450`): the operator of the Laplacian of the posterior."""
    from linpde_gp_amd.linfuncops import diffops
    u, post = _chain_1d(lp)
    rng = np.random.default_rng(2)
    Xt = np.linspace(-0.9, 0.9, 23)[:, None]
    Lu = diffops.Laplacian((1,))(u)
    ref = post.cov(Xt, Ltest={(2,): 1.0})
    op = Lu.cov.linop(Xt)
    scale = np.max(np.abs(ref))
    V = rng.standard_normal((23, 4))
    np.testing.assert_allclose(op @ V, ref @ V, rtol=0, atol=1e-8 * scale * np.sqrt(23))
    np.testing.assert_allclose(op.todense(), ref, rtol=0, atol=1e-8 * scale)


def test_operator_is_a_value(lp):
    """A later conditioning of the posterior extends the shared device matrix; the operator taken before keeps its answer."""
    u, post = _scattered_2d(lp, n=300)
    rng = np.random.default_rng(3)
    X0 = rng.uniform(-1, 1, (40, 2))
    op = u.cov.linop(X0)
    before = op @ np.eye(40)
    Xn = rng.uniform(-1, 1, (150, 2))
    u2 = u.condition_on_observations(np.zeros(150), Xn, b=lp.randvars.Normal(np.zeros(150), 1e-2 * np.eye(150)))
    u2.predict(X0)
    np.testing.assert_allclose(op @ np.eye(40), before, rtol=0, atol=1e-14)
    np.testing.assert_allclose(before, post.cov(X0), rtol=0, atol=1e-9 * np.max(np.abs(post.cov(X0))))
    assert np.max(np.abs(u2.cov.linop(X0).todense() - before)) > 1e-6        # (the new posterior is a different one)


def test_empty_and_prior_cases(lp):
    cf = lp.randprocs.covfuncs
    u, _ = _chain_1d(lp)
    op = u.cov.linop(np.zeros((0, 1)), np.linspace(0, 1, 4)[:, None])
    assert op.shape == (0, 4) and (op @ np.ones(4)).shape == (0,) and op.todense().shape == (0, 4)
    with pytest.raises(ValueError):
        u.cov.linop(np.zeros((2, 3, 1)))


@pytest.mark.parametrize("transa,transb", [(False, False), (True, False), (False, True), (True, True)])
def test_gemm_host(lp, transa, transb):
    """`lpgp_gemm_host`: C-order host arrays of any size through the 128 x 128-tile MFMA kernel (padded on the device)."""
    from linpde_gp_amd import _engine
    ctx = _engine.default_context()
    rng = np.random.default_rng(5)
    for m, n, k in ((37, 211, 130), (1, 1, 1), (128, 256, 16), (300, 129, 257)):
        A = rng.standard_normal((k, m) if transa else (m, k))
        B = rng.standard_normal((n, k) if transb else (k, n))
        C0 = rng.standard_normal((m, n))
        opA, opB = (A.T if transa else A), (B.T if transb else B)
        ref = opA @ opB
        bound = 1e-14 * k * np.max(np.abs(opA)) * np.max(np.abs(opB))
        np.testing.assert_allclose(_engine.gemm(ctx, A, B, transa=transa, transb=transb), ref, rtol=0, atol=bound)
        got = _engine.gemm(ctx, A, B, transa=transa, transb=transb, alpha=-0.5, beta=2.0, C=C0)
        np.testing.assert_allclose(got, -0.5 * ref + 2.0 * C0, rtol=0, atol=bound + 1e-14)
    with pytest.raises(ValueError):
        _engine.gemm(ctx, np.zeros((3, 4)), np.zeros((5, 6)))


def test_gram_todense_on_the_device(lp):
    u, post = _scattered_2d(lp, n=300)
    np.testing.assert_allclose(u.gram.todense(), post.G, rtol=0, atol=1e-12 * np.max(np.abs(post.G)))
