"""Matrix-free posteriors (`randprocs/_matrix_free.py`): no Gram matrix, products by
`lpgp_kernel_matvec`, solves by preconditioned conjugate gradients -- against the oracle at small sizes (mixed differential /
value blocks, re-conditioning) and against the package's own dense path at N = 32 768."""
import numpy as np
import pytest

from oracle import covfuncs as ocf
from oracle import gp as ogp

pytestmark = pytest.mark.gpu


@pytest.fixture
def lp():
    import linpde_gp_amd
    saved = {k: getattr(linpde_gp_amd.config, k) for k in ("matrix_free", "matrix_free_above", "matrix_free_preconditioner_rank", "matrix_free_rtol")}
    yield linpde_gp_amd
    for k, v in saved.items():
        setattr(linpde_gp_amd.config, k, v)


def test_matrix_free_posterior_vs_oracle_with_reconditioning(lp):
    cf = lp.randprocs.covfuncs
    from linpde_gp_amd.linfuncops import diffops
    lp.config.matrix_free = True
    rng = np.random.default_rng(4)
    prior = lp.GaussianProcess(lp.functions.Constant((2,), 0.7),
                               1.5**2 * cf.TensorProduct(cf.Matern((), nu=2.5, lengthscales=0.9), cf.Matern((), nu=2.5, lengthscales=1.1)))
    okern = [(2.25, [("matern", 2.5, 0.9), ("matern", 2.5, 1.1)])]
    X1, Y1 = rng.uniform(-1, 1, (400, 2)), rng.normal(size=400)
    X2, Y2 = rng.uniform(-1, 1, (150, 2)), rng.normal(size=150)
    lap = {(2, 0): -1.0, (0, 2): -1.0}
    u1 = prior.condition_on_observations(Y1, X1, b=lp.randvars.Normal(np.zeros(400), 1e-2 * np.eye(400)))
    from linpde_gp_amd.randprocs._matrix_free import MatrixFreeConditionalGaussianProcess
    assert isinstance(u1, MatrixFreeConditionalGaussianProcess)
    assert u1.representer_weights.shape == (400,)          # (solved now: the next conditioning warm-starts from them)
    u2 = u1.condition_on_observations(Y2, X2, L=-1.0 * diffops.Laplacian((2,)), b=lp.randvars.Normal(np.zeros(150), 1e-1 * np.eye(150)))
    Xt = rng.uniform(-1, 1, (37, 2))
    b1 = ogp.ObsBlock(X1, ocf.identity(2), Y1, 0.0, 1e-2)
    b2 = ogp.ObsBlock(X2, lap, Y2, 0.0, 1e-1)
    for u, blocks in ((u1, [b1]), (u2, [b1, b2])):
        post = ogp.condition(okern, blocks, mean_const=0.7)
        m, v = u.predict(Xt)
        assert np.max(np.abs(m - post.mean(Xt))) <= 1e-7 * np.max(np.abs(post.mean(Xt)))
        assert np.max(np.abs(v - post.var(Xt))) <= 1e-7 * np.max(np.abs(post.var(Xt)))
        np.testing.assert_allclose(u.representer_weights, post.weights, rtol=0, atol=1e-6 * np.max(np.abs(post.weights)))
        np.testing.assert_allclose(u.cov.matrix(Xt[:6]), post.cov(Xt[:6]), rtol=0, atol=1e-7 * np.max(np.abs(post.var(Xt))) + 1e-9)
        G = u.gram
        V = rng.standard_normal((G.shape[0], 3))
        np.testing.assert_allclose(G @ V, post.G @ V, rtol=0, atol=1e-10 * np.abs(post.G).max() * G.shape[0])
        np.testing.assert_allclose(G.solve(post.G @ V), V, rtol=0, atol=1e-6)
        assert u.last_solve_info["converged"]
    with pytest.raises(NotImplementedError):
        u2.gram.todense()
    # the warm start from the previous weights is used (fewer iterations than from zero is not guaranteed; the result is)
    assert u2._warm is not None and u2._warm.shape == (550,)


def test_matrix_free_not_positive_definite_is_reported(lp):
    cf = lp.randprocs.covfuncs
    lp.config.matrix_free = True
    prior = lp.GaussianProcess(lp.functions.Zero((1,)), cf.ExpQuad((1,), lengthscales=1.0))
    X = np.array([[0.0], [0.0], [0.5]])
    u = prior.condition_on_observations(np.array([1.0, -1.0, 0.3]), X, b=lp.randvars.Normal(np.zeros(3), -1e-3 * np.eye(3)))
    with pytest.raises(np.linalg.LinAlgError):
        u.representer_weights


@pytest.mark.slow
def test_matrix_free_matches_the_dense_path_at_32k(lp):
    """`gram.solve` by preconditioned CG on the matrix-free product against the dense factorisation, at
    N = 32 768 scattered noisy observations (an 8.6-GB Gram matrix the matrix-free path never forms)."""
    from linpde_gp_amd import problems
    wl = problems.scattered_2d(n=32768, m=48, noise_var=1e-2, seed=3)
    o = wl.observations[0]
    prior = problems.build_prior(wl)
    b = lp.randvars.Normal(np.zeros(o.X.shape[0]), np.full(o.X.shape[0], o.noise_var))
    lp.config.gram_capacity_hint = wl.n_total
    try:
        dense = prior.condition_on_observations(o.Y, o.X, b=b)
        md, vd = dense.predict(wl.Xtest)
        wd = dense.representer_weights
    finally:
        lp.config.gram_capacity_hint = 0
    del dense
    lp.config.matrix_free_above = 20000            # > 20 000 observations: no dense matrix
    lp.config.matrix_free_rtol = 1e-11
    free = prior.condition_on_observations(o.Y, o.X, b=b)
    from linpde_gp_amd.randprocs._matrix_free import MatrixFreeConditionalGaussianProcess
    assert isinstance(free, MatrixFreeConditionalGaussianProcess)
    wf = free.representer_weights
    info = free.last_solve_info
    mf_, vf = free.predict(wl.Xtest)
    print(f"N = {wl.n_total}: CG {info['iterations']} iterations for the weights, rank-{free._precond.rank} pivoted-Cholesky preconditioner "
          f"(delta {free._precond.delta:.2e}); {free._G.products / 1e9:.1f} G kernel entries evaluated in all")
    assert np.max(np.abs(wf - wd)) <= 1e-6 * np.max(np.abs(wd))
    assert np.max(np.abs(mf_ - md)) <= 1e-8 * np.max(np.abs(md))
    assert np.max(np.abs(vf - vd)) <= 1e-7 * np.max(np.abs(vd))
    r = free.gram @ wf - (o.Y)
    assert np.linalg.norm(r) <= 1e-9 * np.linalg.norm(o.Y)


def test_device_resident_iteration_equals_the_host_loop(lp):
    """Round 6: iterates, residuals and search directions resident in HBM, one iteration = launches only (`lpgp_pcg_step`,
    `lpgp_kernel_matvec_dev`) -- against the host loop of round 5 on the same problem: same iteration count (+-1: the dots are
    summed in another order), same weights / mean / variance; two observation blocks (a differential one, offsets into the
    resident vectors), a warm-started re-conditioning, and a multi-column solve (`gram.solve`)."""
    from linpde_gp_amd.linfuncops import diffops
    from linpde_gp_amd.randprocs import _matrix_free as mfree
    cf = lp.randprocs.covfuncs
    rng = np.random.default_rng(12)
    n0, n1 = 1500, 700
    X0, X1 = rng.uniform(-1, 1, (n0, 2)), rng.uniform(-1, 1, (n1, 2))
    prior = lp.GaussianProcess(lp.functions.Zero((2,)), 1.2**2 * cf.TensorProduct(cf.Matern((), nu=2.5, lengthscales=0.5), cf.Matern((), nu=2.5, lengthscales=0.6)))
    Y0 = np.sin(2 * X0[:, 0]) * np.cos(X0[:, 1]) + 0.05 * rng.standard_normal(n0)
    Y1 = rng.standard_normal(n1)
    Xt = rng.uniform(-1, 1, (40, 2))
    saved = (lp.config.matrix_free, lp.config.matrix_free_device_iteration, lp.config.matrix_free_rtol)
    lp.config.matrix_free, lp.config.matrix_free_rtol = True, 1e-11
    out = {}
    try:
        for dev in (False, True):
            lp.config.matrix_free_device_iteration = dev
            u = prior.condition_on_observations(Y0, X0, b=lp.randvars.Normal(np.zeros(n0), np.full(n0, 1e-2)))      # (a DIAGONAL noise: dense noise blocks keep the host loop)
            w0 = np.array(u.representer_weights)
            it0 = u.last_solve_info["iterations"]
            assert bool(u.last_solve_info.get("device_resident", False)) is dev
            u2 = u.condition_on_observations(Y1, X1, L=-1.0 * diffops.Laplacian((2,)), b=lp.randvars.Normal(np.zeros(n1), np.full(n1, 0.5)))
            w = np.array(u2.representer_weights)
            it1 = u2.last_solve_info["iterations"]
            mean, var = u2.predict(Xt)
            S = u2.gram.solve(np.stack([np.concatenate([Y0, Y1]), np.ones(n0 + n1)], axis=1))
            out[dev] = (w0, it0, w, it1, mean, var, S)
    finally:
        lp.config.matrix_free, lp.config.matrix_free_device_iteration, lp.config.matrix_free_rtol = saved
    (w0h, it0h, wh, it1h, mh, vh, Sh), (w0d, it0d, wd, it1d, md, vd, Sd) = out[False], out[True]
    assert abs(it0d - it0h) <= 1 and abs(it1d - it1h) <= 1, (it0h, it0d, it1h, it1d)
    for a, b_ in ((w0d, w0h), (wd, wh), (md, mh), (Sd, Sh)):
        assert np.max(np.abs(a - b_)) <= 1e-8 * np.max(np.abs(b_))
    assert np.max(np.abs(vd - vh)) <= 1e-7 * np.max(np.abs(vh))


def test_operators_apply_to_a_matrix_free_posterior(lp):
    """`LinearFunctionOperator.__call__(u)` / `LinearFunctional.__call__(u)` dispatch on the posterior's type and knew the
    dense posterior only; a matrix-free posterior is read out through the same operator maps (`_conditional.py:432-467`).  The
    Laplacian of the posterior and a Dirac functional, against the dense path of the same problem."""
    from linpde_gp_amd.linfuncops import diffops
    cf = lp.randprocs.covfuncs
    rng = np.random.default_rng(4)
    n = 400
    X = rng.uniform(-1, 1, (n, 2))
    Y = np.sin(2 * X[:, 0]) * np.cos(X[:, 1])
    prior = lp.GaussianProcess(lp.functions.Zero((2,)), cf.TensorProduct(cf.Matern((), nu=3.5, lengthscales=0.6), cf.Matern((), nu=3.5, lengthscales=0.7)))
    b = lp.randvars.Normal(np.zeros(n), np.full(n, 1e-3))
    Xt = rng.uniform(-0.9, 0.9, (12, 2))
    dense = prior.condition_on_observations(Y, X, b=b)
    Ld = (-1.0 * diffops.Laplacian((2,)))(dense)
    md, vd = Ld.predict(Xt)
    saved = (lp.config.matrix_free, lp.config.matrix_free_rtol)
    lp.config.matrix_free, lp.config.matrix_free_rtol = True, 1e-12
    try:
        free = prior.condition_on_observations(Y, X, b=b)
        Lf = (-1.0 * diffops.Laplacian((2,)))(free)
        mf_, vf = Lf.predict(Xt)
        rv = Lf(Xt)
        with pytest.raises(NotImplementedError):
            Lf.condition_on_observations(Y[:3], X[:3])
    finally:
        lp.config.matrix_free, lp.config.matrix_free_rtol = saved
    assert type(Lf).__name__ == "MatrixFreeConditionalGaussianProcess"
    assert np.max(np.abs(mf_ - md)) <= 1e-7 * np.max(np.abs(md))
    assert np.max(np.abs(vf - vd)) <= 1e-7 * np.max(np.abs(vd))
    np.testing.assert_allclose(np.diag(rv.cov), vf, rtol=0, atol=1e-7 * np.max(np.abs(vd)))
    np.testing.assert_allclose(rv.cov, Ld.cov.matrix(Xt), rtol=0, atol=1e-6 * np.max(np.abs(vd)))


def test_a_dense_chain_continues_matrix_free_past_the_threshold(lp):
    """`matrix_free_above` is consulted at EVERY conditioning: a dense posterior conditioned past it continues without a
    Gram matrix, on the same observation blocks, and agrees with the all-dense chain."""
    from linpde_gp_amd.randprocs._matrix_free import MatrixFreeConditionalGaussianProcess
    cf = lp.randprocs.covfuncs
    rng = np.random.default_rng(9)
    X0, X1 = rng.uniform(-1, 1, (300, 2)), rng.uniform(-1, 1, (250, 2))
    Y0, Y1 = np.sin(3 * X0[:, 0]) + X0[:, 1], np.sin(3 * X1[:, 0]) + X1[:, 1]
    prior = lp.GaussianProcess(lp.functions.Zero((2,)), cf.TensorProduct(cf.Matern((), nu=2.5, lengthscales=0.5), cf.Matern((), nu=2.5, lengthscales=0.5)))
    b0, b1 = lp.randvars.Normal(np.zeros(300), np.full(300, 1e-2)), lp.randvars.Normal(np.zeros(250), np.full(250, 1e-2))
    Xt = rng.uniform(-1, 1, (20, 2))
    dense = prior.condition_on_observations(Y0, X0, b=b0).condition_on_observations(Y1, X1, b=b1)
    md, vd = dense.predict(Xt)
    saved = (lp.config.matrix_free_above, lp.config.matrix_free_rtol)
    lp.config.matrix_free_above, lp.config.matrix_free_rtol = 400, 1e-12
    try:
        u0 = prior.condition_on_observations(Y0, X0, b=b0)
        assert type(u0).__name__ == "ConditionalGaussianProcess"
        _ = u0.representer_weights
        u1 = u0.condition_on_observations(Y1, X1, b=b1)
        assert isinstance(u1, MatrixFreeConditionalGaussianProcess)
        mf_, vf = u1.predict(Xt)
    finally:
        lp.config.matrix_free_above, lp.config.matrix_free_rtol = saved
    assert np.max(np.abs(mf_ - md)) <= 1e-8 * np.max(np.abs(md))
    assert np.max(np.abs(vf - vd)) <= 1e-7 * np.max(np.abs(vd))
