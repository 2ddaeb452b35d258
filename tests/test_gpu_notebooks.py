"""The reference's canonical call sequences, run unmodified against the MI355X path through
the problem/domain builders:
  * `tests/linpde_gp/problems/test_heat.py:56-99` (heat equation: IC + noisy BCs + PDE)
  * `experiments/0001_poisson_dirichlet_2d.ipynb` cells 6-22 (2-D Poisson-Dirichlet)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_heat_equation_like_reference_test():
    import linpde_gp_amd as lp
    from linpde_gp_amd import domains
    from linpde_gp_amd.problems import pde
    cf = lp.randprocs.covfuncs
    spatial = domains.asdomain([-1.0, 1.0])
    ibvp = pde.HeatEquationDirichletProblem(
        t0=0.0, T=5.0, spatial_domain=spatial, alpha=0.1,
        initial_values=pde.TruncatedSineSeries(spatial, coefficients=[1.0, 2.0]))
    prior = lp.GaussianProcess(
        mean=lp.functions.Zero(input_shape=(2,)),
        cov=1.0**2 * cf.TensorProduct(cf.Matern((), nu=1.5, lengthscales=2.5), cf.Matern((), nu=2.5, lengthscales=2.0)))

    def noise(X):
        n = int(np.prod(X.shape[:-1]))
        return lp.randvars.Normal(np.zeros(X.shape[:-1]), np.diag(1e-5 * np.ones(n)))

    X_ic = ibvp.initial_domain.uniform_grid(5, inset=1e-6)
    Y_ic = ibvp.initial_condition.values(X_ic[..., 1])
    u = prior.condition_on_observations(Y_ic, X_ic)
    assert np.allclose(u.mean(X_ic), Y_ic, rtol=0.0, atol=3e-2)
    for bc in ibvp.boundary_conditions:
        X_bc = bc.boundary.uniform_grid(50)
        Y_bc = bc.values(X_bc)
        u = u.condition_on_observations(Y_bc, X=X_bc, b=noise(X_bc))
        assert np.allclose(u.mean(X_bc), Y_bc, rtol=0.0, atol=3e-2)
    X_pde = ibvp.domain.uniform_grid((100, 20))
    Y_pde = ibvp.pde.rhs(X_pde)
    u = u.condition_on_observations(Y_pde, X=X_pde, L=ibvp.pde.diffop)
    X_test = ibvp.domain.uniform_grid((50, 50))
    Y_test = ibvp.solution(X_test)
    vals = u.mean(X_test)
    std = np.nan_to_num(u.std(X_test))
    assert np.min(vals + 2 * std - Y_test) > -3e-2
    assert np.min(Y_test - (vals - 2 * std)) > -3e-2
    assert np.mean(np.abs(vals - Y_test)) < 3e-2


def test_poisson_2d_notebook_sequence():
    import linpde_gp_amd as lp
    from linpde_gp_amd import domains
    from linpde_gp_amd.problems import pde
    from oracle import gp as ogp
    cf = lp.randprocs.covfuncs
    bvp = pde.PoissonEquationDirichletProblem(
        domain=domains.Box([[-1.0, 1.0], [-1.0, 1.0]]),
        rhs=lp.functions.Constant((2,), 2.0), boundary_values=lp.functions.Constant((2,), 0.0))
    prior = lp.GaussianProcess(
        mean=lp.functions.Zero((2,)),
        cov=2.0**2 * cf.TensorProduct(cf.Matern((), nu=2.5, lengthscales=1.0), cf.Matern((), nu=2.5, lengthscales=1.0)))
    u = prior
    oblocks = []
    N_bc, N_pde = 20, 20                              # the notebook's sizes (cells 12, 19)
    for bc in bvp.boundary_conditions:
        X_bc = bc.boundary.uniform_grid(N_bc, inset=1e-6)
        Y_bc = bc.values(X_bc)
        u = u.condition_on_observations(Y_bc, X=X_bc, b=lp.randvars.Normal(np.zeros(Y_bc.shape), 1e-10))
        oblocks.append(ogp.ObsBlock(X_bc.reshape(-1, 2), {(0, 0): 1.0}, Y_bc.reshape(-1), None, 1e-10))
    X_pde = bvp.domain.uniform_grid((N_pde, N_pde))
    Y_pde = bvp.pde.rhs(X_pde)
    u = u.condition_on_observations(Y=Y_pde, L=bvp.pde.diffop, X=X_pde)
    oblocks.append(ogp.ObsBlock(np.asarray(X_pde).reshape(-1, 2), {(2, 0): -1.0, (0, 2): -1.0}, Y_pde.reshape(-1)))
    plt_grid = bvp.domain.uniform_grid((50, 50))
    mean = u.mean(plt_grid)
    std = u.std(np.asarray(plt_grid).reshape(-1, 2))
    assert mean.shape == (50, 50) and std.shape == (2500,)
    post = ogp.condition([(4.0, [("matern", 2.5, 1.0), ("matern", 2.5, 1.0)])], oblocks)
    Xt = np.asarray(plt_grid).reshape(-1, 2)
    ref_mean, ref_var = post.mean(Xt), post.var(Xt)
    assert np.max(np.abs(mean.reshape(-1) - ref_mean)) / np.max(np.abs(ref_mean)) < 1e-7
    assert np.max(np.abs(std**2 - ref_var)) / np.max(np.abs(ref_var)) < 1e-6
    # posterior over the PDE residual -Lap u - f at the collocation points is (numerically) zero (cell 16)
    Du = bvp.pde.diffop(u)
    res = Du.mean(X_pde) - Y_pde
    assert np.max(np.abs(res)) < 1e-6
