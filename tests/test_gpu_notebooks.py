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


def _cpu_die_setup(lp):
    """Geometry, material and heat sources of the reference's CPU-die experiment, simplified 1-D model
    (`experiments/cpu.py:19-137,150-229`), restated with the package's own builders."""
    from linpde_gp_amd import domains, functions
    from linpde_gp_amd.linfuncops import diffops
    from linpde_gp_amd.problems import pde
    width, height, depth = 16.28, 9.19, 0.37
    dom = domains.Box([[0.0, width], [0.0, height], [0.0, depth]])
    A_sink_1D = width * height + 2 * height * depth
    core_width, core_offset_x, core_distance_x = 2.5, 1.95, 0.35
    centers = core_offset_x + (core_width + core_distance_x) * np.arange(3, dtype=np.double) + core_width / 2.0
    kappa, TDP = 15.6, 95.0
    xs, ys, eps = [0.0], [0.0], core_distance_x / 3
    for c, h in zip(centers, [0.9, 0.75, 1.0]):
        xs += [c - core_width / 2 - eps, c - core_width / 2, c + core_width / 2, c + core_width / 2 + eps]
        ys += [0.0, h, h, 0.0]
    xs += [width]
    ys += [0.0]
    unnorm = functions.PiecewiseLinear.from_points(xs, ys)
    norm = float(np.trapz(ys, xs))                       # (`LebesgueIntegral(domain[0])(heat_dist_unnorm)`: exact for a piecewise-linear function)
    heat_x = (1.0 / norm) * unnorm
    q_src = (TDP / depth / height) * heat_x
    q_sink = functions.Constant((), -TDP / A_sink_1D / depth)
    q_A = np.full(2, -TDP / A_sink_1D)
    eq = pde.PoissonEquation(domain=dom[0], rhs=q_src + q_sink, alpha=kappa)
    bcs = [pde.BoundaryCondition(dom[0].boundary[0], -kappa * diffops.DirectionalDerivative(1.0), q_A[0]),
           pde.BoundaryCondition(dom[0].boundary[1], -kappa * diffops.DirectionalDerivative(-1.0), q_A[1])]
    bvp = pde.BoundaryValueProblem(pde=eq, boundary_conditions=bcs, solution=None)
    return dict(width=width, kappa=kappa, bvp=bvp, centers=centers, q_A=q_A, rhs=eq.rhs)


def _ivp_solution_1d(rhs, kappa, u0, du0, x):
    """u with -kappa u'' = rhs, u(0) = u0, u'(0) = du0 for a piecewise-polynomial rhs, integrated piece by piece in closed form
    (`Solution_PoissonEquation_IVP_1D_RHSPiecewisePolynomial`, `experiments/cpu.py:204-211`): TEST infrastructure."""
    from linpde_gp_amd import functions
    x = np.atleast_1d(np.asarray(x, dtype=np.double))
    out = np.empty_like(x)
    u, du = float(u0), float(du0)
    for lo, hi, piece in zip(rhs.xs[:-1], rhs.xs[1:], rhs.pieces):
        f1 = (-1.0 / kappa) * piece                     # u'' on the piece
        F1 = f1.integrate()
        F2 = F1.integrate()
        c1 = du - float(F1(np.asarray(lo)))
        c0 = u - float(F2(np.asarray(lo))) - c1 * lo
        sel = (x >= lo) & (x <= hi)
        out[sel] = F2(x[sel]) + c1 * x[sel] + c0
        u = float(F2(np.asarray(hi))) + c1 * hi + c0
        du = float(F1(np.asarray(hi))) + c1
    return out


def test_cpu_die_stationary_1d_notebook_sequence():
    """`experiments/0000_cpu_stationary_1d.ipynb` cells 28-48 (what the matrix-free / Neumann row is
    for): prior with the constant mean 57.0 and 3^2 Matern-5/2(0.75 w) on scalar inputs, 17 collocation points of
    -kappa u'' = q_src + q_sink with a piecewise-linear heat source, the two NEUMANN blocks -kappa DirectionalDerivative(+-1)
    at the interval's end points (`experiments/cpu.py:214-229`), three noisy digital-thermal-sensor values at the core centres
    (sigma = 0.5) -- every posterior of the chain against the oracle, and the physics: the flux at the boundary."""
    import linpde_gp_amd as lp
    from oracle import gp as ogp
    cf = lp.randprocs.covfuncs
    cpu = _cpu_die_setup(lp)
    width, kappa, bvp = cpu["width"], cpu["kappa"], cpu["bvp"]
    u = lp.GaussianProcess(mean=lp.functions.Constant(input_shape=(), value=57.0),
                           cov=3.0**2 * cf.Matern(input_shape=(), nu=2.5, lengthscales=0.75 * width))
    okern = [(9.0, [("matern", 2.5, 0.75 * width)])]
    X_pde = bvp.domain.uniform_grid(17, inset=0.03 * width)
    Y_pde = bvp.pde.rhs(X_pde)
    assert Y_pde.shape == (17,) and np.all(np.isfinite(Y_pde))
    u_pde = u.condition_on_observations(Y=Y_pde, X=X_pde, L=bvp.pde.diffop)
    blocks = [ogp.ObsBlock(np.asarray(X_pde)[:, None], {(2,): -kappa}, Y_pde)]
    xt = np.linspace(0.0, width, 101)

    def check(gp, blocks_, tol=1e-8):
        post = ogp.condition(okern, blocks_, mean_const=57.0)
        m, v = gp.predict(xt)
        rm, rv = post.mean(xt[:, None]), post.var(xt[:, None])
        assert np.max(np.abs(m - rm)) <= tol * np.max(np.abs(rm)), np.max(np.abs(m - rm)) / np.max(np.abs(rm))
        assert np.max(np.abs(v - rv)) <= tol * np.max(np.abs(rv)), np.max(np.abs(v - rv)) / np.max(np.abs(rv))
        np.testing.assert_allclose(gp.std(xt), np.sqrt(np.maximum(rv, 0.0)), rtol=0, atol=1e-7 * np.sqrt(np.max(rv)))
        return post

    check(u_pde, blocks)
    # Neumann boundary conditions, one at a time, X given as the boundary POINT (cell 43)
    u_nbc = u_pde
    for bc, sign, x_b in zip(bvp.boundary_conditions, (1.0, -1.0), (0.0, width)):
        u_nbc = u_nbc.condition_on_observations(Y=bc.values.value, L=bc.operator, X=bc.boundary)
        blocks = blocks + [ogp.ObsBlock(np.array([[x_b]]), {(1,): -kappa * sign}, np.atleast_1d(float(bc.values.value)))]
        post = check(u_nbc, blocks)
    # the posterior honours the prescribed flux: -kappa u'(0) = q_A, +kappa u'(w) = q_A (read out through the operator)
    flux0 = (-kappa * lp.linfuncops.diffops.DirectionalDerivative(1.0))(u_nbc).mean(np.array([0.0]))
    flux1 = (-kappa * lp.linfuncops.diffops.DirectionalDerivative(-1.0))(u_nbc).mean(np.array([width]))
    assert abs(flux0[0] - cpu["q_A"][0]) < 1e-6 * abs(cpu["q_A"][0]) and abs(flux1[0] - cpu["q_A"][1]) < 1e-6 * abs(cpu["q_A"][1])
    # noisy DTS values at the core centres (cells 34, 48; `experiments/cpu.py:262-276`: the analytic IVP solution + noise, seed 33215)
    X_dts = cpu["centers"]
    y_true = _ivp_solution_1d(cpu["rhs"], kappa, 60.0, -cpu["q_A"][0] / kappa, X_dts)
    y_dts = y_true + 0.5 * np.random.default_rng(33215).standard_normal(3)
    noise = lp.randvars.Normal(np.zeros(3), 0.5**2 * np.eye(3))
    u_dts = u_nbc.condition_on_observations(Y=y_dts, X=X_dts, b=noise)
    blocks = blocks + [ogp.ObsBlock(X_dts[:, None], {(0,): 1.0}, y_dts, 0.0, 0.25)]
    post = check(u_dts, blocks)
    # three sensors of sigma = 0.5 on a 57 +- 3 prior pin the level: the analytic temperature profile lies inside the 3-sigma band
    m, v = u_dts.predict(xt)
    truth = _ivp_solution_1d(cpu["rhs"], kappa, 60.0, -cpu["q_A"][0] / kappa, xt)
    assert np.all(np.abs(m - truth) < 3.0 * np.sqrt(v) + 1e-9)
    np.testing.assert_allclose(u_dts.representer_weights, post.weights, rtol=1e-6, atol=1e-9 * np.max(np.abs(post.weights)))
