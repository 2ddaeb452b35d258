"""Host mirror of the reference's operator interface: everything that needs no GPU.

Covers the coefficient algebra the reference pins in
`tests/linpde_gp/linfuncops/diffops/test_coefficients.py:61-144`, the lowering of
(kernel, L0, L1) to the C-ABI descriptor, shape/argument validation, and that liblpgp.so
loads and exports every symbol declared in include/lpgp.h."""
import os
import re

import numpy as np
import pytest

import linpde_gp_amd as lp
from linpde_gp_amd import _lib
from linpde_gp_amd.linfuncops import diffops
from linpde_gp_amd.randprocs import covfuncs as cf


def test_c_abi_exports_every_declared_symbol():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "lpgp.h")).read()
    declared = set(re.findall(r"\b(lpgp_[a-z0-9_]+)\s*\(", header))
    declared -= {"lpgp_kernel_id", "lpgp_family"}
    assert declared, "no declarations found"
    for name in sorted(declared):
        assert hasattr(_lib.lib, name), f"{name} declared in lpgp.h but not exported by liblpgp.so"
    assert set(_lib.EXPORTED) <= declared


def test_product_library_exports_no_test_hooks():
    """the raw-kernel test entry points, probes and diagnostics live in a library of their own
    (include/lpgp_test.h, liblpgp_testhooks.so: tests/_hooks.py), not in the shipping liblpgp.so."""
    import subprocess
    import _hooks
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "lpgp_test.h")).read()
    declared = set(re.findall(r"\b(lpgp_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_hooks.EXPORTED)
    for name in sorted(declared):
        assert hasattr(_hooks.lib, name), f"{name} declared in lpgp_test.h but not exported by liblpgp_testhooks.so"
    syms = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    leaked = [ln for ln in syms.splitlines() if re.search(r"\blpgp_(test|probe|debug)_", ln)]
    assert not leaked, leaked
    # ... and the product package does not import SciPy or the hooks
    import linpde_gp_amd, pkgutil
    pkg_dir = os.path.dirname(linpde_gp_amd.__file__)
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert "import scipy" not in text and "testhooks" not in text and "import oracle" not in text and "from oracle" not in text, os.path.join(dirpath, f)


def test_multi_index_and_coefficients_algebra():
    mi = diffops.MultiIndex((1, 2, 0))
    assert mi.order == 3 and mi.is_mixed and mi == diffops.MultiIndex([1, 2, 0]) and hash(mi) == hash(diffops.MultiIndex([1, 2, 0]))
    with pytest.raises(ValueError):
        diffops.MultiIndex((1, -1))
    assert diffops.MultiIndex.from_index((1,), (3,), 2) == diffops.MultiIndex((0, 2, 0))
    a = diffops.PartialDerivativeCoefficients({(): {diffops.MultiIndex((2, 0)): 1.0, diffops.MultiIndex((0, 2)): 1.0}}, (2,), ())
    b = diffops.PartialDerivativeCoefficients({(): {diffops.MultiIndex((2, 0)): 3.0, diffops.MultiIndex((1, 1)): -1.0}}, (2,), ())
    s = a + b
    assert s[()][diffops.MultiIndex((2, 0))] == 4.0 and s[()][diffops.MultiIndex((1, 1))] == -1.0 and s.num_entries == 3
    assert (-a)[()][diffops.MultiIndex((0, 2))] == -1.0
    assert (2.5 * a)[()][diffops.MultiIndex((2, 0))] == 2.5
    assert (a - a)[()][diffops.MultiIndex((2, 0))] == 0.0
    assert s.has_mixed and not a.has_mixed
    with pytest.raises(ValueError):
        a + diffops.PartialDerivativeCoefficients({(): {diffops.MultiIndex((2,)): 1.0}}, (1,), ())
    with pytest.raises(ValueError):
        diffops.PartialDerivativeCoefficients({(): {diffops.MultiIndex((2,)): 1.0}}, (2,), ())
    with pytest.raises(ValueError):
        diffops.PartialDerivativeCoefficients({(1,): {diffops.MultiIndex((2,)): 1.0}}, (1,), ())


def test_operator_classes():
    lap = diffops.Laplacian((2,))
    assert lap.coefficients_dict() == {(2, 0): 1.0, (0, 2): 1.0}
    assert (-0.5 * lap).coefficients_dict() == {(2, 0): -0.5, (0, 2): -0.5}
    assert (3.0 * (-0.5 * lap)).scalar == pytest.approx(-1.5)
    assert diffops.WeightedLaplacian([0.0, 2.0]).coefficients_dict() == {(0, 2): 2.0}
    assert diffops.SpatialLaplacian((3,)).coefficients_dict() == {(0, 2, 0): 1.0, (0, 0, 2): 1.0}
    assert diffops.TimeDerivative((2,)).coefficients_dict() == {(1, 0): 1.0}
    assert diffops.HeatOperator((2,), alpha=0.1).coefficients_dict() == {(1, 0): 1.0, (0, 2): -0.1}
    assert diffops.DirectionalDerivative([0.5, -2.0]).coefficients_dict() == {(1, 0): 0.5, (0, 1): -2.0}
    assert diffops.Derivative(2).coefficients_dict() == {(2,): 1.0}
    assert (lap + diffops.TimeDerivative((2,))).coefficients_dict() == {(2, 0): 1.0, (0, 2): 1.0, (1, 0): 1.0}
    with pytest.raises(ValueError):
        diffops.HeatOperator(((2, 2)))
    with pytest.raises(ValueError):
        diffops.SpatialLaplacian((1,))
    with pytest.raises(ValueError):
        diffops.Derivative(-1)


def test_lowering_poisson_and_heat():
    k = 2.0**2 * cf.TensorProduct(cf.Matern((), nu=2.5, lengthscales=1.0), cf.Matern((), nu=2.5, lengthscales=0.5))
    D = -1.0 * diffops.Laplacian((2,))
    (g,) = D(D(k, argnum=1), argnum=0).lower()
    assert g["d"] == 2 and g["scale"] == 4.0 and g["family"] == [1, 1] and g["p"] == [2, 2] and g["lengthscale"] == [1.0, 0.5]
    assert sorted(g["terms"]) == sorted([(1.0, (2, 0), (2, 0)), (1.0, (2, 0), (0, 2)), (1.0, (0, 2), (2, 0)), (1.0, (0, 2), (0, 2))])
    (gc,) = D(k, argnum=1).lower()
    assert sorted(gc["terms"]) == sorted([(-1.0, (0, 0), (2, 0)), (-1.0, (0, 0), (0, 2))])
    kh = cf.TensorProduct(cf.Matern((), nu=1.5, lengthscales=2.5), cf.Matern((), nu=2.5, lengthscales=2.0))
    H = diffops.HeatOperator((2,), alpha=0.1)
    (gh,) = H(H(kh, argnum=1), argnum=0).lower()
    assert {(a, b) for _, a, b in gh["terms"]} == {((1, 0), (1, 0)), ((1, 0), (0, 2)), ((0, 2), (1, 0)), ((0, 2), (0, 2))}
    # sum kernels give one group per summand, scalars distribute
    groups = (1.5 * cf.ExpQuad((2,), lengthscales=[0.4, 1.3]) + k).lower()
    assert len(groups) == 2 and groups[0]["family"] == [2, 2] and groups[0]["scale"] == 1.5
    # ctypes marshalling
    arr = _lib.make_kdesc_array(groups)
    assert arr[1].nterms == 1 and arr[1].p[0] == 2 and arr[0].lengthscale[1] == 1.3


def test_lowering_errors():
    k = cf.TensorProduct(cf.Matern((), nu=1.5), cf.Matern((), nu=2.5))
    lap = diffops.Laplacian((2,))
    with pytest.raises(ValueError):           # Matern-3/2 is not 4x differentiable
        lap(lap(k, argnum=1), argnum=0).lower()
    with pytest.raises(ValueError):           # dimension mismatch
        diffops.Laplacian((3,))(k, argnum=0)
    with pytest.raises(NotImplementedError):  # no closed form for non-half-integer nu
        cf.Matern((), nu=2.2)
    # multivariate isotropic Matern: identity and directional derivatives only
    kiso = cf.Matern((2,), nu=2.5, lengthscales=[0.5, 2.0])
    dd = diffops.DirectionalDerivative(np.array([1.0, -2.0]))
    g, = dd(dd(kiso, argnum=1), argnum=0).lower()
    assert g["family"] == [lp._lib.MATERN_ISO] * 2 and g["p"] == [2, 2] and g["lengthscale"] == [0.5, 2.0]
    assert sorted((c, a, b) for c, a, b in g["terms"]) == sorted(
        [(1.0, (1, 0), (1, 0)), (-2.0, (1, 0), (0, 1)), (-2.0, (0, 1), (1, 0)), (4.0, (0, 1), (0, 1))])
    with pytest.raises(NotImplementedError):  # the reference falls back to JAX autodiff here
        lap(kiso, argnum=0).lower()
    with pytest.raises(ValueError):           # not enough differentiability (cases_matern.py:64-65)
        dd(dd(cf.Matern((2,), nu=1.5), argnum=1), argnum=0).lower()
    with pytest.raises(ValueError):
        cf.TensorProduct(cf.Matern((1,), nu=2.5))
    with pytest.raises(ValueError):
        cf.ExpQuad((2,), lengthscales=[1.0, -1.0])


def test_functionals_and_shapes():
    X = np.zeros((4, 3, 2))
    L = diffops.Laplacian((2,)).to_linfunctl(X)
    assert L.output_shape == (4, 3) and L.points().shape == (12, 2)
    assert L.coefficients_dict() == {(2, 0): 1.0, (0, 2): 1.0}
    ev = lp.linfunctls._EvaluationFunctional((2,), (), X)
    assert ev.coefficients_dict() == {(0, 0): 1.0}
    assert np.array_equal(ev(lp.functions.Constant((2,), 3.0)), np.full((4, 3), 3.0))
    assert np.array_equal(L(lp.functions.Constant((2,), 3.0)), np.zeros((4, 3)))
    with pytest.raises(ValueError):
        lp.linfunctls._EvaluationFunctional((2,), (), np.zeros((4, 3)))
    with pytest.raises(ValueError):
        lp.GaussianProcess(lp.functions.Zero((2,)), cf.Matern((1,), nu=2.5))
    with pytest.raises(TypeError):
        lp.GaussianProcess(lambda x: x, cf.Matern((1,), nu=2.5))


def test_normal_randvar():
    n = lp.randvars.Normal(np.zeros(3), 0.25)
    assert np.array_equal(n.cov, 0.25 * np.eye(3)) and np.array_equal(n.cov_diag, np.full(3, 0.25))
    d = lp.randvars.Normal(np.zeros(3), np.array([1.0, 2.0, 3.0]))
    assert np.array_equal(d.var, [1.0, 2.0, 3.0]) and d.cov.shape == (3, 3)
    full = lp.randvars.Normal(np.zeros(2), np.array([[2.0, 0.5], [0.5, 1.0]]))
    assert full.cov_diag is None and np.allclose(full.std, np.sqrt([2.0, 1.0]))
    with pytest.raises(ValueError):
        lp.randvars.Normal(np.zeros(3), np.eye(2))


def test_no_cpu_fallback():
    """Without a GPU every evaluation raises (there is no CPU path in the product)."""
    import ctypes
    ndev = ctypes.c_int(0)
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        rc = hip.hipGetDeviceCount(ctypes.byref(ndev))
    except OSError:
        rc, ndev = 1, ctypes.c_int(0)
    if rc == 0 and ndev.value > 0:
        pytest.skip("a GPU is visible")
    prior = lp.GaussianProcess(lp.functions.Zero((1,)), cf.Matern((1,), nu=2.5))
    with pytest.raises(_lib.LpgpError):
        prior.condition_on_observations(np.zeros(2), np.array([[0.0], [1.0]]))


def test_domains_and_problem_builders():
    from linpde_gp_amd import domains
    from linpde_gp_amd.problems import pde
    box = domains.Box([[-1.0, 1.0], [0.0, 2.0]])
    assert box.shape == (2,) and len(box.boundary) == 4 and box.volume == 4.0 and [0.0, 1.0] in box
    g = box.uniform_grid((3, 4))
    assert isinstance(g, domains.TensorProductGrid) and g.shape == (3, 4, 2) and len(g.factors) == 2
    np.testing.assert_array_equal(g[:, 0, 0], [-1.0, 0.0, 1.0])
    edge = box.boundary[0]
    ge = edge.uniform_grid(5, inset=1e-6)
    assert ge.shape == (1, 5, 2) and np.all(ge[..., 0] == -1.0) and ge[0, 0, 1] == pytest.approx(1e-6)
    iv = domains.asdomain([-1.0, 1.0])
    assert isinstance(iv, domains.Interval) and [float(np.asarray(p)) for p in iv.boundary] == [-1.0, 1.0]
    np.testing.assert_allclose(iv.uniform_grid(3, inset=0.5), [-0.5, 0.0, 0.5])
    with pytest.raises(ValueError):
        domains.Interval(1.0, 0.0)
    with pytest.raises(ValueError):
        domains.Box([[0.0, 1.0, 2.0]])
    bvp = pde.PoissonEquationDirichletProblem(box, rhs=lp.functions.Constant((2,), 2.0))
    assert bvp.pde.diffop.coefficients_dict() == {(2, 0): -1.0, (0, 2): -1.0} and len(bvp.boundary_conditions) == 4
    b1 = pde.PoissonEquationDirichletProblem(iv, rhs=lp.functions.Constant((), 2.0), boundary_values=(0.0, 1.0))
    # -u'' = 2, u(-1) = 0, u(1) = 1  =>  u = (x+1)/2 + (1 - x^2)
    np.testing.assert_allclose(b1.solution(np.array([-1.0, 0.0, 1.0])), [0.0, 1.5, 1.0])
    h = pde.HeatEquationDirichletProblem(0.0, iv, T=5.0, alpha=0.1, initial_values=pde.TruncatedSineSeries(iv, [1.0, 2.0]))
    assert h.pde.diffop.coefficients_dict() == {(1, 0): 1.0, (0, 2): -0.1}
    assert h.initial_domain.uniform_grid(5).shape == (1, 5, 2)
    x = np.linspace(-1, 1, 7)
    np.testing.assert_allclose(h.solution(np.stack([np.zeros(7), x], -1)), h.initial_condition.values(x), atol=1e-14)


def test_functional_arithmetic():
    """`-L`, `a * L`, `L1 + L2`, `L1 - L2`, `L @ D`, DiracFunctional (`linfunctls/_linfunctl.py:76-112`,
    `_arithmetic.py:13-174`, `_dirac.py:10-45`): everything lowers to one coefficient map over one point set."""
    lf = lp.linfunctls
    X = np.linspace(0.0, 1.0, 12).reshape(6, 2)
    ev = lf._EvaluationFunctional((2,), (), X)
    dn = ev @ diffops.DirectionalDerivative(np.array([0.0, -1.0]))
    robin = 2.0 * ev + 0.5 * dn
    assert isinstance(robin, lf.SumLinearFunctional) and robin.output_shape == (6,)
    assert robin.coefficients_dict() == {(0, 0): 2.0, (0, 1): -0.5}
    assert np.array_equal(robin.points(), X)
    assert (-ev).coefficients_dict() == {(0, 0): -1.0}
    assert (3.0 * (2.0 * ev)).scalar == 6.0
    assert (ev - dn).coefficients_dict() == {(0, 0): 1.0, (0, 1): 1.0}
    dirac = lf.DiracFunctional((2,), (), X)
    assert dirac.output_shape == (6,) and dirac.X_batch_ndim == 1 and dirac.coefficients_dict() == {(0, 0): 1.0}
    f = lp.functions.Constant((2,), 3.0)
    assert np.array_equal(robin(f), np.full(6, 6.0)) and np.array_equal(dirac(f), np.full(6, 3.0))
    with pytest.raises(NotImplementedError):          # different point sets: not one observation block
        (ev + lf._EvaluationFunctional((2,), (), X + 1.0)).points()
    with pytest.raises(ValueError):
        lf.SumLinearFunctional(ev, lf._EvaluationFunctional((1,), (), X[:, :1]))
    with pytest.raises(ValueError):
        lf.ScaledLinearFunctional(ev, np.ones(2))
    with pytest.raises(TypeError):
        np.ones(3) * ev


def test_covariance_classes():
    """`randvars.Covariance` family (`randvars/_covariance.py:13-224` of the reference): shapes, both
    representations, C-order flattening, arithmetic, error behaviour."""
    rv = lp.randvars
    A = np.arange(24.0).reshape(2, 3, 4)
    c = rv.ArrayCovariance(A, (2, 3), (4,))
    assert (c.shape0, c.shape1, c.ndim0, c.ndim1, c.size0, c.size1) == ((2, 3), (4,), 2, 1, 6, 4)
    assert c.array is not None and np.array_equal(c.matrix, A.reshape(6, 4)) and c.linop.shape == (6, 4)
    np.testing.assert_array_equal(c.linop @ np.ones(4), A.reshape(6, 4).sum(1))
    np.testing.assert_array_equal(c.flatten0(np.arange(6).reshape(2, 3)), np.arange(6))
    with pytest.raises(ValueError):
        c.flatten1(np.zeros(3))
    with pytest.raises(ValueError):
        rv.ArrayCovariance(A, (2, 3), (5,))
    s = rv.ArrayCovariance.from_scalar(2.5)
    assert s.shape0 == () and s.matrix.shape == (1, 1) and float(s.array) == 2.5
    assert np.array_equal((-c).array, -A) and np.array_equal((2.0 * c).array, 2.0 * A)
    assert np.array_equal((c + c).array, 2 * A) and np.array_equal((c - c).array, 0 * A)
    lo = rv.LinearOperatorCovariance(A.reshape(6, 4), (2, 3), 4)           # int shape = (4,)
    assert lo.shape1 == (4,) and np.array_equal(lo.array, A) and np.array_equal(lo.matrix, A.reshape(6, 4))
    assert isinstance(lo + c, rv.ArrayCovariance) and np.array_equal((c + lo).array, 2 * A)
    np.testing.assert_array_equal(lo.linop.T @ np.ones(6), A.reshape(6, 4).sum(0))
    with pytest.raises(ValueError):
        rv.LinearOperatorCovariance(A.reshape(6, 4), (2, 3), (3,))
    with pytest.raises(TypeError):
        c + 1.0


def test_mean_functions_with_closed_form_derivatives():
    """Non-constant prior means under differential operators (`functions/_polynomial.py:39-98`, `_affine.py:9-51`; the
    reference differentiates them by JAX autodiff, `diffops/_lindiffop.py:104-129` -- here in closed form)."""
    F = lp.functions
    p = F.Polynomial([1.0, -2.0, 0.5, 3.0])                       # 1 - 2x + x^2/2 + 3x^3
    x = np.linspace(-2.0, 2.0, 17)
    np.testing.assert_allclose(p(x), np.polyval([3.0, 0.5, -2.0, 1.0], x), rtol=1e-15)
    assert p.degree == 3 and p.differentiate().coefficients == (-2.0, 1.0, 9.0)
    np.testing.assert_allclose(p.integrate().differentiate().coefficients, p.coefficients)
    assert (p + F.Constant((), 2.0)).coefficients[0] == 3.0 and (2.0 * p - p).coefficients == p.coefficients
    assert F.Monomial(3).coefficients == (0.0, 0.0, 0.0, 1.0) and F.Polynomial([4.0]).differentiate().coefficients == (0.0,)
    # operators in canonical form: -alpha u'' + beta u, one-dimensional
    L = -0.7 * diffops.Laplacian(()) + 1.5 * lp.linfuncops.Identity(())
    Lp = L(p)
    np.testing.assert_allclose(Lp(x), -0.7 * np.polyval(np.polyder([3.0, 0.5, -2.0, 1.0], 2), x) + 1.5 * p(x), rtol=1e-14)
    # through a functional (what `condition_on_observations` evaluates as L[m](X))
    np.testing.assert_allclose(L.to_linfunctl(x)(p), Lp(x))
    # affine mean of the real line
    a = F.Affine(2.0, -1.0)
    np.testing.assert_allclose(diffops.PartialDerivative(diffops.MultiIndex((1,)))(a)(x), np.full_like(x, 2.0))
    np.testing.assert_allclose(L(a)(x), 1.5 * (2.0 * x - 1.0))
    # vector-valued affine maps evaluate with the reference's shape rules
    A = F.Affine(np.array([[1.0, 2.0], [0.0, -1.0], [3.0, 0.5]]), np.array([0.5, 0.0, -1.0]))
    assert A.input_shape == (2,) and A.output_shape == (3,)
    X2 = np.random.default_rng(0).standard_normal((5, 2))
    np.testing.assert_allclose(A(X2), X2 @ A.A.T + A.b)
    # a callable with analytic derivatives supplied by the caller, 2-D Laplacian
    f = F.LambdaFunction(lambda z: np.sin(z[..., 0]) * z[..., 1] ** 2, (2,),
                         derivatives={(2, 0): lambda z: -np.sin(z[..., 0]) * z[..., 1] ** 2,
                                      (0, 2): lambda z: 2.0 * np.sin(z[..., 0])})
    np.testing.assert_allclose(diffops.Laplacian((2,))(f)(X2), -np.sin(X2[:, 0]) * X2[:, 1] ** 2 + 2.0 * np.sin(X2[:, 0]))
    with pytest.raises(NotImplementedError):                       # a derivative nobody supplied
        diffops.PartialDerivative(diffops.MultiIndex((1, 0)))(f)
    with pytest.raises(NotImplementedError):                       # no derivatives at all: the reference's JAX fallback
        diffops.Laplacian((2,))(F.LambdaFunction(lambda z: z[..., 0], (2,)))


def test_spawn_front_reports_a_failed_bring_up():
    """The single-process multi-GPU front (`_spawn.WorkerGroup`) on a box WITHOUT a GPU: the worker processes start (fresh
    interpreters), fail to open a device, and the caller gets ONE error that names the cause -- no hang, no orphan."""
    import subprocess, sys, os, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {os.path.join(root, 'linpde-gp_amd')!r})
        import linpde_gp_amd as lp
        try:
            lp.spawn(2, transport="host", timeout=120.0)
        except RuntimeError as exc:
            print("SPAWN-ERR", str(exc)[:300].replace("\\n", " "))
        else:
            print("SPAWN-UP")
    """)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert "SPAWN-ERR" in out.stdout and "bring-up failed" in out.stdout, out.stdout + out.stderr[-2000:]


def test_tensor_product_grid_survives_pickling():
    import pickle
    import numpy as np
    from linpde_gp_amd.domains import TensorProductGrid
    g = TensorProductGrid(np.linspace(0, 1, 3), np.linspace(-1, 1, 5))
    h = pickle.loads(pickle.dumps(g))
    assert type(h) is TensorProductGrid and h.shape == (3, 5, 2) and np.array_equal(h, g)
    assert all(np.array_equal(a, b) for a, b in zip(h.factors, g.factors))


def test_bench_reports_pmc_traffic_only_from_the_running_sources(tmp_path):
    """`bench.py` cannot measure FETCH_SIZE / WRITE_SIZE in-process: `roofline.traffic` comes from a committed rocprofv3
    summary -- and only from one collected on the kernel sources the process is running (`config.csrc_sha16`); a summary of
    other sources, or of another roofline kernel, is named under `traffic_stale` instead."""
    import importlib.util, json, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lpgp_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sha = bench.csrc_sha16()
    assert len(sha) == 16 and sha == bench.csrc_sha16()
    sym = "gemm_f64_kernel<false, false, 1>"

    def write(name, at, kernel, value):
        with open(tmp_path / name, "w") as f:
            json.dump({"syrk_hbm_bytes_per_launch": value, "syrk_kernel": f"void lpgp::{kernel}(lpgp::GemmArgs)",
                       "bench_line": {"config": {"csrc_sha16": at}}}, f)

    write("r02_bench_c3_summary.json", None, sym, 1.9e9)                       # (round 2 recorded no source identity)
    got = bench.pmc_traffic(sha, sym, str(tmp_path))
    assert got["traffic"] is None and got["traffic_stale"]["value"] == 1.9e9
    write("r03_bench_c3_summary.json", "0" * 16, sym, 1.8e9)                    # other sources
    got = bench.pmc_traffic(sha, sym, str(tmp_path))
    assert got["traffic"] is None and got["traffic_stale"]["collected_at_csrc"] == "0" * 16
    write("r04_bench_c3_summary.json", sha, "gemm3_f64_kernel<false, 1>", 2.2e9)    # these sources, another roofline kernel
    assert bench.pmc_traffic(sha, sym, str(tmp_path))["traffic"] is None
    write("r05_bench_c3_summary.json", sha, sym, 1.7e9)
    got = bench.pmc_traffic(sha, sym, str(tmp_path))
    assert got["traffic"] == 1.7e9 and got["traffic_collected_at_csrc"] == sha and "r05_" in got["traffic_source"]
    # the collection must be of the SAME bench command: c2 / c4 / c5 have summaries of their own, any other workload none
    assert bench.pmc_traffic(sha, sym, str(tmp_path), tag="c2")["traffic"] is None
    write("r05_bench_c2_summary.json", sha, sym, 4.6e8)
    assert bench.pmc_traffic(sha, sym, str(tmp_path), tag="c2")["traffic"] == 4.6e8
    assert bench.pmc_traffic(sha, sym, str(tmp_path), tag=None) == {"traffic": None, "traffic_source": None,
                                                                     "traffic_note": bench.pmc_traffic(sha, sym, str(tmp_path), tag=None)["traffic_note"]}
    assert [bench.profile_tag(*a) for a in (("poisson2d", 128, 64), ("poisson2d", 256, 128), ("poisson2d", 64, 32), ("poisson1d",), ("heat1d",),
                                             ("scattered2d",), ("poisson1d_c1",), ("heat_reference",))] == ["c3", "c4", None, "c2", "c5", None, None, None]


def test_bench_gpus_n_without_launcher_fails_once_on_a_box_without_gpus():
    """`python bench.py --gpus 2` called plainly starts its own ranks -- and on a box with fewer GPUs than asked for it
    prints ONE error, no JSON line, and exits non-zero (round 3: it silently measured one GPU and printed n_gpus 1)."""
    import importlib.util
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    spec = importlib.util.spec_from_file_location("lpgp_bench_l", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    if (bench.visible_gpus() or 0) >= 2:
        pytest.skip("two GPUs visible: this is the failure-path test")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert res.returncode != 0
    assert "{" not in res.stdout
    assert len([ln for ln in res.stderr.splitlines() if ln.startswith("bench.py:")]) == 1
    if (bench.visible_gpus() or 0) == 0:
        # the ranks' own failure path: the launcher believes there are two GPUs, the ranks cannot open one
        res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu"],
                             env=dict(env, LPGP_BENCH_ASSUME_GPUS="2"), capture_output=True, text=True, timeout=120)
        assert res.returncode != 0 and "{" not in res.stdout
        assert len([ln for ln in res.stderr.splitlines() if ln.startswith("bench.py:")]) == 1
        assert "no line printed" in res.stderr
    # a launcher-given WORLD_SIZE that disagrees with --gpus is refused as well (never a line for another count)
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                         env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and "{" not in res.stdout


def test_pcg_and_preconditioner_on_a_plain_matrix():
    """The iteration itself (host side), on a dense SPD matrix with a low-rank + noise structure."""
    from linpde_gp_amd.randprocs import _matrix_free as mf
    rng = np.random.default_rng(0)
    x = np.sort(rng.uniform(-1, 1, 300))
    A = np.exp(-0.5 * (x[:, None] - x[None, :]) ** 2 / 0.2**2) + 1e-4 * np.eye(300)        # kernel matrix + noise: cond ~ 1e6
    B = rng.standard_normal((300, 5))

    class Dense:
        n = 300
        def diag(self): return np.diag(A).copy()
        def row(self, p): return A[p].copy()
    X0, i0 = mf.pcg(lambda V: A @ V, B, None, rtol=1e-11, maxiter=3000)
    P = mf.PivotedCholeskyPreconditioner(Dense(), rank=40)
    X1, i1 = mf.pcg(lambda V: A @ V, B, P, rtol=1e-11, maxiter=3000)
    ref = np.linalg.solve(A, B)
    assert i0["converged"] and i1["converged"] and 3 * i1["iterations"] < i0["iterations"], (i0["iterations"], i1["iterations"])
    np.testing.assert_allclose(X0, ref, rtol=0, atol=1e-7 * np.abs(ref).max())
    np.testing.assert_allclose(X1, ref, rtol=0, atol=1e-7 * np.abs(ref).max())
    x, info = mf.pcg(lambda V: A @ V, B[:, 0], P, rtol=1e-11)
    assert x.shape == (300,) and info["converged"]
