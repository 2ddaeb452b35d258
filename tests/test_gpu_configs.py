"""Every BASELINE.json config through the product path (C ABI) against the CPU oracle, with the ONE
posterior criterion of tests/conftest.py (`assert_posterior_close`: 1e-8 relative on the mean,
1e-8 relative on the variance, no absolute slack).

  c1  1-D Poisson, N = 512 + 32 repeated noisy boundary observations      vs oracle
  c2  1-D Poisson, N = 8192 + 2                                           vs oracle (1.8 s)
  c3  2-D Poisson 128x128 + 4x128, M = 64x64  (the metric's config)       vs oracle AT FULL SIZE (slow)
  c4  2-D Poisson 256x256 + 4x256                                         size-independent properties
  c5  heat 1-D space-time, mixed blocks:  N_tot = 9 024                   vs oracle
                                          N_tot = 33 600 (full size)      properties + analytic solution, and
                                                                          vs oracle AT FULL SIZE (slow, ~1 min of host time)
  c4  at full size vs the oracle: tests/test_gpu_zz_c4_full.py (runs last: ~5 min of host time)
(reference test being mirrored: tests/linpde_gp/randprocs/test_posterior_gp.py:152-178,
tests/linpde_gp/problems/test_heat.py:56-99.)  c4 / c5 on several GPUs: tests/test_gpu_dist.py.
"""
import numpy as np
import pytest

from conftest import assert_posterior_close, prior_variance
from oracle import workloads as owl

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lp():
    import linpde_gp_amd
    return linpde_gp_amd


def _run(lp, wl):
    from linpde_gp_amd import problems
    lp.config.gram_capacity_hint = wl.n_total
    try:
        return problems.condition_and_predict(wl)
    finally:
        lp.config.gram_capacity_hint = 0


def _run_fused(lp, wl):
    """The same sequence in the opt-in throughput mode: factorisations enqueued, the prediction riding inside the last one
    (`lpgp_potrf_predict`)."""
    saved = lp.config.lazy_factorization
    lp.config.lazy_factorization = True
    try:
        return _run(lp, wl)
    finally:
        lp.config.lazy_factorization = saved


def _vs_oracle(lp, wl, ref=None):
    """BOTH modes of the package against ONE oracle run: the default (status read back inside every conditioning, prediction
    as a second pipeline) and the fused factor-and-predict pipeline; and against each other."""
    u, mean, var = _run(lp, wl)
    ref = owl.run(wl) if ref is None else ref
    rm, rv = assert_posterior_close(mean, var, ref["mean"], ref["var"])
    uf, mean_f, var_f = _run_fused(lp, wl)
    fm, fv = assert_posterior_close(mean_f, var_f, ref["mean"], ref["var"])
    del uf
    print(f"{wl.name}: N_tot={wl.n_total} M={wl.Xtest.shape[0]} mean err {rm:.2e} x tol, var err {rv:.2e} x tol "
          f"(fused pipeline: {fm:.2e}, {fv:.2e}); oracle {ref['seconds']['total']:.1f} s")
    return u, mean, var, ref


def test_c1_poisson1d_repeated_noisy_boundary(lp):
    """c1: N = 512 collocation + 16 noisy repeats per endpoint (sigma^2 = 1e-4): the repeated rows make the
    boundary block singular without its noise (SURVEY.md:412)."""
    from linpde_gp_amd import problems
    wl = problems.poisson_1d(512, n_bdry_repeats=16, noise_var=1e-4, m=256)
    assert wl.n_total == 512 + 32
    u, mean, var, ref = _vs_oracle(lp, wl)
    np.testing.assert_allclose(u.representer_weights, ref["weights"], rtol=0,
                               atol=1e-7 * np.max(np.abs(ref["weights"])))
    # -u'' = pi^2 sin(pi x), u(+-1) = 0  =>  u = sin(pi x); 32 noisy boundary values leave ~1e-3 of slack
    assert np.max(np.abs(mean - np.sin(np.pi * wl.Xtest[:, 0]))) < 5e-3


def test_c2_poisson1d_8192(lp):
    from linpde_gp_amd import problems
    wl = problems.poisson_1d()
    assert wl.n_total == 8194
    u, mean, var, _ = _vs_oracle(lp, wl)
    assert np.max(np.abs(mean - np.sin(np.pi * wl.Xtest[:, 0]))) < 1e-5


@pytest.mark.slow
def test_c3_poisson2d_full_size_vs_oracle(lp):
    """The metric's own configuration, mean AND variance on the whole 64x64 prediction grid against the
    oracle at N_tot = 16 896 (2.3 GB Gram, LAPACK dpotrf + dtrtrs on the host cores)."""
    from linpde_gp_amd import problems
    wl = problems.poisson_2d()
    assert wl.n_total == 16896 and wl.Xtest.shape[0] == 4096
    u, mean, var, _ = _vs_oracle(lp, wl)
    assert abs(mean.max() - 0.5894) < 2e-2


def test_c5_heat_mixed_blocks_vs_oracle(lp):
    """c5's workload (IC + 2 BC + heat-operator collocation + noisy interior VALUE observations: mixed
    differential / functional blocks, five conditionings) at N_tot = 9 024."""
    from linpde_gp_amd import problems
    wl = problems.heat_1d(nt=128, nx=64, m_side=32)
    assert wl.n_total == 128 * 64 + 64 + 2 * 256 + 256
    _vs_oracle(lp, wl)


def test_c5_heat_full_size_properties(lp):
    """c5 at full size (512x64 collocation, N_tot = 33 600): residual of G w = r on re-evaluated rows,
    variance inside [0, k(x,x)], and the analytic solution (`problems/pde/_heat.py:96-132`) within the
    reference's own accuracy bar (test_heat.py:25-28: 3e-2)."""
    from linpde_gp_amd import problems
    wl = problems.heat_1d()
    assert wl.n_total == 33600
    u, mean, var = _run(lp, wl)
    assert np.all(np.isfinite(mean)) and np.all(var > -1e-10) and np.all(var < prior_variance(wl))
    res = problems.row_residual(u, wl, np.array([0, 63, 64 * 200 + 31, 32767]))
    assert np.max(np.abs(res)) < 1e-6
    assert np.max(np.abs(mean - problems.analytic_solution(wl))) < 3e-2


@pytest.mark.slow
def test_c5_heat_full_size_vs_oracle(lp):
    """c5 at FULL size (N_tot = 33 600, M = 4 096) against the oracle (`oracle.workloads.run_in_place`: 9 GB factor, LAPACK
    dpotrf + dtrtrs on the host cores), mean and variance on the whole prediction grid with the one criterion."""
    import os
    import psutil
    from linpde_gp_amd import problems
    wl = problems.heat_1d()
    assert wl.n_total == 33600 and wl.Xtest.shape[0] == 4096
    workers = max(1, min(16, (os.cpu_count() or 1) // 8))
    need = owl.host_memory_needed(wl, 1024, workers)
    if psutil.virtual_memory().available < 1.2 * need:
        pytest.skip(f"full-size oracle needs {need / 1e9:.0f} GB of host memory")
    u, mean, var = _run(lp, wl)
    del u
    ref = owl.run_in_place(wl, chunk=1024, workers=workers)
    rm, rv = assert_posterior_close(mean, var, ref["mean"], ref["var"])
    uf, mean_f, var_f = _run_fused(lp, wl)            # the fused factor-and-predict pipeline against the same oracle run
    del uf
    fm, fv = assert_posterior_close(mean_f, var_f, ref["mean"], ref["var"])
    print(f"{wl.name}: N_tot={wl.n_total} M={wl.Xtest.shape[0]} mean err {rm:.2e} x tol, var err {rv:.2e} x tol "
          f"(fused pipeline: {fm:.2e}, {fv:.2e}); oracle {ref['seconds']['total']:.1f} s ({workers} assembly threads)")


def test_c4_poisson2d_256_properties(lp):
    """c4 on one GPU (N_tot = 66 560: 35.4 GB factor in HBM): the properties of test_full_size_properties
    at four times the matrix order."""
    from linpde_gp_amd import problems
    wl = problems.poisson_2d(256, m_side=24)
    assert wl.n_total == 66560
    u, mean, var = _run(lp, wl)
    assert np.all(np.isfinite(mean)) and np.all(var > -1e-9) and np.all(var < 4.0)
    res = problems.row_residual(u, wl, np.array([0, 255, 256 * 100 + 17, 65535]))
    assert np.max(np.abs(res)) < 1e-5 * 2.0
    Xs = wl.Xtest[:8]
    C = u.cov.matrix(Xs)
    np.testing.assert_allclose(C, C.T, rtol=0, atol=1e-12)
    np.testing.assert_allclose(np.diag(C), var[:8], rtol=1e-7, atol=1e-11)
    assert abs(mean.max() - 0.5894) < 1e-2
    del u
