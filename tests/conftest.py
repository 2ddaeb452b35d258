import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "linpde-gp_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: full-size BASELINE configuration replayed on the CPU oracle (tens of seconds of host time)")


_gpu_state = {}


def _gpu_available() -> bool:
    """True if `lpgp_init` succeeds (a HIP device is visible).  Decided once per session."""
    if "ok" not in _gpu_state:
        try:
            from linpde_gp_amd import _engine
            _engine.default_context()
            _gpu_state["ok"] = True
        except Exception as exc:  # noqa: BLE001
            _gpu_state["ok"] = False
            _gpu_state["why"] = f"{type(exc).__name__}: {exc}"
    return _gpu_state["ok"]


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a GPU: gpu-marked tests are skipped, not failed.
    With `-m gpu` on a box without a GPU they FAIL instead (the driver's GPU run must not go green
    on a machine that lost its device)."""
    gpu_items = [it for it in items if it.get_closest_marker("gpu") is not None]
    if not gpu_items:
        return
    explicit = "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or "")
    if explicit or _gpu_available():
        return
    skip = pytest.mark.skip(reason="no HIP device: " + _gpu_state.get("why", ""))
    for it in gpu_items:
        it.add_marker(skip)


@pytest.fixture
def kronecker_everywhere():
    """Tensor-grid blocks of ANY size through the Kronecker assembly (`lpgp_gram_assemble_grid`): by default grids below
    `config.grid_assembly_min_points` points take the per-entry kernel (faster there); the full-size c3 / c4 / c5 tests go through the
    Kronecker path as the product does."""
    from linpde_gp_amd import config
    saved = config.grid_assembly_min_points
    config.grid_assembly_min_points = 0
    yield
    config.grid_assembly_min_points = saved


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# ---------------------------------------------------------------------------------------------
# THE parity criterion for posterior mean and marginal variance (north_star: "posterior within
# 1e-8 rel-err of the CPU reference"; SURVEY.md §8d "Parity bar").  Stated once, used by every
# GPU-vs-oracle posterior comparison in tests/, by bench.py --check and by smoke().
#
#   mean:      max|mean - ref| <= 1e-8 * max|ref_mean|
#   variance:  max|var  - ref| <= 1e-8 * max|ref_var|
#
# Nothing else: no absolute slack.  (Until the middle of round 2 the variance bound carried an extra
# 2 sqrt(N_tot) eps k(x,x) "rounding floor".  It was covering for the ORACLE: `np.sum(V * V, axis=0)`
# adds the N_tot rows one after another, and at full c5 that naive accumulation alone sits 1.6e-13 =
# 2e-8 of max var away from the long-double sum while the device is 6e-15 away from it
# (profiles/r02_c5_full_parity.txt).  The oracle now sums pairwise (`oracle.gp.colsumsq`) and every
# GPU test passes the plain bound.)
# ---------------------------------------------------------------------------------------------
POSTERIOR_RTOL = 1e-8


def posterior_tolerances(ref_mean, ref_var):
    return POSTERIOR_RTOL * float(np.max(np.abs(ref_mean))), POSTERIOR_RTOL * float(np.max(np.abs(ref_var)))


def assert_posterior_close(mean, var, ref_mean, ref_var):
    """Returns the measured (mean error / mean_atol, var error / var_atol) ratios (both <= 1)."""
    mean_atol, var_atol = posterior_tolerances(ref_mean, ref_var)
    em = float(np.max(np.abs(np.asarray(mean) - ref_mean)))
    assert em <= mean_atol, f"posterior mean: max abs err {em:.3e} > {mean_atol:.3e} (1e-8 of max |mean|)"
    ev = 0.0
    if var is not None:
        ev = float(np.max(np.abs(np.asarray(var) - ref_var)))
        assert ev <= var_atol, (f"posterior variance: max abs err {ev:.3e} > {var_atol:.3e} "
                                f"(1e-8 of max |var|)")
    return em / max(mean_atol, 1e-300), ev / max(var_atol, 1e-300)


def prior_variance(wl) -> float:
    """k(x,x) of a workload's stationary prior: the sum of the kernel scales."""
    return float(sum(sc for sc, _ in wl.kernel))
