"""c4 (BASELINE.json config 4: 2-D Poisson-Dirichlet, 256 x 256 collocation + 4 x 256 boundary observations, N_tot = 66 560:
a 35.4 GB Gram matrix) on one GPU AT FULL SIZE against the CPU oracle, mean and variance on the 128 x 128 prediction grid of SURVEY's c4 (M = 16 384) with
the one criterion of tests/conftest.py (1e-8 of max |mean| / max |var|).

Default: against the oracle's COMMITTED output for exactly these inputs (tests/golden/c4_posterior.npz, generator
tests/golden/make_c4_golden.py).  LPGP_C4_LIVE_ORACLE=1 also runs the oracle live, as rounds 1-4 did:

The oracle (`oracle.workloads.run_in_place`: chunked NumPy assembly into ONE column-major array, LAPACK dpotrf in place,
dtrtrs overwriting K^T) needs ~60 GB of host memory and ~9.8e13 flop of LAPACK: about five minutes on the GPU box's 256
host cores -- hence a file of its own whose name sorts LAST in the suite.  On a host with fewer than 128 cores or less
than 160 GB of free memory the same test runs at 192 x 192 (N_tot = 37 632) and says so.
(SURVEY.md §8d sizes; reference test mirrored: tests/linpde_gp/randprocs/test_posterior_gp.py:152-178.)
"""
import os
import sys

import numpy as np
import pytest

from conftest import assert_posterior_close
from oracle import workloads as owl

pytestmark = pytest.mark.gpu


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c4_posterior.npz")


def test_c4_poisson2d_256_full_size_vs_oracle_fixture():
    """c4 at FULL size (N_tot = 66 560, M = 16 384) against the committed oracle posterior `tests/golden/c4_posterior.npz`
    (`tests/golden/make_c4_golden.py`: `oracle.workloads.run_in_place` run once in the build container; the inputs are
    regenerated here and matched by checksum).  Round 4 ran that oracle live on the GPU box: 241 s of a 635-s suite."""
    import linpde_gp_amd as lp
    from linpde_gp_amd import problems
    sys.path.insert(0, os.path.dirname(GOLDEN))
    from make_c4_golden import workload_digest
    gold = np.load(GOLDEN)
    wl = problems.poisson_2d(256, m_side=128)
    assert wl.n_total == 66560 == int(gold["n_total"]) and wl.Xtest.shape[0] == 16384 == int(gold["m"])
    assert workload_digest(wl) == str(gold["digest"]), "the fixture was generated for other inputs"
    lp.config.gram_capacity_hint = wl.n_total
    try:
        u, mean, var = problems.condition_and_predict(wl)
    finally:
        lp.config.gram_capacity_hint = 0
    del u
    rm, rv = assert_posterior_close(mean, var, gold["mean"], gold["var"])
    # ... and the opt-in fused factor-and-predict pipeline (lazy mode) against the same fixture
    lp.config.gram_capacity_hint, lp.config.lazy_factorization = wl.n_total, True
    try:
        u, mean_f, var_f = problems.condition_and_predict(wl)
    finally:
        lp.config.gram_capacity_hint, lp.config.lazy_factorization = 0, False
    del u
    fm, fv = assert_posterior_close(mean_f, var_f, gold["mean"], gold["var"])
    msg = (f"{wl.name}: N_tot={wl.n_total} M={wl.Xtest.shape[0]} mean err {rm:.2e} x tol, var err {rv:.2e} x tol (fused pipeline: {fm:.2e}, {fv:.2e}) "
           f"vs committed oracle fixture ({gold['provenance']})")
    print(msg)
    import warnings
    warnings.warn("c4 full-size parity ran as: " + msg)
    assert abs(mean.max() - 0.5894) < 1e-2


@pytest.mark.skipif(not os.environ.get("LPGP_C4_LIVE_ORACLE"), reason="the live c4 oracle (~4 min of a 256-core host) runs with LPGP_C4_LIVE_ORACLE=1; the default suite compares against its committed output")
@pytest.mark.slow
def test_c4_poisson2d_256_full_size_vs_oracle():
    import psutil
    import linpde_gp_amd as lp
    from linpde_gp_amd import problems
    cores, avail = os.cpu_count() or 1, psutil.virtual_memory().available
    full = cores >= 128 and avail >= 160e9
    # M = 16 384 prediction points (128 x 128) -- at full size when the host also has the memory
    # for the oracle's 66 560 x 16 384 right-hand side (8.7 GB more); 64 x 64 otherwise
    m_side = 128 if (full and avail >= 200e9) else 64
    wl = problems.poisson_2d(256 if full else 192, m_side=m_side)
    assert wl.n_total == (66560 if full else 37632) and wl.Xtest.shape[0] == m_side * m_side
    if cores >= 128 and psutil.virtual_memory().total >= 400e9:
        assert full, f"a {cores}-core host with {psutil.virtual_memory().total / 1e9:.0f} GB must run c4 at full size ({avail / 1e9:.0f} GB free)"
    workers = max(1, min(16, cores // 8))
    need = owl.host_memory_needed(wl, 512, workers)
    if avail < 1.2 * need:
        pytest.skip(f"the oracle at N_tot = {wl.n_total} needs {need / 1e9:.0f} GB of host memory, {avail / 1e9:.0f} GB free")
    lp.config.gram_capacity_hint = wl.n_total
    try:
        u, mean, var = problems.condition_and_predict(wl)
    finally:
        lp.config.gram_capacity_hint = 0
    del u
    ref = owl.run_in_place(wl, chunk=512, workers=workers)
    rm, rv = assert_posterior_close(mean, var, ref["mean"], ref["var"])
    msg = (f"{wl.name}{'' if full else ' (REDUCED: host too small for 256 x 256)'}: N_tot={wl.n_total} M={wl.Xtest.shape[0]} "
           f"mean err {rm:.2e} x tol, var err {rv:.2e} x tol; oracle {ref['seconds']} ({workers} assembly threads, {cores} cores)")
    print(msg)
    import warnings
    warnings.warn("c4 full-size parity ran as: " + msg)        # (a warning reaches the driver's `pytest -q` log: which size ran)
    assert abs(mean.max() - 0.5894) < 1e-2
