"""The resident single-vector solve (`csrc/trsv.hip`, round 6): representer weights `gram.solve(Y - Lm)` of every conditioning
(`_conditional.py:44,96-110`; `BlockMatrix2x2._solve`, `linops/_block.py:244-268`) in ONE launch per direction whose workgroups
hand the solution over block by block -- against LAPACK on the oracle's Gram matrix, against the per-tile launches of rounds 1-5
(`trsv_resident = 0`) and against the multi-right-hand-side path (`lpgp_potrs`)."""
import numpy as np
import pytest
import scipy.linalg

from oracle import covfuncs as ocf
from oracle import gp as ogp

pytestmark = pytest.mark.gpu


@pytest.fixture
def ctx():
    from linpde_gp_amd import _engine
    c = _engine.default_context()
    saved = c.get_option("trsv_resident")
    yield c
    c.set_option("trsv_resident", saved)


def _problem(lp, n, seed, noise=1e-3, ls=0.35):
    cf = lp.randprocs.covfuncs
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, (n, 2))
    Y = np.sin(3 * X[:, 0]) * np.cos(2 * X[:, 1]) + 0.01 * rng.standard_normal(n)
    prior = lp.GaussianProcess(lp.functions.Zero((2,)), 1.3**2 * cf.TensorProduct(cf.Matern((), nu=2.5, lengthscales=ls), cf.Matern((), nu=1.5, lengthscales=ls)))
    okern = [(1.69, [("matern", 2.5, ls), ("matern", 1.5, ls)])]
    return prior, okern, X, Y, lp.randvars.Normal(np.zeros(n), noise * np.eye(n))


# one tile (no hand-over at all), two, three (the stream's padding cases), a ragged last tile, 13 and 37 tile rows
@pytest.mark.parametrize("n", [100, 128, 256, 300, 384, 1000, 1600, 4700])
def test_resident_solve_against_lapack_and_the_per_tile_path(ctx, n):
    import linpde_gp_amd as lp
    assert ctx.get_option("trsv_resident") == 1
    prior, okern, X, Y, b = _problem(lp, n, seed=n)
    u = prior.condition_on_observations(Y, X, b=b)
    w_res = np.array(u.representer_weights)
    ctx.set_option("trsv_resident", 0)
    u2 = prior.condition_on_observations(Y, X, b=b)
    w_tile = np.array(u2.representer_weights)
    post = ogp.condition(okern, [ogp.ObsBlock(X, ocf.identity(2), Y, 0.0, 1e-3)])
    w_ref = scipy.linalg.cho_solve(scipy.linalg.cho_factor(post.G, lower=True), Y)
    scale = np.max(np.abs(w_ref))
    # residual of the system itself (cond ~1e5-1e6: LAPACK-level backward error) ...
    assert np.max(np.abs(post.G @ w_res - Y)) <= 1e-11 * np.max(np.abs(post.G)) * scale
    # ... and the weights
    assert np.max(np.abs(w_res - w_ref)) <= 1e-9 * scale
    assert np.max(np.abs(w_res - w_tile)) <= 1e-11 * scale
    # the multi-right-hand-side path of the same factor
    w_potrs = u.gram.solve(Y[:, None])[:, 0]
    assert np.max(np.abs(w_res - w_potrs)) <= 1e-10 * scale


def test_resident_solve_over_appended_blocks(ctx):
    """A chain of conditionings (ragged blocks, identity-padded tails): the vector passes padded rows, whose diagonal tiles are
    part identity, and the weights of EVERY posterior of the chain are those of its own leading blocks."""
    import linpde_gp_amd as lp
    prior, okern, X, Y, _ = _problem(lp, 900, seed=3)
    cuts = [0, 130, 131, 500, 900]
    u = prior
    blocks = []
    for a, c in zip(cuts[:-1], cuts[1:]):
        u = u.condition_on_observations(Y[a:c], X[a:c], b=lp.randvars.Normal(np.zeros(c - a), 1e-3 * np.eye(c - a)))
        blocks.append(ogp.ObsBlock(X[a:c], ocf.identity(2), Y[a:c], 0.0, 1e-3))
        post = ogp.condition(okern, list(blocks))
        w_ref = scipy.linalg.cho_solve(scipy.linalg.cho_factor(post.G, lower=True), Y[:c])
        assert np.max(np.abs(np.array(u.representer_weights) - w_ref)) <= 1e-9 * np.max(np.abs(w_ref))


def test_resident_solve_is_one_launch_per_direction(ctx):
    """Profiling slot `trsm_gemm` counts the solve's launches: two for the resident form whatever the size, 2 T for the per-tile form."""
    import linpde_gp_amd as lp
    prior, okern, X, Y, b = _problem(lp, 1600, seed=11)
    counts = {}
    for mode in (1, 0):
        ctx.set_option("trsv_resident", mode)
        u = prior.condition_on_observations(Y, X, b=b)
        u.predict(X[:8])                                     # (factor in place, nothing pending)
        ctx.profile_reset(); ctx.profile_enable(True)
        _ = u.representer_weights
        prof = ctx.profile_get(); ctx.profile_enable(False)
        counts[mode] = prof["trsm_gemm"]["launches"]
    assert counts[1] == 2 and counts[0] == 0, counts       # (the per-tile kernels were never bracketed: slot stays empty)
