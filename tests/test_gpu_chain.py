"""The resident panel chain (csrc/chain.hip, round 5): the whole chain of a four-tile panel in ONE launch whose workgroups hand
over through device flags -- against the tile-by-tile launches it replaces (same arithmetic: the factors agree to rounding),
against the oracle, and in its failure path (a pivot that is not positive inside a resident panel)."""
import numpy as np
import pytest

from oracle import covfuncs as ocf
from oracle import gp as ogp

pytestmark = pytest.mark.gpu


@pytest.fixture
def ctx():
    from linpde_gp_amd import _engine
    c = _engine.default_context()
    saved = c.get_option("chain_resident_max_rows"), c.get_option("chain_resident2_max_rows"), c.get_option("chain_ahead"), c.get_option("chain_ahead_min_rows")
    yield c
    c.set_option("chain_resident_max_rows", saved[0])
    c.set_option("chain_resident2_max_rows", saved[1])
    c.set_option("chain_ahead", saved[2])
    c.set_option("chain_ahead_min_rows", saved[3])


def _posterior(lp, n, seed, noise=1e-3, ls=0.35):
    cf = lp.randprocs.covfuncs
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, (n, 2))
    Y = np.sin(3 * X[:, 0]) * np.cos(2 * X[:, 1]) + 0.01 * rng.standard_normal(n)
    prior = lp.GaussianProcess(lp.functions.Zero((2,)), 1.3**2 * cf.TensorProduct(cf.Matern((), nu=2.5, lengthscales=ls), cf.Matern((), nu=1.5, lengthscales=ls)))
    okern = [(1.69, [("matern", 2.5, ls), ("matern", 1.5, ls)])]
    return prior, okern, X, Y, lp.randvars.Normal(np.zeros(n), noise * np.eye(n))


@pytest.mark.parametrize("n", [512, 520, 1024, 1500, 2048 + 77, 4608])     # 4 tiles exactly (no rows below) ... 36 tiles: nine panels
def test_resident_chain_equals_the_tile_by_tile_chain(ctx, n):
    import linpde_gp_amd as lp
    prior, okern, X, Y, b = _posterior(lp, n, seed=n)
    Xt = np.random.default_rng(1).uniform(-1, 1, (33, 2))
    out = {}
    for mode, rows in (("resident", 64), ("tiles", -1)):
        ctx.set_option("chain_resident_max_rows", rows)
        u = prior.condition_on_observations(Y, X, b=b)
        out[mode] = (u.gram.cholesky(), u.predict(Xt), u.representer_weights)
        del u
    Lr, Lt = out["resident"][0], out["tiles"][0]
    assert np.max(np.abs(Lr - Lt)) <= 1e-12 * np.max(np.abs(Lt))
    post = ogp.condition(okern, [ogp.ObsBlock(X, ocf.identity(2), Y, 0.0, 1e-3)])
    m, v = out["resident"][1]
    assert np.max(np.abs(m - post.mean(Xt))) <= 1e-8 * np.max(np.abs(post.mean(Xt)))
    assert np.max(np.abs(v - post.var(Xt))) <= 1e-8 * np.max(np.abs(post.var(Xt)))
    np.testing.assert_allclose(Lr @ Lr.T, post.G, rtol=0, atol=1e-12 * np.max(np.abs(post.G)) * 4)


def test_resident_chain_reports_the_first_bad_pivot(ctx):
    """A Gram matrix that is not positive definite INSIDE a resident panel: the status names the leading minor (here in the third
    tile of the second panel), the conditioning raises, the parent stays usable."""
    import linpde_gp_amd as lp
    ctx.set_option("chain_resident_max_rows", 64)
    prior, okern, X, Y, _ = _posterior(lp, 1100, seed=5)
    noise = np.full(1100, 1e-3)
    noise[830] = -5.0                                  # row 830: tile 6 = panel 1, third tile
    with pytest.raises(np.linalg.LinAlgError) as e:
        prior.condition_on_observations(Y, X, b=lp.randvars.Normal(np.zeros(1100), noise))
    assert "831-th" in str(e.value) or "83" in str(e.value), str(e.value)
    u = prior.condition_on_observations(Y, X, b=lp.randvars.Normal(np.zeros(1100), np.full(1100, 1e-3)))
    assert np.all(np.isfinite(u.predict(X[:5])[0]))


def test_resident_chain_in_an_appended_block(ctx):
    """Block append: old panels first (phase A), then resident panels whose first tile column is not a multiple of four."""
    import linpde_gp_amd as lp
    ctx.set_option("chain_resident_max_rows", 64)
    prior, okern, X, Y, _ = _posterior(lp, 1300, seed=9)
    cuts = [0, 130, 390, 1300]                         # blocks of 2, 3 and 8 tiles (130 -> 256, 260 -> 384, 910 -> 1024 padded rows)
    u = prior
    blocks = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        u = u.condition_on_observations(Y[lo:hi], X[lo:hi], b=lp.randvars.Normal(np.zeros(hi - lo), 1e-3 * np.eye(hi - lo)))
        blocks.append(ogp.ObsBlock(X[lo:hi], ocf.identity(2), Y[lo:hi], 0.0, 1e-3))
    post = ogp.condition(okern, blocks)
    Xt = np.random.default_rng(2).uniform(-1, 1, (40, 2))
    m, v = u.predict(Xt)
    assert np.max(np.abs(m - post.mean(Xt))) <= 1e-8 * np.max(np.abs(post.mean(Xt)))
    assert np.max(np.abs(v - post.var(Xt))) <= 1e-8 * np.max(np.abs(post.var(Xt)))


def test_a_timed_out_hand_over_kills_the_chain_for_good(ctx):
    """a negative device status (a hand-over inside a resident kernel timed out: the panel's contents are undefined)
    was reported ONCE -- `lpgp_mat_check` cleared `unchecked` before failing, `verify` cleared `pending` before the check raised --
    and the next `predict` ran on the garbage factor as if it had been verified.  Now the library keeps the matrix marked undefined
    and every object of the chain raises from then on.  The status is forced through a test hook (a real time-out needs a
    serialising profiler)."""
    import linpde_gp_amd as lp
    import _hooks
    from linpde_gp_amd._lib import LpgpError
    prior, okern, X, Y, b = _posterior(lp, 700, seed=2)
    Xt = np.random.default_rng(1).uniform(-1, 1, (9, 2))
    lp.config.lazy_factorization = True
    try:
        u = prior.condition_on_observations(Y, X, b=b)
        u._state.flush()                                        # enqueued, status not read yet
        assert u._state.pending
        _hooks.force_status(ctx, u._state.mat, -(2**31))
        with pytest.raises(LpgpError, match="timed out"):
            u.predict(Xt)
        # the second use does not run on the undefined factor: the object is dead, the library refuses the matrix
        with pytest.raises(np.linalg.LinAlgError, match="timed out"):
            u.predict(Xt)
        with pytest.raises(np.linalg.LinAlgError):
            u.condition_on_observations(Y[:5], X[:5] + 0.01, b=lp.randvars.Normal(np.zeros(5), 1e-3 * np.eye(5)))
        with pytest.raises(LpgpError, match="undefined"):
            u._state.mat.solve_weights(Y)
        # the prior is untouched: a fresh conditioning works
        m, v = prior.condition_on_observations(Y, X, b=b).predict(Xt)
        post = ogp.condition(okern, [ogp.ObsBlock(X, ocf.identity(2), Y, 0.0, 1e-3)])
        assert np.max(np.abs(m - post.mean(Xt))) <= 1e-8 * np.max(np.abs(post.mean(Xt)))
    finally:
        lp.config.lazy_factorization = False


@pytest.mark.parametrize("n", [640, 1500, 4608, 8320])       # 1 ... 61 tile rows below the first panel (c2's shape: 65 tile rows)
def test_two_kernel_resident_chain_equals_the_tile_by_tile_chain(ctx, n):
    """Round 6: panels with MORE rows below than the one-launch chain holds (one 152-KB workgroup per CU) run as TWO launches that
    talk through the same device flags -- factor + in-block workgroups on the panel stream, the rows below as 16-row workgroups of
    68 KB, two per CU, on an idle masked stream (`panel_chain_rows_kernel`).  Every panel forced through that form (one-launch
    limit 0 rows, two-launch limit 64) against the tile-by-tile chain: factors to 1e-12, posterior against the oracle, in the
    default mode and with the substitution riding inside (lazy)."""
    import linpde_gp_amd as lp
    prior, okern, X, Y, b = _posterior(lp, n, seed=n + 1)
    Xt = np.random.default_rng(1).uniform(-1, 1, (33, 2))
    out = {}
    for mode, (r1, r2) in (("two", (0, 64)), ("tiles", (-1, 0))):
        ctx.set_option("chain_resident_max_rows", r1)
        ctx.set_option("chain_resident2_max_rows", r2)
        ctx.profile_reset(); ctx.profile_enable(True)
        u = prior.condition_on_observations(Y, X, b=b)
        pred = u.predict(Xt)
        prof = ctx.profile_get(); ctx.profile_enable(False)
        out[mode] = (u.gram.cholesky(), pred, prof["potrf_tile"]["launches"])
        del u
    assert out["tiles"][2] > 0
    if n <= 65 * 128 - 512 - 128:
        assert out["two"][2] <= 4, out["two"][2]            # (every four-tile panel went through the chain kernels; a ragged last panel may not)
    Lr, Lt = out["two"][0], out["tiles"][0]
    assert np.max(np.abs(Lr - Lt)) <= 1e-12 * np.max(np.abs(Lt))
    post = ogp.condition(okern, [ogp.ObsBlock(X, ocf.identity(2), Y, 0.0, 1e-3)])
    m, v = out["two"][1]
    assert np.max(np.abs(m - post.mean(Xt))) <= 1e-8 * np.max(np.abs(post.mean(Xt)))
    assert np.max(np.abs(v - post.var(Xt))) <= 1e-8 * np.max(np.abs(post.var(Xt)))
    # the fused pipeline over the same panels
    ctx.set_option("chain_resident_max_rows", 0)
    ctx.set_option("chain_resident2_max_rows", 64)
    lp.config.lazy_factorization = True
    try:
        m2, v2 = prior.condition_on_observations(Y, X, b=b).predict(Xt)
    finally:
        lp.config.lazy_factorization = False
    assert np.max(np.abs(m2 - m)) <= 1e-11 * np.max(np.abs(m)) and np.max(np.abs(v2 - v)) <= 1e-10 * np.max(np.abs(v))


@pytest.mark.parametrize("n", [1500, 4608, 8320])
def test_look_ahead_update_fused_into_the_next_chain(ctx, n):
    """Round 6: where two resident panels follow each other the look-ahead update between them is not a launch -- the NEXT chain
    kernel's row workgroups apply it to their own rows first (`panel_chain_kernel<4>`: sixteen more products per row workgroup,
    four workgroups for the rows of the panel's first tile, the factor workgroup waits for them).  Forced on every eligible panel
    (no minimum of rows) against the separate launch: same factor to 1e-12, fewer look-ahead launches, posterior against the oracle;
    default mode and the fused factor-and-predict pipeline."""
    import linpde_gp_amd as lp
    prior, okern, X, Y, b = _posterior(lp, n, seed=n + 7)
    Xt = np.random.default_rng(2).uniform(-1, 1, (21, 2))
    ctx.set_option("chain_resident_max_rows", 64)
    ctx.set_option("chain_ahead_min_rows", 0)
    out = {}
    for ahead in (1, 0):
        ctx.set_option("chain_ahead", ahead)
        ctx.profile_reset(); ctx.profile_enable(True)
        u = prior.condition_on_observations(Y, X, b=b)
        pred = u.predict(Xt)
        prof = ctx.profile_get(); ctx.profile_enable(False)
        out[ahead] = (u.gram.cholesky(), pred, prof["syrk_lookahead"]["launches"] + prof["gemm_small"]["launches"], np.array(u.representer_weights))
        del u
    assert out[1][2] < out[0][2], (out[1][2], out[0][2])                 # look-ahead launches disappeared
    assert np.max(np.abs(out[1][0] - out[0][0])) <= 1e-12 * np.max(np.abs(out[0][0]))
    post = ogp.condition(okern, [ogp.ObsBlock(X, ocf.identity(2), Y, 0.0, 1e-3)])
    m, v = out[1][1]
    assert np.max(np.abs(m - post.mean(Xt))) <= 1e-8 * np.max(np.abs(post.mean(Xt)))
    assert np.max(np.abs(v - post.var(Xt))) <= 1e-8 * np.max(np.abs(post.var(Xt)))
    ctx.set_option("chain_ahead", 1)
    lp.config.lazy_factorization = True
    try:
        m2, v2 = prior.condition_on_observations(Y, X, b=b).predict(Xt)
    finally:
        lp.config.lazy_factorization = False
    assert np.max(np.abs(m2 - m)) <= 1e-11 * np.max(np.abs(m)) and np.max(np.abs(v2 - v)) <= 1e-10 * np.max(np.abs(v))
