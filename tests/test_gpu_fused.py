"""The fused factor-and-predict pipeline (`lpgp_potrf_predict`, round 5): in the opt-in lazy mode a conditioning only
assembles its block; the first call that needs the factor enqueues the factorisation -- and a `predict` lets the forward
substitution of the cross-covariance ride INSIDE it (potrf.hip, `ride_panel`).  Same numbers as the two pipelines of the
default mode (1e-12), same numbers as the oracle (the one criterion of tests/conftest.py), for every shape of chain:
single block, appended blocks (old panels), ragged sizes, multi-panel factors, wide and narrow right-hand sides.
Reference sequence mirrored: `u.mean(x)`, `u.std(x)` (experiments/0001_poisson_dirichlet_2d.ipynb cell 22;
`_conditional.py:193-197,223-231`).
"""
import numpy as np
import pytest

from conftest import assert_posterior_close
from oracle import covfuncs as ocf
from oracle import gp as ogp
from oracle import workloads as owl

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lp():
    import linpde_gp_amd
    return linpde_gp_amd


@pytest.fixture
def lazy(lp):
    saved = (lp.config.lazy_factorization, lp.config.variance_with_mean)
    lp.config.lazy_factorization = True
    yield lp
    lp.config.lazy_factorization, lp.config.variance_with_mean = saved


def _both_modes(lp, wl):
    from linpde_gp_amd import problems
    lp.config.lazy_factorization = False
    u0, m0, v0 = problems.condition_and_predict(wl)
    lp.config.lazy_factorization = True
    u1, m1, v1 = problems.condition_and_predict(wl)
    return (u0, m0, v0), (u1, m1, v1)


@pytest.mark.parametrize("name,make", [
    ("poisson2d_24", lambda P: P.poisson_2d(n_side=24, n_bdry=24, m_side=12)),            # 5 blocks, 6 tiles, one appended panel
    ("poisson2d_40_ragged", lambda P: P.poisson_2d(n_side=40, n_bdry=37, m_side=17)),      # ragged blocks, 4 old tiles + 13 new
    ("poisson1d_1500", lambda P: P.poisson_1d(1500, m=300)),                                # 13 tiles: four panels, narrow RHS
    ("poisson1d_c1", lambda P: P.poisson_1d(512, n_bdry_repeats=16, noise_var=1e-4, m=256)),
    ("heat_small", lambda P: P.heat_1d(nt=48, nx=24, m_side=20)),                           # mixed blocks, a block AFTER the big one
    ("scattered_2000", lambda P: P.scattered_2d(n=2000, m=700)),                            # ONE block: no append phase at all
    ("heat_reference", lambda P: P.heat_reference()),
])
def test_fused_pipeline_equals_two_pipelines_and_oracle(lazy, name, make):
    from linpde_gp_amd import problems
    lp = lazy
    wl = make(problems)
    (u0, m0, v0), (u1, m1, v1) = _both_modes(lp, wl)
    assert u1._state.deferred is False and u1._state.pending is False
    sm, sv = np.max(np.abs(m0)), np.max(np.abs(v0))
    assert np.max(np.abs(m1 - m0)) <= 1e-12 * sm, np.max(np.abs(m1 - m0)) / sm
    assert np.max(np.abs(v1 - v0)) <= 1e-12 * sv + 1e-13 * 1.0, np.max(np.abs(v1 - v0)) / sv
    ref = owl.run(wl)
    assert_posterior_close(m1, v1, ref["mean"], ref["var"])
    # the factor the fused pipeline left behind IS the factor: weights, a second prediction elsewhere, the covariance
    # (the representer weights carry the condition number: two orders of elimination agree to cond x eps, not to 1e-12)
    np.testing.assert_allclose(u1.representer_weights, u0.representer_weights, rtol=0, atol=1e-7 * np.max(np.abs(u0.representer_weights)))
    xs = wl.Xtest[:5]
    np.testing.assert_allclose(u1.cov.matrix(xs), u0.cov.matrix(xs), rtol=0, atol=1e-12 * max(sv, 1e-3))
    m2, v2 = u1.predict(wl.Xtest[::3])            # (another number of right-hand sides: other kernels, other blocking -> 1e-10)
    np.testing.assert_allclose(m2, m0[::3], rtol=0, atol=1e-10 * sm)
    np.testing.assert_allclose(v2, v0[::3], rtol=0, atol=1e-10 * sv + 1e-13)


def test_reference_sequence_mean_then_std(lazy):
    """`u.mean(x)` then `u.std(x)` (notebook 0001 cell 22) == `u.predict(x)` to 1e-12, in the default mode, in the lazy mode,
    and in the lazy mode with `variance_with_mean` (the mean call takes the fused pipeline and keeps the variance)."""
    from linpde_gp_amd import problems
    lp = lazy
    wl = problems.poisson_2d(n_side=32, n_bdry=32, m_side=16)
    lp.config.lazy_factorization = False
    u, mean, var = problems.condition_and_predict(wl)
    sd = np.sqrt(np.maximum(var, 0.0))
    for lazy_mode, vwm in ((False, False), (False, True), (True, False), (True, True)):
        lp.config.lazy_factorization, lp.config.variance_with_mean = lazy_mode, vwm
        prior = problems.build_prior(wl)
        w = prior
        for o in wl.observations:
            X, Y = o.X_as_given()
            b = None if o.noise_var is None else lp.randvars.Normal(np.zeros(Y.shape), np.full(o.X.shape[0], o.noise_var))
            w = w.condition_on_observations(Y, X=X, L=problems.operator_of(o.op, wl.d), b=b)
        assert w._state.deferred is lazy_mode
        m = w.mean(wl.Xtest)
        assert w._state.deferred is False
        assert (w._pred_cache[3] is not None) is vwm
        s = w.std(wl.Xtest)
        assert np.max(np.abs(m - mean)) <= 1e-12 * np.max(np.abs(mean)), (lazy_mode, vwm)
        assert np.max(np.abs(s - sd)) <= 1e-12 * np.max(sd) + 1e-14, (lazy_mode, vwm)
        # the cached prediction is keyed by the VALUES of the points
        x2 = wl.Xtest.copy()
        x2[0] += 0.01
        m2 = w.mean(x2)
        assert abs(m2[0] - m[0]) > 0 and np.max(np.abs(m2[1:] - m[1:])) <= 1e-12 * np.max(np.abs(mean))


def test_deferred_block_is_factored_by_whoever_needs_it_first(lazy):
    """Lazy mode: every way of touching the factor before a `predict` enqueues the deferred factorisation first."""
    lp = lazy
    cf = lp.randprocs.covfuncs
    okern = [(1.0, [("expquad", 0.7)])]
    ident = ocf.identity(1)
    rng = np.random.default_rng(21)
    X1, Y1 = rng.uniform(-1, 1, (150, 1)), rng.normal(size=150)
    X2, Y2 = rng.uniform(-1, 1, (140, 1)), rng.normal(size=140)
    Xt = np.linspace(-1, 1, 9)[:, None]
    post = ogp.condition(okern, [ogp.ObsBlock(X1, ident, Y1, 0.0, 1e-2), ogp.ObsBlock(X2, ident, Y2, 0.0, 1e-2)])
    prior = lp.GaussianProcess(lp.functions.Zero((1,)), cf.ExpQuad((1,), lengthscales=0.7))

    def chain():
        u1 = prior.condition_on_observations(Y1, X1, b=lp.randvars.Normal(np.zeros(150), 1e-2 * np.eye(150)))
        u2 = u1.condition_on_observations(Y2, X2, b=lp.randvars.Normal(np.zeros(140), 1e-2 * np.eye(140)))
        assert u2._state.deferred
        return u1, u2

    for first_use in ("weights", "mean", "cov", "cholesky", "older_object", "predict"):
        u1, u2 = chain()
        if first_use == "weights":
            np.testing.assert_allclose(u2.representer_weights, post.weights, rtol=1e-7, atol=1e-9)
        elif first_use == "mean":
            np.testing.assert_allclose(u2.mean(Xt), post.mean(Xt), rtol=0, atol=1e-8)
        elif first_use == "cov":
            np.testing.assert_allclose(u2.cov.matrix(Xt), post.cov(Xt), rtol=0, atol=1e-8)
        elif first_use == "cholesky":
            Lf = u2.gram.cholesky()
            np.testing.assert_allclose(Lf @ Lf.T, post.G, rtol=0, atol=1e-10)
        elif first_use == "older_object":
            p1 = ogp.condition(okern, [ogp.ObsBlock(X1, ident, Y1, 0.0, 1e-2)])
            m1, v1 = u1.predict(Xt)                 # a view on the leading block: the newest block is factored behind it
            np.testing.assert_allclose(m1, p1.mean(Xt), rtol=0, atol=1e-8)
        assert (u2._state.deferred is False) or first_use == "predict"
        m, v = u2.predict(Xt)
        np.testing.assert_allclose(m, post.mean(Xt), rtol=0, atol=1e-8)
        np.testing.assert_allclose(v, post.var(Xt), rtol=0, atol=1e-9)


def test_fused_pipeline_on_a_matrix_that_is_not_positive_definite(lazy):
    """The prediction that rode inside a failing factorisation is discarded; the object raises; its parent stays exact."""
    lp = lazy
    cf = lp.randprocs.covfuncs
    okern = [(1.0, [("expquad", 1.0)])]
    ident = ocf.identity(1)
    prior = lp.GaussianProcess(lp.functions.Zero((1,)), cf.ExpQuad((1,), lengthscales=1.0))
    rng = np.random.default_rng(3)
    X1, Y1 = rng.uniform(-1, 1, (300, 1)), rng.normal(size=300)
    u1 = prior.condition_on_observations(Y1, X1, b=lp.randvars.Normal(np.zeros(300), 1e-2 * np.eye(300)))
    Xbad = np.concatenate([np.array([[0.2], [0.2]]), rng.uniform(-1, 1, (200, 1))])
    u2 = u1.condition_on_observations(np.zeros(202), Xbad, b=lp.randvars.Normal(np.zeros(202), -1e-3 * np.eye(202)))
    Xt = np.linspace(-1, 1, 7)[:, None]
    with pytest.raises(np.linalg.LinAlgError):
        u2.predict(Xt)
    with pytest.raises(np.linalg.LinAlgError):
        u2.predict(Xt)
    post1 = ogp.condition(okern, [ogp.ObsBlock(X1, ident, Y1, 0.0, 1e-2)])
    m1, v1 = u1.predict(Xt)
    np.testing.assert_allclose(m1, post1.mean(Xt), rtol=0, atol=1e-8 * np.max(np.abs(post1.mean(Xt))))
    np.testing.assert_allclose(v1, post1.var(Xt), rtol=0, atol=1e-9)


@pytest.mark.parametrize("stream,gate", [(0, 100), (1, 100), (2, 100), (3, 100), (1 + 8 * 4, 100), (4 + 8 * 1, 50), (1 + 8 * 2, 0), (1 + 8 * 7, 30)])
def test_ride_stream_variants_agree(lazy, stream, gate):
    """The substitution's steps on each of the candidate streams (masked outer stream, unmasked, narrow, the panel stream
    itself), as one sequence or as two halves of the right-hand side on two streams, released at once or held back behind
    the gate: same values."""
    from linpde_gp_amd import _engine, problems
    lp = lazy
    ctx = _engine.default_context()
    wl = problems.poisson_2d(n_side=40, n_bdry=40, m_side=24)
    lp.config.lazy_factorization = False
    _, m0, v0 = problems.condition_and_predict(wl)
    lp.config.lazy_factorization = True
    saved = ctx.get_option("ride_stream"), ctx.get_option("ride_gate_pct")
    try:
        ctx.set_option("ride_stream", stream)
        ctx.set_option("ride_gate_pct", gate)
        _, m1, v1 = problems.condition_and_predict(wl)
    finally:
        ctx.set_option("ride_stream", saved[0])
        ctx.set_option("ride_gate_pct", saved[1])
    assert np.max(np.abs(m1 - m0)) <= 1e-12 * np.max(np.abs(m0)) and np.max(np.abs(v1 - v0)) <= 1e-12 * np.max(np.abs(v0)) + 1e-13


@pytest.mark.parametrize("make", [lambda P: P.poisson_2d(n_side=40, n_bdry=40, m_side=24),          # 4 old tiles + 13: panels on the block grid
                                  lambda P: P.poisson_1d(3000, m=1100),                              # 1 old tile + 24: panels shifted by one tile
                                  lambda P: P.heat_1d(nt=64, nx=32, m_side=36)])                     # blocks of 1, 2, 2, 16, 2 tiles
@pytest.mark.parametrize("halves", [False, True])
def test_two_level_ride(lazy, make, halves):
    """The two-level form of the riding substitution (outer blocks, here of 1 024 rows instead of 4 096): steps that do not end on
    the block grid (appended blocks shift the panel grid) are covered by the three-row margin."""
    from linpde_gp_amd import _engine, problems
    lp = lazy
    ctx = _engine.default_context()
    wl = make(problems)
    lp.config.lazy_factorization = False
    _, m0, v0 = problems.condition_and_predict(wl)
    lp.config.lazy_factorization = True
    keys = ("ride_outer_rows", "ride_outer_min_tiles", "ride_stream", "ride_same_stream_max_tiles")
    saved = {k: ctx.get_option(k) for k in keys}
    try:
        ctx.set_option("ride_outer_rows", 1024)
        ctx.set_option("ride_outer_min_tiles", 1)
        ctx.set_option("ride_same_stream_max_tiles", 0)
        ctx.set_option("ride_stream", 1 + 8 * 4 if halves else 1 + 8 * 7)
        _, m1, v1 = problems.condition_and_predict(wl)
    finally:
        for k, v in saved.items():
            ctx.set_option(k, v)
    assert np.max(np.abs(m1 - m0)) <= 1e-11 * np.max(np.abs(m0)) and np.max(np.abs(v1 - v0)) <= 1e-11 * np.max(np.abs(v0)) + 1e-13


def test_a_large_block_in_the_middle_of_a_chain_stays_deferred(lazy):
    """Lazy mode: a block of at least `config.defer_min_rows` rows that is followed by further conditionings is factored TOGETHER with
    them at the first use, the prediction riding inside all of it (c5's shape: collocation block, then interior values)."""
    from linpde_gp_amd import problems
    lp = lazy
    saved = lp.config.defer_min_rows
    wl = problems.heat_1d(nt=48, nx=24, m_side=20)          # blocks of 64, 256, 256, 1 152, 256 rows
    try:
        lp.config.lazy_factorization = False
        _, m0, v0 = problems.condition_and_predict(wl)
        lp.config.lazy_factorization = True
        for thr, expect_joint in ((1000, True), (10**9, False)):
            lp.config.defer_min_rows = thr
            prior = problems.build_prior(wl)
            u = prior
            seen = []
            for o in wl.observations:
                X, Y = o.X_as_given()
                b = None if o.noise_var is None else lp.randvars.Normal(np.zeros(Y.shape), np.full(o.X.shape[0], o.noise_var))
                u = u.condition_on_observations(Y, X=X, L=problems.operator_of(o.op, wl.d), b=b)
                seen.append(u._state.deferred_rows)
            assert seen[3] == 1152 and seen[4] == (1152 + 256 if expect_joint else 256), seen
            m1, v1 = u.predict(wl.Xtest)
            assert np.max(np.abs(m1 - m0)) <= 1e-11 * np.max(np.abs(m0)) and np.max(np.abs(v1 - v0)) <= 1e-11 * np.max(np.abs(v0)) + 1e-13
    finally:
        lp.config.defer_min_rows = saved


def test_batched_block_rows_are_bit_identical():
    """Round 5: the blocks of a block row that share a descriptor are assembled by ONE launch (a table of point sets in the kernel
    arguments; `lpgp_mat_condition`, `lpgp_cross_assemble_row`).  Same code per workgroup: the factor and the cross-covariance
    are bit-identical to the one-launch-per-block path (`asm_batch = 0`)."""
    import linpde_gp_amd as lp
    from linpde_gp_amd import _engine, problems
    ctx = _engine.default_context()
    wl = problems.poisson_2d(n_side=20, n_bdry=37, m_side=9)            # four ragged boundary blocks + a grid block
    out = {}
    saved = ctx.get_option("asm_batch")
    try:
        for batch in (1, 0):
            ctx.set_option("asm_batch", batch)
            u, m, v = problems.condition_and_predict(wl)
            out[batch] = (u.gram.cholesky(), m, v, u._cross(_engine.Points(ctx, wl.Xtest)).to_host())
            del u
    finally:
        ctx.set_option("asm_batch", saved)
    for a, b in zip(out[1], out[0]):
        np.testing.assert_array_equal(a, b)
    # ... and nine value blocks in one chain: a row of more blocks than one job table holds (eight)
    cf = lp.randprocs.covfuncs
    prior = lp.GaussianProcess(lp.functions.Zero((1,)), cf.Matern((1,), nu=2.5, lengthscales=0.4))
    rng = np.random.default_rng(8)
    res = {}
    try:
        for batch in (1, 0):
            ctx.set_option("asm_batch", batch)
            r2 = np.random.default_rng(8)
            u = prior
            for k in range(17):                  # (17 ragged blocks: also more padding tails than one launch of rhs_pad_kernel takes)
                X = r2.uniform(-1, 1, (17 + 11 * k, 1))
                # noise: sigma^2 I or a vector -- the block's identity tail and its noise are one launch (finish_block_kernel)
                b = [lp.randvars.Normal(np.zeros(X.shape[0]), 1e-3 * np.eye(X.shape[0])),
                     lp.randvars.Normal(np.zeros(X.shape[0]), np.diag(1e-3 * (1.0 + r2.uniform(0, 1, X.shape[0]))))][k % 2]
                u = u.condition_on_observations(np.sin(3 * X[:, 0]), X, b=b)
            res[batch] = u.predict(np.linspace(-1, 1, 50)[:, None])
            del u
    finally:
        ctx.set_option("asm_batch", saved)
    np.testing.assert_array_equal(res[1][0], res[0][0])
    np.testing.assert_array_equal(res[1][1], res[0][1])
    assert prior._rows_seen == sum(-(-(17 + 11 * k) // 128) * 128 for k in range(17))        # (the next chain from this prior starts with room for it)


@pytest.mark.parametrize("make", [lambda P: P.poisson_1d(512, n_bdry_repeats=16, noise_var=1e-4, m=256),        # one resident panel behind an old one
                                  lambda P: P.poisson_2d(n_side=32, m_side=16),                                # two resident panels
                                  lambda P: P.poisson_2d(n_side=40, n_bdry=40, m_side=24),                     # panels off the block grid, 640 columns
                                  lambda P: P.heat_reference()])                                               # 2 560 columns: 80 waiting workgroups
def test_substitution_follows_the_resident_chain_through_its_flags(lazy, make):
    """Round 5: for a panel the resident chain factors, the substitution's panel step does not wait for the chain kernel to end: it
    follows the factor workgroup through the chain's flags (`panel_chain_v_kernel`, option `ride_vchain_max_wgs`).  Same products in
    the same order as the panel step it replaces: the prediction agrees with the one behind the event (option 0) to rounding, and
    with the oracle; the factor is untouched by it."""
    import linpde_gp_amd as lp
    from linpde_gp_amd import _engine, problems
    ctx = _engine.default_context()
    wl = make(problems)
    saved = ctx.get_option("ride_vchain_max_wgs")
    out = {}
    try:
        for wgs in (96, 0):
            ctx.set_option("ride_vchain_max_wgs", wgs)
            u, m, v = problems.condition_and_predict(wl)
            out[wgs] = (m, v, u.gram.cholesky())
            del u
    finally:
        ctx.set_option("ride_vchain_max_wgs", saved)
    np.testing.assert_array_equal(out[96][2], out[0][2])
    sm, sv = np.max(np.abs(out[0][0])), np.max(np.abs(out[0][1]))
    assert np.max(np.abs(out[96][0] - out[0][0])) <= 1e-12 * sm and np.max(np.abs(out[96][1] - out[0][1])) <= 1e-12 * sv + 1e-15
    ref = owl.run(wl)
    assert_posterior_close(out[96][0], out[96][1], ref["mean"], ref["var"])


@pytest.mark.parametrize("make", [lambda P: P.poisson_2d(n_side=32, m_side=16), lambda P: P.poisson_2d(n_side=40, n_bdry=37, m_side=17),
                                  lambda P: P.heat_reference(), lambda P: P.heat_1d(nt=48, nx=24, m_side=20)])
def test_small_grids_take_the_per_entry_kernel_and_agree_with_the_kronecker_path(lazy, make):
    """Round 5: grids below `config.grid_assembly_min_points` (12 000) points are assembled entry by entry -- eight launches of 1-D
    factor matrices and a latency-bound expansion cost a 32 x 32 grid 72 us against 8 -- and the Kronecker path stays for the large
    ones (c3 on).  Both evaluate the same product form: same Gram matrix to rounding, same posterior, both within the oracle's bar."""
    from linpde_gp_amd import config, problems
    wl = make(problems)
    saved = config.grid_assembly_min_points
    if saved <= 0:
        pytest.skip("LPGP_GRID_MIN_POINTS=0 in the environment: every grid takes the Kronecker path")
    out = {}
    try:
        for thr in (saved, 0):
            config.grid_assembly_min_points = thr
            u, m, v = problems.condition_and_predict(wl)
            out[thr] = (m, v, u.gram.cholesky())
            del u
    finally:
        config.grid_assembly_min_points = saved
    L1, L0 = out[saved][2], out[0][2]
    G1, G0 = L1 @ L1.T, L0 @ L0.T
    assert np.max(np.abs(G1 - G0)) <= 1e-12 * np.max(np.abs(G0))
    ref = owl.run(wl)
    for thr in out:
        assert_posterior_close(out[thr][0], out[thr][1], ref["mean"], ref["var"])


def test_misuse_of_the_conditioning_entry_point_rolls_the_block_back():
    """`lpgp_mat_condition` with a row entry whose point set does not match the block it names (a C-API misuse the host package
    never commits): the call fails AND the block it had declared is dropped again, on the batched row path like on the
    per-entry one -- the matrix is as before and the same conditioning with the right points then succeeds."""
    import linpde_gp_amd as lp
    from linpde_gp_amd import _engine, _lib
    from linpde_gp_amd.randprocs._gaussian_process import _lowered
    ctx = _engine.default_context()
    cf = lp.randprocs.covfuncs
    k = cf.Matern((1,), nu=2.5, lengthscales=0.5)
    rng = np.random.default_rng(17)
    X1, X2, Xbad = rng.uniform(-1, 1, (40, 1)), rng.uniform(-1, 1, (25, 1)), rng.uniform(-1, 1, (31, 1))
    ident = {(0,): 1.0}
    kd = _lowered(k, ident, ident)
    saved = ctx.get_option("asm_batch")
    try:
        for batch in (1, 0):
            ctx.set_option("asm_batch", batch)
            mat = _engine.GramMatrix(ctx)
            P1, P2, Pbad = (_engine.Points(ctx, X) for X in (X1, X2, Xbad))
            assert mat.condition(40, P1, [(kd, None)], noise_scalar=1e-2, lazy=0) == 0
            size = (mat.n, mat.num_blocks_total, list(mat.block_sizes))
            with pytest.raises(_lib.LpgpError, match="shape mismatch|size"):
                mat.condition(25, P2, [(kd, Pbad), (kd, None)], noise_scalar=1e-2, lazy=0)        # 31 points named for a block of 40
            assert (mat.n, mat.num_blocks_total, list(mat.block_sizes)) == size
            assert mat.condition(25, P2, [(kd, P1), (kd, None)], noise_scalar=1e-2, lazy=0) == 0
            assert mat.n == 65
            G = np.asarray(k.matrix(np.vstack([X1, X2]))) + 1e-2 * np.eye(65)
            L = np.tril(mat.todense("factor"))
            assert np.max(np.abs(L @ L.T - G)) <= 1e-13 * np.max(np.abs(G))
    finally:
        ctx.set_option("asm_batch", saved)


def test_kronecker_expansion_with_16_byte_stores_is_bit_identical(kronecker_everywhere):
    """Round 5: `kron2w_kernel` (a lane owns two consecutive fast rows: 16-byte stores) against `kron2_kernel`, diagonal (lower-only)
    and cross blocks, fast extents that are multiples of 128, even but ragged, and odd (falls back to the 8-byte kernel)."""
    import linpde_gp_amd as lp
    from linpde_gp_amd import _engine
    from linpde_gp_amd.linfuncops import diffops
    ctx = _engine.default_context()
    cf = lp.randprocs.covfuncs
    prior = lp.GaussianProcess(lp.functions.Zero((2,)), 1.7 * cf.TensorProduct(cf.Matern((), nu=2.5, lengthscales=0.8), cf.Matern((), nu=3.5, lengthscales=1.1)))
    saved = ctx.get_option("kron_wide")
    try:
        for n_slow, n_fast in ((9, 128), (7, 150), (6, 67), (3, 256)):
            g0 = lp.domains.TensorProductGrid(np.linspace(-1, 1, n_slow), np.linspace(-1, 1, n_fast))
            g1 = lp.domains.TensorProductGrid(np.linspace(-0.9, 0.8, 5), np.linspace(-1, 1, 40))
            res = {}
            for wide in (1, 0):
                ctx.set_option("kron_wide", wide)
                u = prior.condition_on_observations(np.zeros(g1.shape[:-1]), X=g1, b=lp.randvars.Normal(np.zeros(g1.shape[:-1]), 1e-2 * np.ones(200)))
                u = u.condition_on_observations(np.ones(g0.shape[:-1]), X=g0, L=-1.0 * diffops.Laplacian((2,)),
                                                b=lp.randvars.Normal(np.zeros(g0.shape[:-1]), 1e-3 * np.ones(n_slow * n_fast)))
                res[wide] = u.gram.cholesky()
                del u
            np.testing.assert_array_equal(res[1], res[0])
    finally:
        ctx.set_option("kron_wide", saved)
