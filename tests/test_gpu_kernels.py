"""Raw-kernel parity on the GPU, through the C ABI (ctypes): MFMA GEMM/SYRK variants,
the 128x128 tile Cholesky + inverse, and the roofline probes."""
import numpy as np
import pytest
import scipy.linalg

import _hooks      # tests/_hooks.py: liblpgp_testhooks.so (include/lpgp_test.h)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from linpde_gp_amd import _engine
    return _engine.default_context()


def test_options_round_trip(ctx):
    """`lpgp_set_option` / `lpgp_get_option`: the knobs bench.py and the tests flip, and what `roofline_kernel_symbol` (the
    name bench.py reports for profiling slot "syrk_trailing") follows."""
    from linpde_gp_amd import _lib
    assert ctx.get_option("nb") == 512 and ctx.get_option("gemm3") == 768 and ctx.get_option("gemm3_fact") == 0
    assert ctx.roofline_kernel_symbol() == "gemm_f64_kernel<false, false, 1>"
    try:
        ctx.set_option("gemm3_fact", 1)
        assert ctx.get_option("gemm3_fact") == 1 and ctx.roofline_kernel_symbol() == "gemm3_f64_kernel<false, 1>"
        ctx.set_option("gemm3", 0)
        assert ctx.roofline_kernel_symbol() == "gemm_f64_kernel<false, false, 1>"
    finally:
        ctx.set_option("gemm3", -1)
        ctx.set_option("gemm3_fact", 0)
    assert ctx.get_option("gemm3") == 768
    with pytest.raises(_lib.LpgpError):
        ctx.get_option("no_such_option")
    with pytest.raises(_lib.LpgpError):
        ctx.set_option("nb", 100)                     # not a multiple of 128


def test_probes(ctx):
    tf = _hooks.probe_mfma_f64(ctx)
    gb = _hooks.probe_hbm_write(ctx, 1 << 30)
    print(f"\n[probe] fp64 MFMA issue loop: {tf:.1f} TFLOP/s ; streaming write: {gb:.0f} GB/s ; {ctx.device_info()}")
    assert tf > 10.0 and gb > 500.0


@pytest.fixture(params=["tile128", "tile64", "tile128x3"])
def gemm_kernel(ctx, request):
    """All GEMM kernels: launches with at most `small_tiles_max` 128x128 tiles run on the
    64x64-tile kernel (default 256); 0 forces the 128x128-tile kernel; "tile128x3": the NT form on the
    three-workgroups-per-CU kernel (`gemm3_f64_kernel`, option `gemm3`), whatever the default is."""
    ctx.set_option("small_tiles_max", 1 << 20 if request.param == "tile64" else 0)
    ctx.set_option("gemm3", {"tile128": 0, "tile64": 0, "tile128x3": 1}[request.param])
    ctx.set_option("gemm3_fact", 1 if request.param == "tile128x3" else 0)     # (symmetric updates on the variant as well)
    yield request.param
    ctx.set_option("small_tiles_max", 256)
    ctx.set_option("gemm3", -1)
    ctx.set_option("gemm3_fact", 0)


@pytest.mark.parametrize("k", [16, 80, 512])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_variants(ctx, gemm_kernel, ta, tb, k):
    from linpde_gp_amd import _engine
    rng = np.random.default_rng(100 + 2 * ta + tb)
    m, n = 384, 256
    Am = rng.standard_normal((m, k))      # logical A (m x k)
    Bm = rng.standard_normal((k, n))      # logical B (k x n)
    C0 = rng.standard_normal((m, n))
    A_store = Am.T if ta else Am          # ta: k fastest => stored (k x m) column-major
    B_store = Bm if tb else Bm.T          # tb=0: stored (n x k) column-major (n fastest)
    out, _ = _hooks.test_gemm(ctx, ta, tb, 0, -1.5, A_store, B_store, 0.5, C0, k)
    ref = 0.5 * C0 - 1.5 * Am @ Bm
    np.testing.assert_allclose(out, ref, rtol=1e-13, atol=1e-12)
    out0, _ = _hooks.test_gemm(ctx, ta, tb, 0, 1.0, A_store, B_store, 0.0, np.full_like(C0, np.nan), k)
    np.testing.assert_allclose(out0, Am @ Bm, rtol=1e-13, atol=1e-12)


def test_syrk_lower_only(ctx, gemm_kernel):
    from linpde_gp_amd import _engine
    rng = np.random.default_rng(7)
    n, k = 640, 512
    P = rng.standard_normal((n, k))
    C0 = rng.standard_normal((n, n))
    out, _ = _hooks.test_gemm(ctx, 0, 0, 1, -1.0, P, P, 1.0, C0, k)
    ref = C0 - P @ P.T
    edge = 64 if gemm_kernel == "tile64" else 128
    tile = np.arange(n) // edge
    lower_tiles = tile[:, None] >= tile[None, :]
    np.testing.assert_allclose(out[lower_tiles], ref[lower_tiles], rtol=1e-12, atol=1e-10)
    # tiles strictly above the diagonal are untouched
    np.testing.assert_array_equal(out[~lower_tiles], C0[~lower_tiles])


def test_gemm_throughput(ctx):
    from linpde_gp_amd import _engine
    rng = np.random.default_rng(11)
    n, k = 8192, 512
    P = rng.standard_normal((n, k))
    C0 = np.zeros((n, n), order="F")
    _, ms = _hooks.test_gemm(ctx, 0, 0, 1, -1.0, P, P, 1.0, C0, k, reps=5)
    flops = n * (n + 1.0) * k
    print(f"\n[syrk] n={n} k={k}: {ms:.3f} ms  -> {flops / ms / 1e9:.1f} TFLOP/s algorithmic")
    _, ms = _hooks.test_gemm(ctx, 0, 1, 0, -1.0, P, np.asfortranarray(rng.standard_normal((k, 4096))), 1.0,
                              np.zeros((n, 4096), order="F"), k, reps=5)
    print(f"[gemm nn] {n}x4096x{k}: {ms:.3f} ms -> {2.0 * n * 4096 * k / ms / 1e9:.1f} TFLOP/s")


@pytest.mark.parametrize("seed", [0, 1])
def test_potrf_tile(ctx, seed):
    from linpde_gp_amd import _engine
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((128, 160))
    A = M @ M.T + 1e-3 * np.eye(128)
    L, Linv, info = _hooks.test_potrf_tile(ctx, A)
    assert info == 0
    Lref = np.linalg.cholesky(A)
    np.testing.assert_allclose(np.tril(L), Lref, rtol=1e-10, atol=1e-11)
    assert np.all(np.triu(L, 1) == 0.0)
    np.testing.assert_allclose(Linv, scipy.linalg.solve_triangular(Lref, np.eye(128), lower=True),
                               rtol=1e-8, atol=1e-9)
    assert np.all(np.triu(Linv, 1) == 0.0)


@pytest.mark.parametrize("case", ["dense 1-D grid", "clustered 2-D"])
def test_potrf_tile_backward_error_on_ill_conditioned_tiles(ctx, case):
    """A Cholesky factor reproduces its matrix to rounding WHATEVER the condition (LAPACK: 8e-16 of max |A| on these tiles).  The
    panel step inside the tile kernel -- the product with the explicit inverse of a 16 x 16 diagonal block -- is refined once
    against the block for that: without the step these tiles (diagonal blocks of condition 1e5) came out at 5e-14 ... 7e-14
    (round 4, `scratch/tile_chol_accuracy.py`; found by a 400-seed survey of tests/test_gpu_random.py)."""
    from linpde_gp_amd import _engine
    rng = np.random.default_rng(0)
    pts, noise = ((np.linspace(-1, 1, 128)[:, None], 1e-10) if case == "dense 1-D grid"
                  else (0.05 * rng.standard_normal((128, 2)), 1e-9))
    K = np.ones((128, 128))
    for d in range(pts.shape[1]):
        r = np.sqrt(5.0) * np.abs(pts[:, None, d] - pts[None, :, d])
        K *= (1 + r + r * r / 3) * np.exp(-r)
    A = 4.0 * K + noise * np.eye(128)
    L, _, info = _hooks.test_potrf_tile(ctx, A)
    assert info == 0
    L = np.tril(L)
    back = np.abs(A - L @ L.T).max() / np.abs(A).max()
    lapack = np.linalg.cholesky(A)
    assert back <= 4e-15, f"backward error {back:.2e} (LAPACK: {np.abs(A - lapack @ lapack.T).max() / np.abs(A).max():.2e})"


def test_potrf_tile_not_pd(ctx):
    from linpde_gp_amd import _engine
    A = np.eye(128)
    A[40, 40] = -1.0
    _, _, info = _hooks.test_potrf_tile(ctx, A)
    assert info == 41


def test_matrix_free_product_matches_dense(ctx):
    """`k.linop(x0, x1) @ V` (lpgp_kernel_matvec: entries evaluated on the fly) vs the dense
    kernel matrix from the assembly kernel, ragged sizes, 1 / 3 / 6 right-hand sides, transposed
    operator of an asymmetric derivative kernel, 1-D and 3-D inputs."""
    import linpde_gp_amd as lp
    from linpde_gp_amd.linfuncops import diffops
    cf = lp.randprocs.covfuncs
    rng = np.random.default_rng(12)
    k2 = 2.0**2 * cf.TensorProduct(cf.Matern((), nu=2.5, lengthscales=1.0), cf.Matern((), nu=2.5, lengthscales=0.7))
    D = -1.0 * diffops.Laplacian((2,))
    X0 = rng.uniform(-1, 1, size=(333, 2))
    X1 = rng.uniform(-1, 1, size=(1000, 2))
    for k in (k2, D(k2, argnum=1), D(D(k2, argnum=1), argnum=0)):
        K = k.matrix(X0, X1)
        op = k.linop(X0, X1)
        assert op.shape == (333, 1000)
        for nrhs in (1, 3, 6):
            V = rng.standard_normal((1000, nrhs))
            np.testing.assert_allclose(op @ V, K @ V, rtol=0, atol=1e-11 * np.abs(K).max() * 1000)
        v = rng.standard_normal(1000)
        np.testing.assert_allclose(op @ v, K @ v, rtol=0, atol=1e-11 * np.abs(K).max() * 1000)
        W = rng.standard_normal((333, 2))
        np.testing.assert_allclose(op.T @ W, K.T @ W, rtol=0, atol=1e-11 * np.abs(K).max() * 333)
    k1 = cf.Matern((), nu=1.5, lengthscales=0.3)
    x = rng.uniform(-1, 1, size=130)
    np.testing.assert_allclose(k1.linop(x) @ np.ones(130), k1.matrix(x) @ np.ones(130), rtol=1e-12)
    k3 = cf.TensorProduct(cf.ExpQuad((), lengthscales=0.5), cf.Matern((), nu=2.5), cf.Matern((), nu=0.5 + 1))
    Y = rng.uniform(-1, 1, size=(70, 3))
    np.testing.assert_allclose(k3.linop(Y) @ np.arange(70.0), k3.matrix(Y) @ np.arange(70.0), rtol=1e-12)


def test_device_exponential_is_correctly_rounded_to_half_an_ulp(ctx):
    """`lpgp_exp_neg` (csrc/eval_entries.h: 256-entry head + tail table of 2^(j/256), degree-4 polynomial) ON THE DEVICE: the
    Matern-1/2 kernel matrix of the origin against points s is e^{-s} with nothing else in the entry (scale 1, polynomial 1),
    compared with 40-digit arithmetic.  Both assembly kernels (specialised and generic) and the matrix-free product share the
    function; the ExpQuad factor exercises the r^2/2 argument."""
    import mpmath as mp
    import linpde_gp_amd as lp
    cf = lp.randprocs.covfuncs
    mp.mp.dps = 40
    rng = np.random.default_rng(8)
    s = np.concatenate([rng.uniform(0, 1e-3, 500), rng.uniform(0, 2, 1500), rng.uniform(0, 40, 1500), rng.uniform(0, 700, 500),
                        np.arange(0, 64) * (np.log(2) / 256), [0.0, 708.0, 744.0, 800.0, 1e10]])
    k = cf.Matern((), nu=0.5, lengthscales=1.0)
    for fast in (1, 0):
        ctx.set_option("asm_fast", fast)
        try:
            row = np.asarray(k.matrix(np.zeros(1), s))[0]
        finally:
            ctx.set_option("asm_fast", 1)
        worst = 0.0
        for si, oi in zip(s, row):
            ex = mp.exp(-mp.mpf(float(si)))
            exd = float(ex)
            if exd < 2.3e-308:
                assert 0.0 <= oi <= 2.3e-308
                continue
            worst = max(worst, float(abs(mp.mpf(float(oi)) - ex) / np.spacing(exd)))
        assert worst <= 0.53, (fast, worst)
        assert row[s == 0.0][0] == 1.0
    q = cf.ExpQuad((), lengthscales=1.0)
    u = rng.uniform(0, 8, 2000)
    got = np.asarray(q.matrix(np.zeros(1), u))[0]
    ref = np.array([float(mp.exp(-mp.mpf(float(x)) ** 2 / 2)) for x in u])
    assert np.max(np.abs(got - ref) / ref) <= 4e-16 * 40        # the argument u^2/2 is rounded before the exponential: |arg| eps


def test_specialised_assembly_is_bit_identical(ctx):
    """`assemble_fast_kernel` (one product-form group, D <= 2, <= 2 parity classes, degrees <= 4: polynomial degrees as
    template parameters, descriptor by value) against the generic `assemble_kernel` (option `asm_fast` = 0): the same
    arithmetic operation for operation, so the blocks must be IDENTICAL -- 1-D Matern of every order pair the lowering
    produces (even and odd parity classes, degrees 0..4), ExpQuad, 2-D products incl. the heat operator (two classes),
    ragged sizes, a symmetric diagonal block; and a descriptor outside the shape (two groups) takes the generic path."""
    import linpde_gp_amd as lp
    from linpde_gp_amd.linfuncops import diffops
    cf = lp.randprocs.covfuncs
    rng = np.random.default_rng(9)
    cases = []
    x0, x1 = rng.uniform(-2, 2, 150), rng.uniform(-2, 2, 77)
    for nu in (0.5, 1.5, 2.5, 3.5, 4.5):
        k = cf.Matern((), nu=nu, lengthscales=0.8)
        p = int(nu - 0.5)
        for a_, b_ in [(0, 0), (1, 0), (0, 1), (1, 1), (2, 0), (2, 2), (2, 1)]:
            if a_ + b_ <= 2 * p and max(a_, b_) <= p:
                cases.append((diffops.Derivative(a_)(diffops.Derivative(b_)(k, argnum=1), argnum=0) if a_ + b_ else k, x0, x1))
    ke = 2.0 * cf.ExpQuad((), lengthscales=0.4)
    cases += [(ke, x0, x1), (diffops.Derivative(2)(diffops.Derivative(1)(ke, argnum=1), argnum=0), x0, x1)]
    X0, X1 = rng.uniform(-1, 1, (130, 2)), rng.uniform(-1, 1, (200, 2))
    k2 = 4.0 * cf.TensorProduct(cf.Matern((), nu=2.5, lengthscales=1.0), cf.Matern((), nu=2.5, lengthscales=0.7))
    D = -1.0 * diffops.Laplacian((2,))
    kh = cf.TensorProduct(cf.Matern((), nu=1.5, lengthscales=2.5), cf.Matern((), nu=2.5, lengthscales=2.0))
    H = diffops.HeatOperator((2,), alpha=0.1)
    km = cf.TensorProduct(cf.ExpQuad((), lengthscales=0.5), cf.Matern((), nu=3.5, lengthscales=0.8))
    cases += [(k2, X0, X1), (D(k2, argnum=1), X0, X1), (D(D(k2, argnum=1), argnum=0), X0, X1), (H(kh, argnum=0), X0, X1),
              (H(H(kh, argnum=1), argnum=0), X0, X1), (km, X0, X1), (D(km, argnum=0), X0, X1), (k2, X0, None),
              (k2 + kh, X0, X1)]                                     # (last: two groups -> generic path either way)
    try:
        for k, a0, a1 in cases:
            V = rng.standard_normal(((a0 if a1 is None else a1).shape[0], 3))
            ctx.set_option("asm_fast", 0)
            ref = k.matrix(a0) if a1 is None else k.matrix(a0, a1)
            ref_mv = (k.linop(a0) if a1 is None else k.linop(a0, a1)) @ V
            ctx.set_option("asm_fast", 1)
            got = k.matrix(a0) if a1 is None else k.matrix(a0, a1)
            np.testing.assert_array_equal(got, ref)
            # the matrix-free product shares the specialised evaluation (matvec_fast_kernel): same entries, same order of sums
            np.testing.assert_array_equal((k.linop(a0) if a1 is None else k.linop(a0, a1)) @ V, ref_mv)
    finally:
        ctx.set_option("asm_fast", 1)


def test_column_tiles_per_workgroup_give_the_same_bits(ctx):
    """`asm_ct` (assemble_fast_kernel walks several consecutive column tiles per workgroup on large launches): a ragged
    rectangular block, and a symmetric diagonal block written through the lower-triangle path of the Gram assembly (a
    conditioning on 6 000 scattered points), against one tile per workgroup -- identical bits."""
    import linpde_gp_amd as lp
    from linpde_gp_amd.linfuncops import diffops
    cf = lp.randprocs.covfuncs
    rng = np.random.default_rng(21)
    k = 1.5 * cf.TensorProduct(cf.Matern((), nu=2.5, lengthscales=0.9), cf.Matern((), nu=1.5, lengthscales=0.6))
    kk = diffops.PartialDerivative(diffops.MultiIndex((1, 0)))(k, argnum=1)          # an odd parity class
    X0, X1 = rng.uniform(-1, 1, (4096 - 29, 2)), rng.uniform(-1, 1, (9000 - 11, 2))
    Xo, Y = rng.uniform(-1, 1, (6000, 2)), rng.standard_normal(6000)
    Xt = rng.uniform(-1, 1, (33, 2))
    got = {}
    try:
        for ct in (1, 4):
            ctx.set_option("asm_ct", ct)
            M = np.asarray(kk.matrix(X0, X1))
            u = lp.GaussianProcess(lp.functions.Zero((2,)), k).condition_on_observations(
                Y, Xo, b=lp.randvars.Normal(np.zeros(6000), np.full(6000, 1e-2)))
            got[ct] = (M, *u.predict(Xt))
    finally:
        ctx.set_option("asm_ct", 4)
    for a, b in zip(got[1], got[4]):
        assert np.array_equal(a, b)


def test_per_point_exponential_factors_option(ctx):
    """Option `asm_factors` (off by default): the exponential of every Matern dimension from per-point factors
    `e^{-+a(x - x0)}` (two multiplies and a minimum per entry instead of an exp; eval_entries.h).  Entries against the
    oracle and against the default per-entry evaluation on the reference's kind of point set, on a grid-like set (tiles of
    small extent), on a product of Matern and ExpQuad dimensions (the ExpQuad keeps its per-entry exp), on an isotropic
    Matern (untouched), and on points spread over hundreds of length scales (per-tile fallback to the per-entry exp)."""
    import linpde_gp_amd as lp
    from linpde_gp_amd.linfuncops import diffops
    from oracle import covfuncs as ocf
    cf = lp.randprocs.covfuncs
    rng = np.random.default_rng(5)
    k2 = 4.0 * cf.TensorProduct(cf.Matern((), nu=2.5, lengthscales=1.0), cf.Matern((), nu=3.5, lengthscales=0.6))
    ok2 = [(4.0, [("matern", 2.5, 1.0), ("matern", 3.5, 0.6)])]
    D = -1.0 * diffops.Laplacian((2,))
    lap, ident = {(2, 0): -1.0, (0, 2): -1.0}, ocf.identity(2)
    km = 1.5 * cf.TensorProduct(cf.ExpQuad((), lengthscales=0.5), cf.Matern((), nu=1.5, lengthscales=0.8))
    okm = [(1.5, [("expquad", 0.5), ("matern", 1.5, 0.8)])]
    kiso = cf.Matern((2,), nu=2.5, lengthscales=[0.7, 1.1])
    g = np.linspace(-1, 1, 40)
    sets = [(rng.uniform(-3, 3, (150, 2)), rng.uniform(-3, 3, (77, 2))),
            (np.stack(np.meshgrid(g, g, indexing="ij"), axis=-1).reshape(-1, 2)[:300], np.stack(np.meshgrid(g, g, indexing="ij"), axis=-1).reshape(-1, 2)),
            (rng.uniform(-150, 150, (130, 2)), rng.uniform(-150, 150, (90, 2)))]
    try:
        for X0, X1 in sets:
            for k, ok, L0, L1 in [(k2, ok2, ident, ident), (D(k2, argnum=1), ok2, ident, lap), (D(D(k2, argnum=1), argnum=0), ok2, lap, lap),
                                  (km, okm, ident, ident)]:
                ctx.set_option("asm_factors", 0)
                K0 = k.matrix(X0, X1)
                ctx.set_option("asm_factors", 1)
                K1 = k.matrix(X0, X1)
                ref = ocf.LkL(ok, L0, L1, X0, X1)
                scale = np.max(np.abs(ref))
                assert np.max(np.abs(K1 - ref)) <= 1e-13 * scale and np.max(np.abs(K1 - K0)) <= 2e-14 * scale
                V = rng.standard_normal((X1.shape[0], 3))
                np.testing.assert_allclose(k.linop(X0, X1) @ V, K1 @ V, rtol=0, atol=1e-11 * scale * X1.shape[0])
            ctx.set_option("asm_factors", 0)
            Ki0 = kiso.matrix(X0, X1)
            ctx.set_option("asm_factors", 1)
            np.testing.assert_array_equal(kiso.matrix(X0, X1), Ki0)
    finally:
        ctx.set_option("asm_factors", 0)


def test_matrix_free_product_throughput(ctx):
    import time
    import linpde_gp_amd as lp
    from linpde_gp_amd.linfuncops import diffops
    cf = lp.randprocs.covfuncs
    k = 2.0**2 * cf.TensorProduct(cf.Matern((), nu=2.5), cf.Matern((), nu=2.5))
    D = -1.0 * diffops.Laplacian((2,))
    kk = D(D(k, argnum=1), argnum=0)
    g = np.linspace(-1, 1, 128)
    X = np.stack(np.meshgrid(g, g, indexing="ij"), axis=-1).reshape(-1, 2)
    op = kk.linop(X)
    V = np.random.default_rng(0).standard_normal((X.shape[0], 4))
    op @ V
    ctx.profile_reset(); ctx.profile_enable(["matvec"])
    t0 = time.perf_counter(); op @ V; dt = time.perf_counter() - t0
    p = ctx.profile_get()["matvec"]; ctx.profile_enable(False)
    n = X.shape[0]
    print(f"\n[matvec] N={n}, 4 rhs: kernel {p['ms']:.3f} ms = {n * n / p['ms'] / 1e6:.1f} G entries/s; "
          f"call incl. H2D/D2H {dt * 1e3:.2f} ms")
    assert p["ms"] > 0


@pytest.mark.parametrize("which,n", [(0, 128), (0, 640), (0, 16896), (1, 128), (1, 1152), (1, 4224)])
@pytest.mark.parametrize("cond", [1e1, 1e6])
def test_tile_solve_refined(which, n, cond):
    """The in-place tile solves of the panel chain (X <- X L^{-T} on n rows) and of the forward substitution
    (V <- L^{-1} V on n right-hand sides), `tile_solve_kernel`: a product with the explicit fp64 inverse plus
    ONE refinement step against the tile itself.  Must be as accurate as a substitution (LAPACK dtrsm) also
    for an ill-conditioned tile (cond 1e6), where the unrefined product X0 = A Linv^T is ~cond eps off."""
    import scipy.linalg
    import linpde_gp_amd  # noqa: F401
    from linpde_gp_amd import _engine
    ctx = _engine.default_context()
    rng = np.random.default_rng(100 * which + n % 97)
    # lower-triangular tile with prescribed condition number: Cholesky factor of Q diag(s) Q^T
    Q, _ = np.linalg.qr(rng.standard_normal((128, 128)))
    sv = np.logspace(0, -np.log10(cond), 128)
    L = np.linalg.cholesky((Q * sv**2) @ Q.T)
    Linv = scipy.linalg.solve_triangular(L, np.eye(128), lower=True)
    XV = rng.standard_normal((n, 128) if which == 0 else (128, n))
    exact = scipy.linalg.solve_triangular(L, XV.T if which == 0 else XV, lower=True)
    exact = exact.T if which == 0 else exact
    got, ms = _hooks.test_tile_step(ctx, which, XV, L, Linv)
    # backward error of the solve: residual against the right-hand side, relative to |X||L| -- the measure that is
    # eps for a substitution and cond * eps for a bare product with the explicit inverse
    res = (got @ L.T - XV) if which == 0 else (L @ got - XV)
    scale = (np.abs(got) @ np.abs(L.T)) if which == 0 else (np.abs(L) @ np.abs(got))
    assert np.max(np.abs(res) / scale) < 2e-14, np.max(np.abs(res) / scale)
    np.testing.assert_allclose(got, exact, rtol=0, atol=1e-9 * cond / 1e6 * np.abs(exact).max() + 1e-13 * np.abs(exact).max())




@pytest.mark.parametrize("nt,cols", [(1, 128), (2, 1152), (3, 256), (4, 4224)])
def test_fused_panel_solve(nt, cols):
    """`panel_solve_kernel`: the whole chain of a panel of the forward substitution -- nt refined tile solves with the
    rank-128 updates of the panel rows below in between -- in one launch, against LAPACK's triangular solve with the
    nt x nt tile block; backward error at the level of a substitution although every tile solve is a product with an
    explicit inverse (refined once)."""
    import scipy.linalg
    import linpde_gp_amd  # noqa: F401
    from linpde_gp_amd import _engine
    ctx = _engine.default_context()
    rng = np.random.default_rng(7 * nt + cols)
    n = nt * 128
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    L = np.linalg.cholesky((Q * np.logspace(0, -5, n) ** 2) @ Q.T)          # cond(L) = 1e5
    V = rng.standard_normal((n, cols))
    got, ms = _hooks.test_panel_solve(ctx, V, L)
    exact = scipy.linalg.solve_triangular(L, V, lower=True)
    res = np.abs(L @ got - V) / (np.abs(L) @ np.abs(got))
    assert np.max(res) < 5e-14, np.max(res)
    np.testing.assert_allclose(got, exact, rtol=0, atol=1e-9 * np.abs(exact).max())
    # the rows orientation (panel solve of the multi-GPU factorisation / block append): X <- X L^{-T}
    got_r, _ = _hooks.test_panel_solve(ctx, np.ascontiguousarray(V.T), L, rows_form=True)
    np.testing.assert_allclose(got_r, exact.T, rtol=0, atol=1e-9 * np.abs(exact).max())
    res_r = np.abs(got_r @ L.T - V.T) / (np.abs(got_r) @ np.abs(L.T))
    assert np.max(res_r) < 5e-14, np.max(res_r)


@pytest.mark.parametrize("chain_bound", [True, False])
def test_two_level_forward_substitution_small(ctx, chain_bound):
    """`trsm_lower_two_level` (potrf.hip; on by default from 384 tile rows on, i.e. only at c4's size) brought down to a
    size the regular suite reaches: outer blocks of 1024 rows from 16 tile rows on, a RAGGED last outer block
    (21 tile rows = 8 + 8 + 5), 1 156 right-hand-side columns, and both branches of its schedule -- the far update
    released with the block (update-bound) and after the look-ahead half (chain-bound), forced through the chain estimate.
    predict() must agree with the plain right-looking solve and with the oracle."""
    from conftest import posterior_tolerances
    from linpde_gp_amd import problems
    from oracle import workloads as owl
    wl = problems.poisson_2d(n_side=50, n_bdry=28, m_side=34)
    assert wl.n_total > 20 * 128 and wl.Xtest.shape[0] >= 1024
    ref = owl.run(wl)
    saved = {k: ctx.get_option(k) for k in ("nb_outer_solve", "nb_outer_solve_min_tiles", "solve_chain_us_tile", "chain_us_fixed")}
    try:
        ctx.set_option("nb_outer_solve", 0)
        u, mean0, var0 = problems.condition_and_predict(wl)
        ctx.profile_reset(); ctx.profile_enable(True)
        ctx.set_option("nb_outer_solve", 1024)
        ctx.set_option("nb_outer_solve_min_tiles", 16)
        ctx.set_option("solve_chain_us_tile", 1000000 if chain_bound else 0)
        ctx.set_option("chain_us_fixed", 0)
        mean1, var1 = u.predict(wl.Xtest)
        prof = ctx.profile_get(); ctx.profile_enable(False)
    finally:
        for k, v in saved.items():
            ctx.set_option(k, v)
    # the two-level path ran: its far updates contract over 1024 rows (the plain path never exceeds its panel width)
    assert prof["panel_fused"]["launches"] >= 6
    ma, va = posterior_tolerances(ref["mean"], ref["var"])
    assert np.max(np.abs(mean1 - ref["mean"])) <= ma and np.max(np.abs(var1 - ref["var"])) <= va
    assert np.max(np.abs(mean0 - ref["mean"])) <= ma and np.max(np.abs(var0 - ref["var"])) <= va
    np.testing.assert_allclose(mean1, mean0, rtol=0, atol=ma)
    np.testing.assert_allclose(var1, var0, rtol=0, atol=va)


@pytest.mark.parametrize("n_side,n_bdry", [(40, 20), (41, 20), (43, 20), (44, 20), (50, 28)])      # 17, 18, 19, 20, 24 tile rows: last panels of 1, 2, 3, 4, 4
def test_forward_substitution_with_fused_look_ahead(ctx, n_side, n_bdry):
    """Round 4: the look-ahead update of the blocked forward substitution rides in front of the next panel's fused chain
    (`panel_solve_kernel<NT, 1, true, 4>`, option `fused_ahead`, default on) instead of being a launch of its own.  Ragged last
    panels of 1, 2, 3 and 4 tile rows (NT = 1 ... 4), 1 156 right-hand-side columns: predict() with and without it, both
    against the oracle with the one criterion."""
    from conftest import posterior_tolerances
    from linpde_gp_amd import problems
    from oracle import workloads as owl
    wl = problems.poisson_2d(n_side=n_side, n_bdry=n_bdry, m_side=34)
    ref = owl.run(wl)
    assert ctx.get_option("fused_ahead") == 1
    try:
        ctx.set_option("fused_ahead_min_us", 0)            # (by default only while the remainder update is long: c3's first panels)
        ctx.profile_reset(); ctx.profile_enable(True)
        u, mean1, var1 = problems.condition_and_predict(wl)
        prof = ctx.profile_get(); ctx.profile_enable(False)
        ctx.set_option("fused_ahead", 0)
        mean0, var0 = u.predict(wl.Xtest)
    finally:
        ctx.set_option("fused_ahead", 1)
        ctx.set_option("fused_ahead_min_us", 800)
    ma, va = posterior_tolerances(ref["mean"], ref["var"])
    assert np.max(np.abs(mean1 - ref["mean"])) <= ma and np.max(np.abs(var1 - ref["var"])) <= va
    assert np.max(np.abs(mean0 - ref["mean"])) <= ma and np.max(np.abs(var0 - ref["var"])) <= va
    np.testing.assert_allclose(mean1, mean0, rtol=0, atol=ma)
    np.testing.assert_allclose(var1, var0, rtol=0, atol=va)
