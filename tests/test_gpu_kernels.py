"""Raw-kernel parity on the GPU, through the C ABI (ctypes): MFMA GEMM/SYRK variants,
the 128x128 tile Cholesky + inverse, and the roofline probes."""
import numpy as np
import pytest
import scipy.linalg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from linpde_gp_amd import _engine
    return _engine.default_context()


def test_probes(ctx):
    tf = ctx.probe_mfma_f64()
    gb = ctx.probe_hbm_write(1 << 30)
    print(f"\n[probe] fp64 MFMA issue loop: {tf:.1f} TFLOP/s ; streaming write: {gb:.0f} GB/s ; {ctx.device_info()}")
    assert tf > 10.0 and gb > 500.0


@pytest.fixture(params=["tile128", "tile64"])
def gemm_kernel(ctx, request):
    """Both GEMM kernels: launches with at most `small_tiles_max` 128x128 tiles run on the
    64x64-tile kernel (default 256); 0 forces the 128x128-tile kernel."""
    ctx.set_option("small_tiles_max", 0 if request.param == "tile128" else 1 << 20)
    yield request.param
    ctx.set_option("small_tiles_max", 256)


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
    out, _ = _engine.test_gemm(ctx, ta, tb, 0, -1.5, A_store, B_store, 0.5, C0, k)
    ref = 0.5 * C0 - 1.5 * Am @ Bm
    np.testing.assert_allclose(out, ref, rtol=1e-13, atol=1e-12)
    out0, _ = _engine.test_gemm(ctx, ta, tb, 0, 1.0, A_store, B_store, 0.0, np.full_like(C0, np.nan), k)
    np.testing.assert_allclose(out0, Am @ Bm, rtol=1e-13, atol=1e-12)


def test_syrk_lower_only(ctx, gemm_kernel):
    from linpde_gp_amd import _engine
    rng = np.random.default_rng(7)
    n, k = 640, 512
    P = rng.standard_normal((n, k))
    C0 = rng.standard_normal((n, n))
    out, _ = _engine.test_gemm(ctx, 0, 0, 1, -1.0, P, P, 1.0, C0, k)
    ref = C0 - P @ P.T
    edge = 128 if gemm_kernel == "tile128" else 64
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
    _, ms = _engine.test_gemm(ctx, 0, 0, 1, -1.0, P, P, 1.0, C0, k, reps=5)
    flops = n * (n + 1.0) * k
    print(f"\n[syrk] n={n} k={k}: {ms:.3f} ms  -> {flops / ms / 1e9:.1f} TFLOP/s algorithmic")
    _, ms = _engine.test_gemm(ctx, 0, 1, 0, -1.0, P, np.asfortranarray(rng.standard_normal((k, 4096))), 1.0,
                              np.zeros((n, 4096), order="F"), k, reps=5)
    print(f"[gemm nn] {n}x4096x{k}: {ms:.3f} ms -> {2.0 * n * 4096 * k / ms / 1e9:.1f} TFLOP/s")


@pytest.mark.parametrize("seed", [0, 1])
def test_potrf_tile(ctx, seed):
    from linpde_gp_amd import _engine
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((128, 160))
    A = M @ M.T + 1e-3 * np.eye(128)
    L, Linv, info = _engine.test_potrf_tile(ctx, A)
    assert info == 0
    Lref = np.linalg.cholesky(A)
    np.testing.assert_allclose(np.tril(L), Lref, rtol=1e-10, atol=1e-11)
    assert np.all(np.triu(L, 1) == 0.0)
    np.testing.assert_allclose(Linv, scipy.linalg.solve_triangular(Lref, np.eye(128), lower=True),
                               rtol=1e-8, atol=1e-9)
    assert np.all(np.triu(Linv, 1) == 0.0)


def test_potrf_tile_not_pd(ctx):
    from linpde_gp_amd import _engine
    A = np.eye(128)
    A[40, 40] = -1.0
    _, _, info = _engine.test_potrf_tile(ctx, A)
    assert info == 41
