// Test hooks of liblpgp.so -- NOT part of the product.  Built into a library of its own (`liblpgp_testhooks.so`, build.sh),
// which links against liblpgp.so and is loaded only by tests/ and scratch/ (tests/_hooks.py): raw kernels on host buffers for
// unit tests and micro-benchmarks, the peak probes, and the host replay of the distributed tile enumeration.  Declared in
// include/lpgp_test.h.  `nm -D liblpgp.so | grep -E "lpgp_(test|probe|debug)"` is empty.

#include <cstdlib>

#include "lpgp_internal.h"
#include "lpgp_test.h"

namespace lpgp {

typedef double v4f64 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma_probe_kernel(double* out, int iters) {
  v4f64 acc[8];
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (v4f64){0.0, 0.0, 0.0, 0.0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * (int64_t)blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void write_probe_kernel(double* out, int64_t n2) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += stride)
    reinterpret_cast<double2*>(out)[i] = make_double2((double)i, 1.0);
}

}  // namespace lpgp

using namespace lpgp;

extern "C" {

// ---- raw kernels for unit tests / microbenchmarks ---------------------------------------------
int lpgp_test_gemm(lpgp_ctx* ctx, int32_t ta, int32_t tb, int32_t lower_only, int64_t m, int64_t n, int64_t k,
                   double alpha, const double* A, int64_t lda, const double* B, int64_t ldb, double beta,
                   double* C, int64_t ldc, int32_t reps, double* ms_per_rep) {
  LPGP_DEVICE(ctx);
  hipStream_t ts = ctx->s_main;
  if (const char* e = std::getenv("LPGP_TEST_GEMM_STREAM")) { const int v = std::atoi(e); ts = v == 1 ? ctx->s_upd : (v == 2 && ctx->s_upd_narrow ? ctx->s_upd_narrow : ctx->s_main); }
  LPGP_CHECK(m % TILE == 0 && n % TILE == 0 && k % 16 == 0, "lpgp_test_gemm: m,n multiples of 128 and k of 16 required");
  const int64_t a_elems = ta ? lda * m : lda * k;
  const int64_t b_elems = tb ? ldb * n : ldb * k;
  double *dA = nullptr, *dB = nullptr, *dC = nullptr;
  LPGP_HIP(hipMalloc(&dA, (size_t)a_elems * sizeof(double)));
  LPGP_HIP(hipMalloc(&dB, (size_t)b_elems * sizeof(double)));
  LPGP_HIP(hipMalloc(&dC, (size_t)ldc * n * sizeof(double)));
  LPGP_HIP(hipMemcpy(dA, A, (size_t)a_elems * sizeof(double), hipMemcpyHostToDevice));
  LPGP_HIP(hipMemcpy(dB, B, (size_t)b_elems * sizeof(double), hipMemcpyHostToDevice));
  LPGP_HIP(hipMemcpy(dC, C, (size_t)ldc * n * sizeof(double), hipMemcpyHostToDevice));
  GemmArgs g;
  g.A = dA; g.B = dB; g.C = dC; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.mt = (int)(m / TILE); g.nt = (int)(n / TILE); g.k = (int)k; g.alpha = alpha; g.beta = beta;
  g.tri = lower_only;
  int rc = launch_gemm(ctx, ts, ta, tb, g, -1);
  if (rc == 0 && hipStreamSynchronize(ts) != hipSuccess) rc = -1;
  if (rc == 0) LPGP_HIP(hipMemcpy(C, dC, (size_t)ldc * n * sizeof(double), hipMemcpyDeviceToHost));
  if (rc == 0 && reps > 0 && ms_per_rep) {
    hipEvent_t e0, e1;
    LPGP_HIP(hipEventCreate(&e0));
    LPGP_HIP(hipEventCreate(&e1));
    LPGP_HIP(hipEventRecord(e0, ts));
    for (int r = 0; r < reps && rc == 0; ++r) rc = launch_gemm(ctx, ts, ta, tb, g, -1);
    LPGP_HIP(hipEventRecord(e1, ts));
    LPGP_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    LPGP_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_rep = ms / reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
  (void)hipFree(dA);
  (void)hipFree(dB);
  (void)hipFree(dC);
  return rc;
}

int lpgp_test_stair_enumerate(int32_t pr, int32_t pc, int32_t my_r, int32_t my_c, int32_t nbt, int32_t T, int32_t row_lo,
                              int32_t col_lo, int32_t* out, int64_t cap) {
  LPGP_CHECK(pr >= 1 && pc >= 1 && my_r >= 0 && my_r < pr && my_c >= 0 && my_c < pc && nbt >= 1 && T >= 0 && out, "lpgp_test_stair_enumerate: bad argument");
  GemmArgs g;
  g.cyc = 1; g.tri = 1;
  g.rowc.P = pr; g.rowc.me = my_r; g.rowc.nbt = nbt;
  g.colc.P = pc; g.colc.me = my_c; g.colc.nbt = nbt;
  g.rt0 = cyc_before(g.rowc, row_lo);
  g.ct0 = cyc_before(g.colc, col_lo);
  g.mt = cyc_before(g.rowc, T) - g.rt0;
  g.nt = cyc_before(g.colc, T) - g.ct0;
  g.g0 = col_lo < row_lo ? col_lo : row_lo;
  if (g.mt <= 0 || g.nt <= 0) return 0;
  const int n = stair_enumerate_host(g, out, cap);
  for (int64_t i = 0; i < n && i < cap; ++i) {      // local -> global tile indices
    out[2 * i] = cyc_l2g(g.rowc, g.rt0 + out[2 * i]);
    out[2 * i + 1] = cyc_l2g(g.colc, g.ct0 + out[2 * i + 1]);
  }
  return n;
}

int lpgp_test_potrf_tile(lpgp_ctx* ctx, double* T, double* Linv, int32_t* info) {
  LPGP_DEVICE(ctx);
  double *dT = nullptr, *dL = nullptr;
  LPGP_HIP(hipMalloc(&dT, (size_t)TILE * TILE * sizeof(double)));
  LPGP_HIP(hipMalloc(&dL, (size_t)TILE * TILE * sizeof(double)));
  LPGP_HIP(hipMemcpy(dT, T, (size_t)TILE * TILE * sizeof(double), hipMemcpyHostToDevice));
  LPGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), ctx->s_main));
  int rc = launch_potrf_tile(ctx, ctx->s_main, dT, TILE, dL, ctx->d_info, 0);
  int h = 0;
  if (rc == 0) {
    LPGP_HIP(hipMemcpyAsync(&h, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->s_main));
    LPGP_HIP(hipStreamSynchronize(ctx->s_main));
    LPGP_HIP(hipMemcpy(T, dT, (size_t)TILE * TILE * sizeof(double), hipMemcpyDeviceToHost));
    LPGP_HIP(hipMemcpy(Linv, dL, (size_t)TILE * TILE * sizeof(double), hipMemcpyDeviceToHost));
  }
  if (info) *info = h;
  (void)hipFree(dT);
  (void)hipFree(dL);
  return rc;
}

int lpgp_test_tile_step(lpgp_ctx* ctx, int32_t which, double* XV, int64_t n, const double* L, const double* Linv, double* ms) {
  LPGP_CHECK(ctx && XV && L && Linv && n > 0 && n % TILE == 0 && (which == 0 || which == 1), "lpgp_test_tile_step: bad argument");
  LPGP_DEVICE(ctx);
  double *d = nullptr, *dl = nullptr, *dt = nullptr;
  const size_t bytes = (size_t)n * TILE * sizeof(double), tbytes = (size_t)TILE * TILE * sizeof(double);
  LPGP_HIP(hipMalloc(&d, bytes));
  LPGP_HIP(hipMalloc(&dl, tbytes));
  LPGP_HIP(hipMalloc(&dt, tbytes));
  LPGP_HIP(hipMemcpy(d, XV, bytes, hipMemcpyHostToDevice));
  LPGP_HIP(hipMemcpy(dl, Linv, tbytes, hipMemcpyHostToDevice));
  LPGP_HIP(hipMemcpy(dt, L, tbytes, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  LPGP_HIP(hipEventCreate(&e0));
  LPGP_HIP(hipEventCreate(&e1));
  const int nt = (int)(n / TILE);
  LPGP_HIP(hipEventRecord(e0, ctx->s_main));
  int rc = which == 0 ? launch_trsm_tile(ctx, ctx->s_main, d, n, dl, dt, TILE, nt, -1)
                      : launch_trsv_tile(ctx, ctx->s_main, d, TILE, dl, dt, TILE, nt, -1);
  LPGP_HIP(hipEventRecord(e1, ctx->s_main));
  LPGP_HIP(hipEventSynchronize(e1));
  float t = 0.f;
  LPGP_HIP(hipEventElapsedTime(&t, e0, e1));
  if (ms) *ms = t;
  if (rc == 0) LPGP_HIP(hipMemcpy(XV, d, bytes, hipMemcpyDeviceToHost));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(d);
  (void)hipFree(dl);
  (void)hipFree(dt);
  return rc;
}

int lpgp_test_panel_solve(lpgp_ctx* ctx, int32_t rows_form, double* V, int32_t nt, int64_t cols, const double* Lblk, const double* Linv, double* ms) {
  LPGP_CHECK(ctx && V && Lblk && Linv && nt >= 1 && nt <= 4 && cols > 0 && cols % TILE == 0, "lpgp_test_panel_solve: bad argument");
  LPGP_DEVICE(ctx);
  const int64_t rows = (int64_t)nt * TILE;
  double *d = nullptr, *dl = nullptr, *di = nullptr;
  const size_t vb = (size_t)rows * cols * sizeof(double), lb = (size_t)rows * rows * sizeof(double), ib = (size_t)nt * TILE * TILE * sizeof(double);
  LPGP_HIP(hipMalloc(&d, vb));
  LPGP_HIP(hipMalloc(&dl, lb));
  LPGP_HIP(hipMalloc(&di, ib));
  LPGP_HIP(hipMemcpy(d, V, vb, hipMemcpyHostToDevice));
  LPGP_HIP(hipMemcpy(dl, Lblk, lb, hipMemcpyHostToDevice));
  LPGP_HIP(hipMemcpy(di, Linv, ib, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  LPGP_HIP(hipEventCreate(&e0));
  LPGP_HIP(hipEventCreate(&e1));
  LPGP_HIP(hipEventRecord(e0, ctx->s_main));
  // rows_form: V holds X (cols rows x nt * 128 columns, column-major ld = cols) and X <- X Lblk^{-T}
  int rc = rows_form ? launch_trsm_panel(ctx, ctx->s_main, d, cols, di, dl, rows, nt, (int)(cols / TILE), -1)
                     : launch_trsv_panel(ctx, ctx->s_main, d, rows, di, dl, rows, nt, (int)(cols / TILE), -1);
  LPGP_HIP(hipEventRecord(e1, ctx->s_main));
  LPGP_HIP(hipEventSynchronize(e1));
  float t = 0.f;
  LPGP_HIP(hipEventElapsedTime(&t, e0, e1));
  if (ms) *ms = t;
  if (rc == 0) LPGP_HIP(hipMemcpy(V, d, vb, hipMemcpyDeviceToHost));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(d);
  (void)hipFree(dl);
  (void)hipFree(di);
  return rc;
}

int lpgp_debug_tile_xcc(lpgp_ctx* ctx, int32_t* out8, int32_t reset) {
  LPGP_CHECK(ctx && out8, "lpgp_debug_tile_xcc: null argument");
  LPGP_DEVICE(ctx);
  LPGP_HIP(hipDeviceSynchronize());
  return debug_tile_xcc(out8, reset);
}

int lpgp_probe_mfma_f64(lpgp_ctx* ctx, double* tflops) {
  LPGP_DEVICE(ctx);
  const int blocks = ctx->cus * 4, iters = 4000;
  double* d = nullptr;
  LPGP_HIP(hipMalloc(&d, (size_t)blocks * 256 * sizeof(double)));
  hipEvent_t e0, e1;
  LPGP_HIP(hipEventCreate(&e0));
  LPGP_HIP(hipEventCreate(&e1));
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, ctx->s_main, d, 100);   // warm-up
  LPGP_HIP(hipEventRecord(e0, ctx->s_main));
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, ctx->s_main, d, iters);
  LPGP_HIP(hipEventRecord(e1, ctx->s_main));
  LPGP_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  LPGP_HIP(hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)blocks * 4.0 * iters * 8.0 * (2.0 * 16 * 16 * 4);
  if (tflops) *tflops = flops / (ms * 1e-3) / 1e12;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(d);
  return 0;
}

int lpgp_probe_hbm_write(lpgp_ctx* ctx, int64_t bytes, double* gbps) {
  LPGP_DEVICE(ctx);
  double* d = nullptr;
  bytes = round_up(bytes, 16);
  LPGP_HIP(hipMalloc(&d, (size_t)bytes));
  hipEvent_t e0, e1;
  LPGP_HIP(hipEventCreate(&e0));
  LPGP_HIP(hipEventCreate(&e1));
  const int grid = ctx->cus * 8;
  hipLaunchKernelGGL(write_probe_kernel, dim3(grid), dim3(256), 0, ctx->s_main, d, bytes / 16);
  LPGP_HIP(hipEventRecord(e0, ctx->s_main));
  for (int r = 0; r < 5; ++r)
    hipLaunchKernelGGL(write_probe_kernel, dim3(grid), dim3(256), 0, ctx->s_main, d, bytes / 16);
  LPGP_HIP(hipEventRecord(e1, ctx->s_main));
  LPGP_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  LPGP_HIP(hipEventElapsedTime(&ms, e0, e1));
  if (gbps) *gbps = 5.0 * (double)bytes / (ms * 1e-3) / 1e9;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(d);
  return 0;
}

// the status word of `mat` as an enqueued factorisation that ended with `value` would leave it (value < 0: a timed-out hand-over)
int lpgp_test_force_status(lpgp_ctx* ctx, lpgp_mat* mat, int32_t value) {
  LPGP_CHECK(ctx && mat, "lpgp_test_force_status: null argument");
  LPGP_DEVICE(ctx);
  LPGP_HIP(hipStreamSynchronize(ctx->s_main));
  LPGP_HIP(hipMemcpy(mat->d_status, &value, sizeof(int), hipMemcpyHostToDevice));
  mat->unchecked = 1;
  mat->status_known = 0;
  return 0;
}

}  // extern "C"
