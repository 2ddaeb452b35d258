#define LPGP_EXPERIMENT 1
#include "../linpde-gp_amd/csrc/gemm.hip"
namespace lpgp { void set_error(const char* fmt, ...) {} void prof_begin(lpgp_ctx*, hipStream_t, int, double, double) {} void prof_end(lpgp_ctx*, hipStream_t) {} }
int main() {
  using namespace lpgp;
  const int m = 4096, n = 2048, k = 8192;
  double *A, *B, *C;
  hipMalloc(&A, (size_t)m * k * 8); hipMalloc(&B, (size_t)n * k * 8); hipMalloc(&C, (size_t)m * n * 8);
  hipMemset(A, 0, (size_t)m * k * 8); hipMemset(B, 0, (size_t)n * k * 8); hipMemset(C, 0, (size_t)m * n * 8);
  lpgp_ctx ctx;
  for (int mode : {0, 1, 2, 3})
  for (int mt : {16, 32}) {
    GemmArgs g; g.A = A; g.B = B; g.C = C; g.lda = m; g.ldb = n; g.ldc = m; g.mt = mt; g.nt = 16; g.k = k;
    g.alpha = -1; g.beta = 1; g.tri = 0; g.row_tile0 = g.col_tile0 = 0; g.ktrim = mode;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch_gemm(&ctx, 0, 0, 0, g, -1);
    hipEventRecord(e0);
    launch_gemm(&ctx, 0, 0, 0, g, -1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode=%d (skipDMA=%d skipBarrier=%d) mt=%d: %.3f ms  %.1f TF\n", mode, mode & 1, (mode >> 1) & 1, mt, ms,
           2.0 * mt * 128 * 2048 * k / ms / 1e9);
  }
  return 0;
}
