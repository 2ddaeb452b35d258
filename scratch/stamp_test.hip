#define LPGP_STAMP 1
#include "../linpde-gp_amd/csrc/gemm.hip"
#include <vector>
#include <algorithm>
namespace lpgp { void set_error(const char* fmt, ...) {} void prof_begin(lpgp_ctx*, hipStream_t, int, double, double) {} void prof_end(lpgp_ctx*, hipStream_t) {} }
int main() {
  using namespace lpgp;
  const int m = 4096, n = 2048, k = 8192;
  double *A, *B, *C; unsigned long long* st;
  hipMalloc(&A, (size_t)m * k * 8); hipMalloc(&B, (size_t)n * k * 8); hipMalloc(&C, (size_t)m * n * 8);
  hipMemset(A, 0, (size_t)m * k * 8); hipMemset(B, 0, (size_t)n * k * 8); hipMemset(C, 0, (size_t)m * n * 8);
  lpgp_ctx ctx;
  for (int rep = 0; rep < 2; ++rep)
  for (int mt : {16, 32}) {
    GemmArgs g; g.A = A; g.B = B; g.C = C; g.lda = m; g.ldb = n; g.ldc = m; g.mt = mt; g.nt = 16; g.k = (mt == 16 ? k : 512);
    g.alpha = -1; g.beta = 1; g.tri = 0; g.row_tile0 = g.col_tile0 = 0; g.ktrim = 0;
    int blocks = ((mt / 8) * 2 + 7) / 8 * 8 * 64;
    hipMalloc(&st, blocks * 32); hipMemset(st, 0, blocks * 32); g.stamps = st;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    launch_gemm(&ctx, 0, 0, 0, g, -1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), st, blocks * 32, hipMemcpyDeviceToHost);
    double s[4] = {0, 0, 0, 0}; int cnt = 0;
    for (int b = 0; b < blocks; ++b) if (h[4 * b + 1]) { for (int j = 0; j < 4; ++j) s[j] += h[4 * b + j]; ++cnt; }
    int KT = g.k / 16;
    printf("mt=%d k=%d: %.3f ms, %d active blocks; per stage cycles: load-issue %.0f, mfma %.0f, wait+store %.0f, barrier %.0f\n",
           mt, g.k, ms, cnt, s[0] / cnt / KT, s[1] / cnt / KT, s[2] / cnt / KT, s[3] / cnt / KT);
    hipFree(st);
  }
  return 0;
}
