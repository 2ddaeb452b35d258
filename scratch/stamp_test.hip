#define LPGP_STAMP 1
#include "../linpde-gp_amd/csrc/gemm.hip"
#include <vector>
#include <algorithm>
namespace lpgp { void set_error(const char* fmt, ...) {} void prof_begin(lpgp_ctx*, hipStream_t, int, double, double) {} void prof_end(lpgp_ctx*, hipStream_t) {} }
int main() {
  using namespace lpgp;
  const int m = 8192, n = 8192, kmax = 8192;
  double *A, *B, *C; unsigned long long* st;
  hipMalloc(&A, (size_t)m * kmax * 8); hipMalloc(&B, (size_t)n * kmax * 8); hipMalloc(&C, (size_t)m * n * 8);
  hipMemset(A, 0, (size_t)m * kmax * 8); hipMemset(B, 0, (size_t)n * kmax * 8); hipMemset(C, 0, (size_t)m * n * 8);
  std::vector<double> hr((size_t)m * kmax);
  for (int mode = 0; mode < 2; ++mode) {
  if (mode == 1) {
    unsigned long long x = 88172645463325252ull;
    for (auto& v : hr) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5; }
    hipMemcpy(A, hr.data(), hr.size() * 8, hipMemcpyHostToDevice); hipMemcpy(B, hr.data(), hr.size() * 8, hipMemcpyHostToDevice);
  }
  printf("---- data: %s\n", mode ? "random" : "zeros");
  lpgp_ctx ctx; ctx.cus = 256;
  struct Case { int mt, nt, k; };
  for (int rep = 0; rep < 2; ++rep)
  for (Case c : {Case{16, 16, 8192}, Case{32, 32, 4096}, Case{64, 64, 512}, Case{32, 16, 512}}) {
    GemmArgs g; g.A = A; g.B = B; g.C = C; g.lda = m; g.ldb = n; g.ldc = m; g.mt = c.mt; g.nt = c.nt; g.k = c.k;
    g.alpha = -1; g.beta = 1; g.tri = 0;
    int nv = 64 * 64 * 2;
    hipMalloc(&st, (size_t)nv * 64); hipMemset(st, 0, (size_t)nv * 64); g.stamps = st;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    launch_gemm(&ctx, 0, 0, 0, g, -1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)nv * 8);
    hipMemcpy(h.data(), st, (size_t)nv * 64, hipMemcpyDeviceToHost);
    double s[7] = {0, 0, 0, 0, 0, 0, 0}; int cnt = 0;
    for (int b = 0; b < nv; ++b) if (h[8 * b + 5]) { for (int j = 0; j < 7; ++j) s[j] += h[8 * b + j]; ++cnt; }
    int KT = g.k / 16;
    double fl = 2.0 * c.mt * 128.0 * c.nt * 128.0 * c.k;
    printf("mt=%d nt=%d k=%d: %.3f ms %.1f TF, %d tiles; per stage (memtime ticks): dma-issue %.0f, frag+mfma %.0f, vmwait %.0f, barrier %.0f | per tile: prologue %.0f kloop %.0f | memtime rate %.3f GHz\n",
           c.mt, c.nt, g.k, ms, fl / ms / 1e9, cnt, s[0] / cnt / KT, s[1] / cnt / KT, s[2] / cnt / KT, s[3] / cnt / KT, s[4] / cnt, s[5] / cnt, s[5] / s[6] * 0.1);
    hipFree(st);
  }
  }
  return 0;
}
