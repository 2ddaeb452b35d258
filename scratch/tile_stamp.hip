// diagnostic build of potrf_tile_kernel with phase stamps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#define LPGP_TILE_STAMP 1
namespace lpgp { void set_error(const char* fmt, ...) {} }
#include "../linpde-gp_amd/csrc/lpgp_internal.h"
namespace lpgp { void prof_begin(lpgp_ctx*, hipStream_t, int, double, double) {} void prof_end(lpgp_ctx*, hipStream_t) {}
int launch_gemm(lpgp_ctx*, hipStream_t, int, int, const GemmArgs&, int) { return 0; } }
__device__ unsigned long long g_stamps[16];
#include "../linpde-gp_amd/csrc/potrf.hip"
int main() {
  using namespace lpgp;
  const int n = 128;
  std::vector<double> A(n * n);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i + j * n] = std::exp(-0.5 * (i - j) * (i - j) / 400.0) + (i == j ? 0.1 : 0.0);
  double *dA, *dL; int* dinfo;
  hipMalloc(&dA, n * n * 8); hipMalloc(&dL, n * n * 8); hipMalloc(&dinfo, 4);
  lpgp_ctx ctx;
  for (int rep = 0; rep < 3; ++rep) {
    hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice); hipMemset(dinfo, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    launch_potrf_tile(&ctx, 0, dA, n, dL, dinfo, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[16];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof(h));
    printf("tile kernel %.1f us; cycles: load %llu, diag %llu, panel %llu, trailing %llu, inverse %llu, writeback %llu (total %llu); wave-0 diagonal work alone %llu\n",
           ms * 1e3, h[0], h[1], h[2], h[3], h[4], h[5], h[0] + h[1] + h[2] + h[3] + h[4] + h[5], h[6]);
    printf("  background of wave 1 per step:"); for (int j = 1; j < 8; ++j) printf(" %llu", h[8 + j]); printf("\n");
  }
  return 0;
}
