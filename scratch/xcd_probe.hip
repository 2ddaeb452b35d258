// Does block b of a grid still land on XCD b % 8 when the stream has a CU mask?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void probe(int* xcc, int iters) {
  extern __shared__ double sm[];
  if (threadIdx.x == 0) {
    unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    xcc[blockIdx.x] = (int)(x & 0xf);
  }
  // keep the block resident for a while so that the grid spans several rounds
  double a = threadIdx.x;
  for (int i = 0; i < iters; ++i) a = a * 1.0000001 + 1e-9;
  if (a == 12345.678) sm[0] = a;
}
int main() {
  const int blocks = 4096;
  int* d; hipMalloc(&d, blocks * 4);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
  for (int mode = 0; mode < 3; ++mode) {
    hipStream_t st;
    if (mode == 0) hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    else {
      std::vector<uint32_t> mask(8, 0xffffffffu);
      int reserve = mode == 1 ? 1 : 8;
      for (int r = 0; r < reserve; ++r) mask[r / 32] &= ~(1u << (r % 32));
      hipExtStreamCreateWithCUMask(&st, 8, mask.data());
    }
    hipMemsetAsync(d, 0xff, blocks * 4, st);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 73728, st, d, 20000);
    hipStreamSynchronize(st);
    std::vector<int> h(blocks); hipMemcpy(h.data(), d, blocks * 4, hipMemcpyDeviceToHost);
    int match = 0; int cnt[8][8] = {};
    for (int b = 0; b < blocks; ++b) { match += (h[b] == (b & 7)); cnt[b & 7][h[b] & 7]++; }
    printf("mode %d (%s): blocks with xcc == b%%8: %d / %d\n", mode, mode == 0 ? "unmasked" : (mode == 1 ? "1 CU masked" : "8 CUs masked"), match, blocks);
    printf("  first 24 blocks' xcc:"); for (int b = 0; b < 24; ++b) printf(" %d", h[b]); printf("\n");
    if (match != blocks) { printf("  row = b%%8, col = xcc:\n"); for (int i = 0; i < 8; ++i) { printf("   "); for (int j = 0; j < 8; ++j) printf(" %4d", cnt[i][j]); printf("\n"); } }
  }
  return 0;
}
