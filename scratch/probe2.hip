#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// 64 accumulators (VGPR doubles), operands from registers, 4x4 operand reuse pattern like the GEMM
template <bool LDS>
__global__ __launch_bounds__(256, 2) void k(double* out, int iters) {
  __shared__ double sm[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = 1.0 + i * 1e-9;
  __syncthreads();
  double acc[4][16];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[t][u] = 0.0;
  const int lane = threadIdx.x & 63;
  double am[4], bn[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) am[t] = 1.0 + (threadIdx.x + t) * 1e-9;
#pragma unroll
  for (int v = 0; v < 4; ++v) bn[v] = 1.0 - (threadIdx.x + v) * 1e-9;
  for (int it = 0; it < iters; ++it) {
    if (LDS) {
#pragma unroll
      for (int t = 0; t < 4; ++t) am[t] = sm[((it & 3) * 4 + (lane >> 4)) * 144 % 3000 + t * 16 + (lane & 15)];
    }
#pragma unroll
    for (int uc = 0; uc < 4; ++uc) {
      if (LDS) {
#pragma unroll
        for (int v = 0; v < 4; ++v) bn[v] = sm[(((it & 3) * 4 + (lane >> 4)) * 144 + 2048) % 3900 + (uc * 4 + v) * 4 + (lane & 3)];
      }
#pragma unroll
      for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[t][uc * 4 + v] = __builtin_amdgcn_mfma_f64_4x4x4f64(bn[v], am[t], acc[t][uc * 4 + v], 0, 0, 0);
    }
  }
  double s = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int u = 0; u < 16; ++u) s += acc[t][u];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <bool LDS>
void run(int bpc, int iters) {
  int blocks = 256 * bpc;
  double* d; hipMalloc(&d, blocks * 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<LDS>, dim3(blocks), dim3(256), 0, 0, d, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<LDS>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * 64 * 512.0;
  printf("LDS=%d blocks/CU=%d: %.1f TFLOP/s (%.3f ms)\n", (int)LDS, bpc, flops / (ms * 1e-3) / 1e12, ms);
}
int main() { run<false>(1, 20000); run<false>(2, 20000); run<true>(1, 20000); run<true>(2, 20000); return 0; }
