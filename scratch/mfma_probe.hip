// Standalone probe: fp64 MFMA issue rate, dependent latency, clock under load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void probe(double* out, unsigned long long* cyc, unsigned long long* rt, int iters) {
  v4f64 acc[NACC];
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (v4f64){0.0, 0.0, 0.0, 0.0};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * (size_t)blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

template <int NACC>
void run(int blocks_per_cu, int threads, int iters) {
  int cus = 256;
  int blocks = cus * blocks_per_cu;
  double* d; unsigned long long *c, *r;
  hipMalloc(&d, (size_t)blocks * threads * 8); hipMalloc(&c, blocks * 8); hipMalloc(&r, blocks * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(threads), 0, 0, d, c, r, iters / 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<NACC>, dim3(blocks), dim3(threads), 0, 0, d, c, r, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> hc(blocks), hr(blocks);
  hipMemcpy(hc.data(), c, blocks * 8, hipMemcpyDeviceToHost); hipMemcpy(hr.data(), r, blocks * 8, hipMemcpyDeviceToHost);
  double waves = (double)blocks * threads / 64;
  double nm = (double)iters * NACC;
  double flops = waves * nm * 2048.0;
  double cyc = (double)hc[blocks / 2], rt = (double)hr[blocks / 2];
  printf("NACC=%d waves/SIMD=%.1f: %.1f TFLOP/s wall %.3f ms | cycles/MFMA/wave %.1f | clock %.2f GHz (memtime/memrealtime*100MHz)\n",
         NACC, blocks_per_cu * threads / 64.0 / 4.0, flops / (ms * 1e-3) / 1e12, ms, cyc / nm, cyc / rt * 0.1);
  hipFree(d); hipFree(c); hipFree(r);
}

__global__ __launch_bounds__(256) void probe44(double* out, unsigned long long* cyc, unsigned long long* rt, int iters) {
  double acc[16];
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * (size_t)blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

void run44(int blocks_per_cu, int iters) {
  int blocks = 256 * blocks_per_cu, threads = 256;
  double* d; unsigned long long *c, *r;
  hipMalloc(&d, (size_t)blocks * threads * 8); hipMalloc(&c, blocks * 8); hipMalloc(&r, blocks * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe44, dim3(blocks), dim3(threads), 0, 0, d, c, r, iters / 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe44, dim3(blocks), dim3(threads), 0, 0, d, c, r, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> hc(blocks), hr(blocks);
  hipMemcpy(hc.data(), c, blocks * 8, hipMemcpyDeviceToHost); hipMemcpy(hr.data(), r, blocks * 8, hipMemcpyDeviceToHost);
  double waves = (double)blocks * threads / 64, nm = (double)iters * 16;
  double flops = waves * nm * 512.0;
  printf("4x4x4_4b waves/SIMD=%d: %.1f TFLOP/s wall %.3f ms | cycles/MFMA/wave %.1f | clock %.2f GHz\n",
         blocks_per_cu, flops / (ms * 1e-3) / 1e12, ms, (double)hc[blocks/2] / nm, (double)hc[blocks/2] / (double)hr[blocks/2] * 0.1);
}

int main() {
  run<2>(8, 256, 20000);   // 8 waves / SIMD
  run<4>(8, 256, 10000);
  run<4>(6, 256, 10000);
  run44(1, 20000); run44(2, 20000); run44(4, 10000);
  run<8>(1, 256, 20000);   // 1 wave / SIMD, 8 independent accumulators
  run<1>(1, 256, 100000);  // 1 wave / SIMD, dependent chain -> latency
  run<2>(1, 256, 50000);
  run<4>(1, 256, 40000);
  run<8>(2, 256, 20000);   // 2 waves / SIMD
  run<8>(4, 256, 10000);   // 4 waves / SIMD
  run<16>(1, 256, 10000);
  return 0;
}
