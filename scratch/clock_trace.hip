// Core clock of the chip over time, sampled from inside: one resident wave reads s_memtime (core clocks) and s_memrealtime
// (100 MHz) every ~50 us for `seconds` seconds; run it BESIDE another process's work (bench.py) to see what the shader clock
// does during a step.  Output: t_ms  MHz  (one line per sample, averaged over `avg` samples).
// build: hipcc --offload-arch=gfx950 -O3 -o clock_trace clock_trace.hip ; run: ./clock_trace <seconds> <avg> > trace.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void sampler(unsigned long long* out, int n, unsigned long long period_ticks) {
  unsigned long long t_prev, c_prev;
  asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev), "=s"(c_prev) :: "memory");
  for (int i = 0; i < n; ++i) {
    unsigned long long t, c;
    do {
      __builtin_amdgcn_s_sleep(32);
      asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "=s"(c) :: "memory");
    } while (t - t_prev < period_ticks);
    if (threadIdx.x == 0) { out[2 * i] = t; out[2 * i + 1] = c - c_prev; out[2 * i] = t; }
    if (threadIdx.x == 0) out[2 * i] = (t << 0);
    if (threadIdx.x == 0) { out[2 * i] = t; out[2 * i + 1] = ((c - c_prev) << 20) | ((t - t_prev) & 0xfffff); }
    t_prev = t; c_prev = c;
  }
}
int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
  const int avg = argc > 2 ? atoi(argv[2]) : 4;
  const unsigned long long period = 5000;          // 50 us in 100 MHz ticks
  const int n = (int)(seconds * 1e8 / period);
  unsigned long long* d; CK(hipMalloc(&d, (size_t)n * 16)); CK(hipMemset(d, 0, (size_t)n * 16));
  hipLaunchKernelGGL(sampler, dim3(1), dim3(64), 0, 0, d, n, period);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(2 * (size_t)n); CK(hipMemcpy(h.data(), d, (size_t)n * 16, hipMemcpyDeviceToHost));
  const unsigned long long t0 = h[0];
  for (int i = 0; i + avg <= n; i += avg) {
    unsigned long long dc = 0, dt = 0;
    for (int j = 0; j < avg; ++j) { dc += h[2 * (i + j) + 1] >> 20; dt += h[2 * (i + j) + 1] & 0xfffff; }
    printf("%.3f %.0f\n", (double)(h[2 * i] - t0) / 1e5, dt ? (double)dc / (double)dt * 100.0 : 0.0);
  }
  return 0;
}
