// Probe (round 4): how many streams of one process run kernels CONCURRENTLY (HIP multiplexes its streams over
// GPU_MAX_HW_QUEUES hardware queues, default 4), and does setting the variable from inside the process before the first
// HIP call raise it?   usage: hwqueue_probe <nstreams> [value for setenv GPU_MAX_HW_QUEUES]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ unsigned long long rt() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
__global__ void spin(unsigned long long* stamps, int us) {
  const unsigned long long t0 = rt();
  stamps[0] = t0;
  while (rt() - t0 < (unsigned long long)us * 100ull) { }
  stamps[1] = rt();
}
int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int n = argc > 1 ? std::atoi(argv[1]) : 6;
  if (argc > 2) setenv("GPU_MAX_HW_QUEUES", argv[2], 1);
  std::vector<hipStream_t> s(n);
  for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
  unsigned long long* st;
  CK(hipMalloc(&st, 2 * n * sizeof(unsigned long long)));
  for (int rep = 0; rep < 2; ++rep) {
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[i], st + 2 * i, 2000);
    CK(hipDeviceSynchronize());
  }
  std::vector<unsigned long long> h(2 * n);
  CK(hipMemcpy(h.data(), st, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  unsigned long long t0 = h[0];
  for (int i = 0; i < n; ++i) t0 = std::min(t0, h[2 * i]);
  int concurrent = 0;
  for (int i = 0; i < n; ++i) {
    std::printf("stream %d: start %+8.1f us, end %+8.1f us\n", i, (double)(h[2 * i] - t0) / 100.0, (double)(h[2 * i + 1] - t0) / 100.0);
    if ((double)(h[2 * i] - t0) / 100.0 < 1000.0) ++concurrent;
  }
  std::printf("GPU_MAX_HW_QUEUES=%s: %d of %d two-millisecond kernels on %d streams started within the first millisecond\n",
              std::getenv("GPU_MAX_HW_QUEUES") ? std::getenv("GPU_MAX_HW_QUEUES") : "(unset)", concurrent, n, n);
  return 0;
}
