// Cost of a dependency that crosses streams: a chain of tiny kernels on ONE stream against the same
// chain alternating between two streams through events (record -> hipStreamWaitEvent).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void tiny(double* p) { if (threadIdx.x == 0) p[0] += 1.0; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  double* d; hipMalloc(&d, 8); hipMemset(d, 0, 8);
  int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
  hipStream_t s1, s2, s3;
  hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi);
  hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, lo);
  std::vector<uint32_t> mask(8, 0xffffffffu); mask[0] &= ~0xffu;
  hipExtStreamCreateWithCUMask(&s3, 8, mask.data());
  const int n = 400;
  std::vector<hipEvent_t> ev(2 * n);
  for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  for (int mode = 0; mode < 4; ++mode) {
    hipStream_t other = mode == 3 ? s3 : s2;
    for (int rep = 0; rep < 3; ++rep) {
      hipDeviceSynchronize();
      const double t0 = now();
      for (int i = 0; i < n; ++i) {
        if (mode == 0) {                       // one stream
          hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s1, d);
          hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s1, d);
        } else if (mode == 1) {                // one stream, an event record between the kernels (no wait)
          hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s1, d);
          hipEventRecord(ev[2 * i], s1);
          hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s1, d);
        } else {                               // ping-pong: s1 -> other -> s1
          hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s1, d);
          hipEventRecord(ev[2 * i], s1);
          hipStreamWaitEvent(other, ev[2 * i], 0);
          hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, other, d);
          hipEventRecord(ev[2 * i + 1], other);
          hipStreamWaitEvent(s1, ev[2 * i + 1], 0);
        }
      }
      hipDeviceSynchronize();
      const double dt = now() - t0;
      if (rep == 2) printf("mode %d (%s): %.2f us per kernel\n", mode,
                           mode == 0 ? "one stream" : mode == 1 ? "one stream + event records" : mode == 2 ? "ping-pong with a normal stream" : "ping-pong with a CU-masked stream",
                           dt * 1e6 / (2 * n));
    }
  }
  return 0;
}
