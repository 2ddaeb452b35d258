// Probe (round 4): can a stream wait for a word that a RUNNING kernel writes (hipStreamWaitValue32), and how long after the write
// does the waiting stream's next kernel start?  A tile-granular hand-off from the update kernel to the panel chain without a
// kernel boundary would rest on this.  Build: hipcc --offload-arch=gfx950 -O2 -o waitvalue_probe waitvalue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned long long rt() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }  // 100 MHz

__global__ void producer(unsigned* flag, unsigned long long* stamps, unsigned value, int us_before, int us_after) {
  const unsigned long long t0 = rt();
  while (rt() - t0 < (unsigned long long)us_before * 100ull) { }
  __threadfence_system();
  __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  stamps[0] = rt();
  while (rt() - t0 < (unsigned long long)(us_before + us_after) * 100ull) { }
  stamps[1] = rt();
}
__global__ void consumer(unsigned long long* stamps) { stamps[2] = rt(); }

static int g_wait_first = 0;
#include <chrono>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int run(const char* what, unsigned* flag) {
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  unsigned long long* stamps;
  CK(hipMalloc(&stamps, 3 * sizeof(unsigned long long)));
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipMemset(stamps, 0, 3 * sizeof(unsigned long long)));
    const unsigned value = (unsigned)rep + 1;
    const double h0 = now_ms();
    if (!g_wait_first) hipLaunchKernelGGL(producer, dim3(1), dim3(64), 0, sa, flag, stamps, value, 2000, 3000);
    hipError_t e = hipStreamWaitValue32(sb, flag, value, hipStreamWaitValueGte, 0xFFFFFFFFu);
    if (e != hipSuccess) { std::printf("%s: hipStreamWaitValue32 -> %s\n", what, hipGetErrorString(e)); return 0; }
    const double h1 = now_ms();
    std::printf("%s rep %d: wait enqueued, the call took %.3f ms of host time\n", what, rep, h1 - h0);
    hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, sb, stamps);
    std::printf("%s rep %d: consumer enqueued (+%.3f ms)\n", what, rep, now_ms() - h1);
    if (g_wait_first) { hipLaunchKernelGGL(producer, dim3(1), dim3(64), 0, sa, flag, stamps, value, 2000, 3000); std::printf("%s rep %d: producer enqueued behind the wait (+%.3f ms)\n", what, rep, now_ms() - h1); }
    CK(hipDeviceSynchronize());
    unsigned long long h[3];
    CK(hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost));
    std::printf("%s rep %d: consumer started %.1f us after the write, %.1f us BEFORE the producer ended\n", what, rep,
                ((double)h[2] - (double)h[0]) / 100.0, ((double)h[1] - (double)h[2]) / 100.0);
  }
  return 0;
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int which = argc > 1 ? std::atoi(argv[1]) : 0;
  g_wait_first = argc > 2 ? std::atoi(argv[2]) : 0;
  unsigned* sig = nullptr;
  hipError_t e = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory);
  if (e == hipSuccess) { CK(hipMemset(sig, 0, 8)); if (which == 0 && run("signal memory", sig)) return 1; }
  else std::printf("hipExtMallocWithFlags(hipMallocSignalMemory) -> %s\n", hipGetErrorString(e));
  unsigned* plain = nullptr;
  CK(hipMalloc(&plain, 8));
  CK(hipMemset(plain, 0, 8));
  if (which == 1 && run("hipMalloc memory", plain)) return 1;
  unsigned* host = nullptr;
  CK(hipHostMalloc(&host, 8, hipHostMallocCoherent));
  *host = 0;
  if (which == 2 && run("coherent host memory", host)) return 1;
  return 0;
}
