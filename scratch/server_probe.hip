// Feasibility probe for a persistent single-workgroup "server" that a stream hands work to and waits for by stream memory
// operations (hipStreamWriteValue32 / hipStreamWaitValue32), next to a chip full of low-priority bulk workgroups:
//   (1) does the device support the stream operations, on which kind of memory;
//   (2) round trip  kernel A -> write READY -> [server: poll, ~35 us of work, release DONE] -> wait DONE -> kernel B
//       against the same chain with the work as an ordinary launch, idle chip and beside the bulk kernel;
//   (3) does a multi-workgroup kernel of the high-priority stream get slots promptly beside an UNMASKED bulk kernel.
// build: hipcc --offload-arch=gfx950 -O3 -o server_probe server_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long now100MHz() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }

__device__ void spin_work(unsigned long long ticks) { const unsigned long long t0 = now100MHz(); while (now100MHz() - t0 < ticks) __builtin_amdgcn_s_sleep(2); }

// the server: one workgroup, whole-CU LDS footprint
__global__ __launch_bounds__(512) void server_kernel(volatile unsigned* ready, unsigned* done, unsigned first, unsigned last, double* payload, int* status) {
  extern __shared__ double sm[];
  __shared__ int s_ok;
  for (unsigned seq = first; seq <= last; ++seq) {
    if (threadIdx.x == 0) {
      const unsigned long long t0 = now100MHz();
      int ok = 1;
      while (__hip_atomic_load((unsigned*)ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
        __builtin_amdgcn_s_sleep(4);
        if (now100MHz() - t0 > 200000000ull) { ok = 0; break; }          // 2 s: give up
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);      // (system scope by default in HIP)
      s_ok = ok;
    }
    __syncthreads();
    if (!s_ok) { if (threadIdx.x == 0) *status = -(int)seq; return; }
    sm[threadIdx.x] = payload[threadIdx.x];          // touch the payload the previous kernel wrote
    spin_work(3500);                                 // ~35 us
    payload[512 + threadIdx.x] = sm[threadIdx.x] + 1.0;
    __syncthreads();
    if (threadIdx.x == 0) {
      __atomic_thread_fence(__ATOMIC_RELEASE);
      __hip_atomic_store(done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (threadIdx.x == 0) *status = 1;
}
__global__ void work_kernel(double* payload, unsigned long long ticks) { extern __shared__ double sm[]; sm[threadIdx.x] = payload[threadIdx.x]; spin_work(ticks); payload[threadIdx.x] = sm[threadIdx.x]; }
// the bulk kernel: WGs of 256 threads, 73.7 KB LDS, ~145 us each
__global__ __launch_bounds__(256) void bulk_kernel(double* sink) { extern __shared__ double sm[]; sm[threadIdx.x] = 1.0; spin_work(14500); if (sm[threadIdx.x] == 2.0) sink[0] = 1.0; }

int main() {
  int can = 0; CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t sP, sS, sB; CK(hipStreamCreateWithPriority(&sP, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithPriority(&sS, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithPriority(&sB, hipStreamNonBlocking, lo));
  double* payload; CK(hipMalloc(&payload, 1024 * 8)); CK(hipMemset(payload, 0, 1024 * 8));
  double* sink; CK(hipMalloc(&sink, 8));
  int* status; CK(hipHostMalloc(&status, 4)); 
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&server_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&work_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 155 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&bulk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 74 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned N = 100;
  for (int memkind = 0; memkind < 2; ++memkind) {
    unsigned *flags = nullptr;
    if (memkind == 0) { if (hipExtMallocWithFlags((void**)&flags, 64, hipMallocSignalMemory) != hipSuccess) { printf("signal memory: allocation failed\n"); (void)hipGetLastError(); continue; } }
    else CK(hipMalloc((void**)&flags, 64));
    unsigned *ready = flags, *done = flags + 1;      // (signal memory is 8 bytes: both words live in it)
    for (int bulk = 0; bulk < 2; ++bulk) {
      for (int mode = 0; mode < 2; ++mode) {         // 0: ordinary launches, 1: server
        CK(hipMemset(flags, 0, 8)); *status = 0; CK(hipDeviceSynchronize());
        if (bulk) hipLaunchKernelGGL(bulk_kernel, dim3(512 * 40), dim3(256), 73728, sB, sink);     // ~40 rounds x 145 us = 5.8 ms
        if (mode == 1) hipLaunchKernelGGL(server_kernel, dim3(1), dim3(512), 155 * 1024, sS, (volatile unsigned*)ready, done, 1u, N, payload, status);
        CK(hipEventRecord(e0, sP));
        bool okops = true;
        for (unsigned i = 1; i <= N; ++i) {
          hipLaunchKernelGGL(work_kernel, dim3(64), dim3(512), 64 * 1024, sP, payload, 1000ull);       // "in-panel update": 64 WGs, 10 us
          if (mode == 0) {
            hipLaunchKernelGGL(work_kernel, dim3(1), dim3(512), 155 * 1024, sP, payload, 3500ull);    // "tile Cholesky" as a launch
          } else {
            if (hipStreamWriteValue32(sP, ready, i, 0) != hipSuccess || hipStreamWaitValue32(sP, done, i, hipStreamWaitValueGte, 0xffffffffu) != hipSuccess) { okops = false; break; }
          }
          hipLaunchKernelGGL(work_kernel, dim3(64), dim3(512), 64 * 1024, sP, payload, 1400ull);       // "tile solve": 64 WGs, 14 us
        }
        if (!okops) { printf("memkind %d: stream value operations refused (%s)\n", memkind, hipGetErrorString(hipGetLastError())); unsigned big = N + 1; CK(hipMemcpy(ready, &big, 4, hipMemcpyHostToDevice)); CK(hipDeviceSynchronize()); break; }
        CK(hipEventRecord(e1, sP));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipDeviceSynchronize());
        printf("flags in %s, %s, %s: %.1f us per step (work alone 59 us)  server status %d\n", memkind == 0 ? "signal memory" : "device memory",
               bulk ? "beside an unmasked bulk kernel" : "idle chip", mode ? "SERVER + stream value ops" : "ordinary launches", ms * 1e3 / N, *status);
      }
    }
    CK(hipFree(flags));
  }
  return 0;
}
