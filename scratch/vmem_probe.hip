#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double v2f64 __attribute__((ext_vector_type(2)));
// each wave: NL loads of 1 KB (dwordx4 per lane), then wait; measure issue time and total latency
template <int NL>
__global__ __launch_bounds__(256) void k(const double* __restrict__ src, size_t stride_elems, unsigned long long* out, double* sink, int iters) {
  const int tid = threadIdx.x;
  const size_t base = ((size_t)blockIdx.x * 4 + (tid >> 6)) * (size_t)NL * 128 * iters;   // per-wave private stream (doubles)
  unsigned long long t_issue = 0, t_done = 0;
  double acc = 0;
  for (int it = 0; it < iters; ++it) {
    v2f64 r[NL];
    unsigned long long t0, t1, t2;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma unroll
    for (int q = 0; q < NL; ++q) {
      const double* p = src + (base + ((size_t)it * NL + q) * 128 + (tid & 63) * 2) % stride_elems;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[q]) : "v"(p) : "memory");
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2) :: "memory");
#pragma unroll
    for (int q = 0; q < NL; ++q) { asm volatile("" : "+v"(r[q])); acc += r[q][0] + r[q][1]; }
    t_issue += t1 - t0; t_done += t2 - t0;
  }
  if ((tid & 63) == 0) { out[(blockIdx.x * 4 + (tid >> 6)) * 2] = t_issue; out[(blockIdx.x * 4 + (tid >> 6)) * 2 + 1] = t_done; }
  sink[blockIdx.x * 256 + tid] = acc;
}
template <int NL>
void run(int blocks, int iters, const double* src, size_t elems) {
  unsigned long long* o; double* sink; hipMalloc(&o, blocks * 4 * 16); hipMalloc(&sink, blocks * 256 * 8);
  hipLaunchKernelGGL(k<NL>, dim3(blocks), dim3(256), 0, 0, src, elems, o, sink, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 8);
  hipMemcpy(h.data(), o, blocks * 64, hipMemcpyDeviceToHost);
  double si = 0, sd = 0; for (int i = 0; i < blocks * 4; ++i) { si += h[2 * i]; sd += h[2 * i + 1]; }
  printf("NL=%d blocks=%d: issue %.0f cycles per batch (%.0f per load), issue+wait %.0f cycles\n", NL, blocks,
         si / (blocks * 4) / iters, si / (blocks * 4) / iters / NL, sd / (blocks * 4) / iters);
  hipFree(o); hipFree(sink);
}
int main() {
  size_t elems = (size_t)1 << 29;  // 4 GiB
  double* src; hipMalloc(&src, elems * 8); hipMemset(src, 0, elems * 8);
  run<8>(256, 200, src, elems); run<8>(512, 200, src, elems);
  run<4>(256, 200, src, elems); run<2>(256, 200, src, elems); run<1>(256, 200, src, elems);
  run<8>(8, 200, src, elems);
  return 0;
}
