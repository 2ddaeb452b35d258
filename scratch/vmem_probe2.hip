#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v2f64 __attribute__((ext_vector_type(2)));
typedef int v4i32 __attribute__((ext_vector_type(4)));
#define STAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")

// MODE 0: global_load_dwordx4 -> VGPR ; 1: raw_buffer_load_b128 -> VGPR ; 2: global_load_lds_dwordx4 ; 3: global_load_dwordx2 x2 per KB
template <int MODE, int NL>
__global__ __launch_bounds__(256) void k(const double* __restrict__ src, size_t elems, unsigned long long* out, double* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const size_t base = ((size_t)blockIdx.x * 4 + w) * (size_t)NL * 128 * iters;
  unsigned long long ti = 0, td = 0;
  double acc = 0;
  // buffer descriptor covering the whole array
  v4i32 rsrc;
  {
    unsigned long long p = (unsigned long long)src;
    rsrc[0] = (int)(p & 0xffffffffu); rsrc[1] = (int)((p >> 32) & 0xffff); rsrc[2] = (int)0xffffffffu; rsrc[3] = 0x00020000;
  }
  for (int it = 0; it < iters; ++it) {
    unsigned long long t0, t1, t2;
    v2f64 r[NL];
    STAMP(t0);
#pragma unroll
    for (int q = 0; q < NL; ++q) {
      size_t off = (base + ((size_t)it * NL + q) * 128 + lane * 2) & (elems - 1);
      if (MODE == 0) {
        const double* p = src + off;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[q]) : "v"(p) : "memory");
      } else if (MODE == 1) {
        unsigned voff = (unsigned)((off * 8) & 0xfffffff0u);   // within first 4 GiB
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(r[q]) : "v"(voff), "s"(rsrc) : "memory");
      } else if (MODE == 2) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off),
                                         (__attribute__((address_space(3))) void*)(lds + (w * NL + q) * 128), 16, 0, 0);
      } else {
        const double* p = src + off;
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(r[q][0]) : "v"(p) : "memory");
      }
    }
    STAMP(t1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(t2);
    if (MODE != 2) {
#pragma unroll
      for (int q = 0; q < NL; ++q) { asm volatile("" : "+v"(r[q])); acc += r[q][0]; }
    } else {
      __syncthreads();
      acc += lds[tid];
    }
    ti += t1 - t0; td += t2 - t0;
  }
  if (lane == 0) { out[(blockIdx.x * 4 + w) * 2] = ti; out[(blockIdx.x * 4 + w) * 2 + 1] = td; }
  sink[blockIdx.x * 256 + tid] = acc;
}
template <int MODE, int NL>
void run(const char* name, int blocks, int iters, const double* src, size_t elems) {
  unsigned long long* o; double* sink; hipMalloc(&o, blocks * 64); hipMalloc(&sink, blocks * 2048);
  hipLaunchKernelGGL((k<MODE, NL>), dim3(blocks), dim3(256), 4 * NL * 1024, 0, src, elems, o, sink, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 8);
  hipMemcpy(h.data(), o, blocks * 64, hipMemcpyDeviceToHost);
  double si = 0, sd = 0; for (int i = 0; i < blocks * 4; ++i) { si += h[2 * i]; sd += h[2 * i + 1]; }
  printf("%-28s NL=%d blocks=%d: issue %.0f cyc/load, batch issue+wait %.0f\n", name, NL, blocks, si / (blocks * 4) / iters / NL, sd / (blocks * 4) / iters);
  hipFree(o); hipFree(sink);
}
int main() {
  size_t elems = (size_t)1 << 28;  // 2 GiB
  double* src; hipMalloc(&src, elems * 8); hipMemset(src, 0, elems * 8);
  run<0, 8>("global_load_dwordx4", 256, 200, src, elems);
  run<1, 8>("buffer_load_dwordx4 offen", 256, 200, src, elems);
  run<2, 8>("global_load_lds_dwordx4", 256, 200, src, elems);
  run<3, 8>("global_load_dwordx2", 256, 200, src, elems);
  run<0, 8>("global_load_dwordx4", 16, 200, src, elems);
  run<1, 8>("buffer_load_dwordx4 offen", 16, 200, src, elems);
  run<2, 8>("global_load_lds_dwordx4", 16, 200, src, elems);
  run<3, 8>("global_load_dwordx2", 16, 200, src, elems);
  return 0;
}
