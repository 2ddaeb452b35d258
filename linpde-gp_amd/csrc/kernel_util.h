// Device helpers shared by the MFMA kernels (gemm.hip, solve.hip).
#pragma once

#include <type_traits>

namespace lpgp {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS read whose completion the COMPILER does not track: with LDS-DMA in flight hipcc turns
// every wait for a ds_read result into s_waitcnt lgkmcnt(0), which also waits for the
// prefetch just issued for the next chunk.  The reads are therefore issued from inline asm
// and retired by hand-counted s_waitcnt lgkmcnt(N) (LDS operations return in order).
template <int OFF_DOUBLES>
__device__ __forceinline__ double lds_read_async(unsigned byte_addr) {
  static_assert(OFF_DOUBLES >= 0 && OFF_DOUBLES * 8 < 65536, "ds_read offset field");
  double d;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(byte_addr), "n"(OFF_DOUBLES * 8));
  return d;
}
#define LDS_WAIT(N) do { asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

}  // namespace lpgp
