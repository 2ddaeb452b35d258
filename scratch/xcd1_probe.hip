// On which XCD does a ONE-workgroup kernel land?  Sequences of launches on one stream (as the panel
// chain issues them): 1-block kernels between kernels of other sizes, with a busy second stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void one(int* out, int slot) {
  extern __shared__ double sm[];
  if (threadIdx.x == 0) {
    unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    out[slot] = (int)(x & 0xf);
  }
  if (sm[0] == 1.2345) out[0] = -1;
}
__global__ void filler(double* p, int iters) {
  double a = threadIdx.x;
  for (int i = 0; i < iters; ++i) a = a * 1.0000001 + 1e-9;
  if (a == 12345.678) p[0] = a;
}
int main() {
  int* d; hipMalloc(&d, 4096 * 4); hipMemset(d, 0xff, 4096 * 4);
  double* p; hipMalloc(&p, 8);
  hipFuncSetAttribute((const void*)one, hipFuncAttributeMaxDynamicSharedMemorySize, 159744);
  int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
  hipStream_t s1, s2;
  hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi);
  hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, lo);
  int slot = 0;
  const int sizes[] = {1, 3, 7, 8, 9, 130, 262, 1000};
  for (int rep = 0; rep < 6; ++rep)
    for (int sz : sizes) {
      hipLaunchKernelGGL(filler, dim3(5000), dim3(256), 0, s2, p, 2000);        // background load
      hipLaunchKernelGGL(filler, dim3(sz), dim3(256), 0, s1, p, 100);           // a chain kernel of sz blocks
      hipLaunchKernelGGL(one, dim3(1), dim3(512), 159744, s1, d, slot++);       // the 1-block kernel (whole-CU LDS)
    }
  hipDeviceSynchronize();
  std::vector<int> h(slot); hipMemcpy(h.data(), d, slot * 4, hipMemcpyDeviceToHost);
  printf("xcc of the 1-block kernel after a kernel of {1,3,7,8,9,130,262,1000} blocks on the same stream:\n");
  for (int i = 0; i < slot; ++i) printf("%d%s", h[i], (i % 8 == 7) ? "\n" : " ");
  return 0;
}
