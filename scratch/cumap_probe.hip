// CU-mask bit -> (XCC, SE, CU) : one stream per mask bit, one tiny kernel each, reading HW_ID / XCC_ID.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void who(unsigned* out, int slot) {
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[2 * slot] = hw; out[2 * slot + 1] = xcc & 0xf;
  }
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 8); hipMemset(d, 0xff, 256 * 8);
  for (int bit = 0; bit < 256; ++bit) {
    std::vector<uint32_t> mask(8, 0u);
    mask[bit / 32] = 1u << (bit % 32);
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, 8, mask.data()) != hipSuccess) { printf("bit %d: stream creation failed\n", bit); continue; }
    hipLaunchKernelGGL(who, dim3(1), dim3(64), 0, st, d, bit);
    hipStreamSynchronize(st);
    hipStreamDestroy(st);
  }
  std::vector<unsigned> h(512); hipMemcpy(h.data(), d, 2048, hipMemcpyDeviceToHost);
  printf("bit: xcc se cu   (HW_ID: cu_id [11:8], sh_id [12], se_id [15:13])\n");
  for (int bit = 0; bit < 256; ++bit) {
    unsigned hw = h[2 * bit];
    printf("%3d: %u %u %2u%s", bit, h[2 * bit + 1], (hw >> 13) & 7, (hw >> 8) & 15, (bit % 8 == 7) ? "\n" : "   |  ");
  }
  return 0;
}
