#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
#include <map>
__global__ void who(unsigned* out) {
  unsigned hwid, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // burn a bit so that all CUs get used
  double x = threadIdx.x;
  for (int i = 0; i < 20000; ++i) x = x * 1.0000001 + 1e-9;
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hwid; out[2 * blockIdx.x + 1] = (xcc & 0xf) | (x > 1e300 ? 16 : 0); }
}
void run(hipStream_t st, const char* name) {
  const int blocks = 4096;
  unsigned* d; hipMalloc(&d, blocks * 8);
  hipLaunchKernelGGL(who, dim3(blocks), dim3(64), 0, st, d);
  hipStreamSynchronize(st);
  std::vector<unsigned> h(blocks * 2);
  hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, std::set<unsigned>> per_xcc;
  for (int b = 0; b < blocks; ++b) {
    unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
    unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
  }
  int total = 0;
  printf("%s:", name);
  for (auto& kv : per_xcc) { printf(" xcc%u:%zu", kv.first, kv.second.size()); total += kv.second.size(); }
  printf("  total distinct CUs %d\n", total);
  hipFree(d);
}
int main() {
  hipStream_t s0; hipStreamCreate(&s0); run(s0, "no mask");
  for (int variant = 0; variant < 4; ++variant) {
    std::vector<uint32_t> mask(8, 0xffffffffu);
    const char* name = "";
    if (variant == 0) { mask[0] &= ~1u; name = "clear bit 0"; }
    if (variant == 1) { mask[0] &= ~0xffu; name = "clear bits 0-7"; }
    if (variant == 2) { for (int w = 0; w < 8; ++w) mask[w] &= ~1u; name = "clear bit 0 of each word"; }
    if (variant == 3) { mask[0] = 0xffff0000u; name = "clear bits 0-15"; }
    hipStream_t s; hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask.data());
    if (e != hipSuccess) { printf("%s: create failed %s\n", name, hipGetErrorString(e)); continue; }
    run(s, name);
  }
  return 0;
}
