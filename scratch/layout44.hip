// Empirical lane-layout discovery for v_mfma_f64_4x4x4_4b_f64 with cbsz/abid broadcast.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CBSZ, int ABID>
__global__ void disc(int* out) {   // out[la*64+lb] = bitmask-lane encoded: first lane nonzero, count
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      double a = (lane == la) ? 1.0 : 0.0;
      double b = (lane == lb) ? 1.0 : 0.0;
      double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
      unsigned long long m = __ballot(d != 0.0);
      if (lane == 0) {
        out[(la * 64 + lb) * 2 + 0] = m ? (int)__builtin_ctzll(m) : -1;
        out[(la * 64 + lb) * 2 + 1] = (int)__builtin_popcountll(m);
      }
    }
}

template <int CBSZ, int ABID>
void run(const char* name) {
  int* d; hipMalloc(&d, 64 * 64 * 2 * 4);
  hipLaunchKernelGGL((disc<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, d);
  std::vector<int> h(64 * 64 * 2);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  printf("== %s\n", name);
  // for each A lane: list of (B lane -> D lane)
  for (int la = 0; la < 64; ++la) {
    printf("A%02d:", la);
    for (int lb = 0; lb < 64; ++lb) {
      int dl = h[(la * 64 + lb) * 2], c = h[(la * 64 + lb) * 2 + 1];
      if (dl >= 0) printf(" B%02d->D%02d%s", lb, dl, c > 1 ? "*" : "");
    }
    printf("\n");
  }
  hipFree(d);
}

int main() {
  run<0, 0>("cbsz=0 abid=0");
  run<2, 0>("cbsz=2 abid=0");
  run<2, 1>("cbsz=2 abid=1");
  run<2, 3>("cbsz=2 abid=3");
  return 0;
}
