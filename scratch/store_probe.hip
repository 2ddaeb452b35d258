// Store-pattern probe: write an n x n column-major matrix (ld = n + 512) in tiles, no compute.
#include <hip/hip_runtime.h>
#include <cstdio>
// A: 64x64 tile, lane = row, wave = 16 cols, 8-B stores (the assembly kernel's pattern)
__global__ __launch_bounds__(256) void pat_a(double* out, long ld, int tiles_r) {
  int tr = blockIdx.x % tiles_r, tc = blockIdx.x / tiles_r;
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  long row = (long)tr * 64 + lane;
  for (int e = 0; e < 16; ++e) { long c = (long)tc * 64 + w * 16 + e; out[row + c * ld] = (double)(row + c); }
}
// B: 128x32 tile, lane = 2 rows (16-B stores), wave = 8 cols
__global__ __launch_bounds__(256) void pat_b(double* out, long ld, int tiles_r) {
  int tr = blockIdx.x % tiles_r, tc = blockIdx.x / tiles_r;
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  long row = (long)tr * 128 + 2 * lane;
  for (int e = 0; e < 8; ++e) { long c = (long)tc * 32 + w * 8 + e; *(double2*)(out + row + c * ld) = make_double2((double)(row + c), 1.0); }
}
// C: 256x16 tile, lane = 4 rows (2 x 16-B stores), wave = 4 cols
__global__ __launch_bounds__(256) void pat_c(double* out, long ld, int tiles_r) {
  int tr = blockIdx.x % tiles_r, tc = blockIdx.x / tiles_r;
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  long row = (long)tr * 256 + 2 * lane;
  for (int e = 0; e < 4; ++e) { long c = (long)tc * 16 + w * 4 + e;
    *(double2*)(out + row + c * ld) = make_double2((double)(row + c), 1.0);
    *(double2*)(out + row + 128 + c * ld) = make_double2((double)(row + c), 2.0); }
}
// D: like A but the four waves of a block take the SAME 16 columns of 4 consecutive row tiles (256 x 16)
__global__ __launch_bounds__(256) void pat_d(double* out, long ld, int tiles_r) {
  int tr = blockIdx.x % tiles_r, tc = blockIdx.x / tiles_r;
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  long row = (long)tr * 256 + w * 64 + lane;
  for (int e = 0; e < 16; ++e) { long c = (long)tc * 16 + e; out[row + c * ld] = (double)(row + c); }
}
// E: pattern A with non-temporal stores
__global__ __launch_bounds__(256) void pat_e(double* out, long ld, int tiles_r) {
  int tr = blockIdx.x % tiles_r, tc = blockIdx.x / tiles_r;
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  long row = (long)tr * 64 + lane;
  for (int e = 0; e < 16; ++e) { long c = (long)tc * 64 + w * 16 + e; __builtin_nontemporal_store((double)(row + c), out + row + c * ld); }
}
// F: column-fastest block order for pattern A (consecutive blocks = neighbouring column tiles)
__global__ __launch_bounds__(256) void pat_f(double* out, long ld, int tiles_r) {
  int tc = blockIdx.x % tiles_r, tr = blockIdx.x / tiles_r;
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  long row = (long)tr * 64 + lane;
  for (int e = 0; e < 16; ++e) { long c = (long)tc * 64 + w * 16 + e; out[row + c * ld] = (double)(row + c); }
}
int main() {
  const long n = 16384, ld = n + 512;
  double* d; hipMalloc(&d, ld * n * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, void (*k)(double*, long, int), int th, int tw) {
    int tiles_r = n / th, tiles_c = n / tw; float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(tiles_r * tiles_c), dim3(256), 0, 0, d, ld, tiles_r); hipEventRecord(e1);
      hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms; }
    printf("%-40s %.3f ms  %.0f GB/s\n", name, best, n * n * 8.0 / best / 1e6);
  };
  run("A 64x64 8B stores (assembly)", pat_a, 64, 64);
  run("B 128x32 16B stores", pat_b, 128, 32);
  run("C 256x16 16B stores x2", pat_c, 256, 16);
  run("D 256x16 8B stores", pat_d, 256, 16);
  run("E 64x64 8B nontemporal", pat_e, 64, 64);
  run("F 64x64 8B column-fastest blocks", pat_f, 64, 64);
  return 0;
}
