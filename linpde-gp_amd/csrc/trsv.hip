// Single right-hand side: representer weights  w = L^{-T} L^{-1} r  -- `gram.solve(Y - Lm)` of every conditioning
// (_conditional.py:44,96-110; BlockMatrix2x2._solve, linops/_block.py:244-268) and the vector behind `u.mean(x)`.
//
// Round 6: ONE resident launch per direction instead of one dependent launch per 128-row tile (rounds 1-5: 2 x T launches of
// a scalar GEMV, ~30 us each: 8 ms at c3 against 0.36 ms for reading the factor once at the achievable HBM rate).
//
// Forward (L y = b).  Workgroup i (tickets: in the order the workgroups start) owns the 128 rows of tile row i.  It streams the
// tiles (i, 0), (i, 1), ... (i, i - 1) of its row strip through registers and multiplies tile (i, j) with y_j as soon as y_j
// has been PUBLISHED by workgroup j; the partial sums of a thread stay in its registers over the whole strip.  Linv_i and L_ii
// are in registers from the start: the workgroup forms  b_i - sum_j L_ij y_j,  solves against its diagonal tile with the explicit tile inverse and ONE step of
// iterative refinement against the tile itself (x0 = Linv b', x = x0 + Linv (b' - L x0): the arithmetic of tile_solve_kernel and
// of the step kernels this replaces), and publishes y_i.
// Backward (L^T x = y): workgroup k owns tile COLUMN i = T - 1 - k, streams (T - 1, i), ..., (i + 1, i), accumulates the
// transposed products per lane and reduces over the lanes once, at the end.
//
// Hand-over (MI355X_MICROARCH.md, inter-workgroup visibility; cdna_hip_programming.md Guideline 16, form R2 "the data IS the
// flag"): the solution vector is filled with a sentinel (all ones: a NaN no arithmetic produces) before the launch; every
// entry is published by ONE naturally aligned 8-byte agent-scope store and read by agent-scope 8-byte loads until it is no
// longer the sentinel -- no flag, no fence, no ordering between entries needed.  A workgroup only ever waits for workgroups
// with a SMALLER ticket, which are running or done: no assumption on residency or dispatch order.  Every poll is bounded
// (~1 s): a value that does not arrive becomes a NaN in the result and a negative status word, never a hung device.
//
// Critical path per tile row: one hand-over (~1.5-3 us) + one tile product + the refined tile solve (three products), all on
// operands already in registers.  Everything else -- 8 N^2 / 2 bytes of factor per direction -- streams underneath it.

#include <climits>
#include <cstdlib>
#include <type_traits>

#include "lpgp_internal.h"

namespace lpgp {

constexpr unsigned long long TRSV_SENT = ~0ull;
constexpr unsigned long long TRSV_QNAN = 0x7FF8000000000000ull;
constexpr int TRSV_SPIN_LIMIT = 1 << 21;

struct TrsvArgs {
  const double* L;         // the factor (lower triangle, diagonal tiles with zeros above the diagonal)
  int64_t ld;
  const double* linv;      // tile inverses, tile k at linv + k * 128 * 128
  const double* b;         // right-hand side (read only)
  double* x;               // solution, sentinel-filled
  int* ticket;             // starts at -1
  int* info;               // status word: INT_MIN if a hand-over timed out
  int32_t T;
  int32_t flags;           // measurement aids (LPGP_TRSV_FLAGS): 1 no refinement step (WRONG results beyond eps cond), 2 polls without s_sleep
};

__device__ __forceinline__ unsigned long long trsv_ld(const double* p) {
  return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void trsv_st(double* p, double v) {
  unsigned long long u = (unsigned long long)__double_as_longlong(v);
  if (u == TRSV_SENT) u = TRSV_QNAN;
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// `v`: the value of a first load issued earlier (so that it is in front of the tile loads in the memory queue)
__device__ __forceinline__ double trsv_poll(const double* p, unsigned long long v, int* info, bool nosleep = false) {
  int spins = 0;
  while (v == TRSV_SENT) {
    if (++spins > TRSV_SPIN_LIMIT) {
      atomicCAS(info, 0, INT_MIN);
      v = TRSV_QNAN;
      break;
    }
    if (!nosleep) __builtin_amdgcn_s_sleep(1);
    v = trsv_ld(p);
  }
  return __longlong_as_double((long long)v);
}

// One 128 x 128 tile in the registers of 512 threads, sixteen 16-byte loads each.  Every load is a BUFFER load: the tile's
// base in a descriptor (scalar registers), the column offset in the scalar offset, ONE 32-bit lane offset for all of them --
// with global loads hipcc folds the lane offset into the base first and keeps a 64-bit vector address per load, which the
// three tile buffers leave no room for.
typedef unsigned int trsv_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t trsv_rsrc(const double* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(base), 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ double2 trsv_ldb(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, int64_t col_doubles) {
  const trsv_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_bytes, (int)(col_doubles * 8), 0);
  double2 d;
  d.x = __longlong_as_double((long long)(((unsigned long long)u.y << 32) | u.x));
  d.y = __longlong_as_double((long long)(((unsigned long long)u.w << 32) | u.z));
  return d;
}
// "Row form" (lane = row pair, wave = 16 columns): M[c] = (tile(2 rp, 16 cg + c), tile(2 rp + 1, 16 cg + c)); a wave's load
// instruction reads 1 KB of one column.  base: element (0, 16 cg) of the tile (uniform); lane_bytes = 16 rp
__device__ __forceinline__ void trsv_load_rows(double2 (&M)[16], const double* base, int64_t ldm, unsigned lane_bytes) {
  const __amdgpu_buffer_rsrc_t r = trsv_rsrc(base);
#pragma unroll
  for (int c = 0; c < 16; ++c) M[c] = trsv_ldb(r, lane_bytes, (int64_t)c * ldm);
}
// "Column form" of the same tile for products with its TRANSPOSE (lane = column pair, wave = 16 rows):
// M[k] = (tile(16 cg + 2 k, 2 cl), tile(16 cg + 2 k + 1, 2 cl)), M[8 + k] the same of column 2 cl + 1: a lane reads one whole
// 128-byte line per column (diagonal tiles only: two per workgroup, loaded ahead of their use).
// base: element (16 cg, 0) of the tile (uniform); lane_bytes = 2 cl * ldm * 8
__device__ __forceinline__ void trsv_load_cols(double2 (&M)[16], const double* base, int64_t ldm, unsigned lane_bytes) {
  const __amdgpu_buffer_rsrc_t r = trsv_rsrc(base);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    M[k] = trsv_ldb(r, lane_bytes, 2 * k);
    M[8 + k] = trsv_ldb(r, lane_bytes, ldm + 2 * k);
  }
}
// row form: (a0 + a2, a1 + a3) += tile(rows 2 rp, 2 rp + 1; columns 16 cg ..) . v[16 cg ..]   (v: wave-uniform LDS reads)
__device__ __forceinline__ void trsv_fma_rows(const double2 (&M)[16], const double* v16, double& a0, double& a1, double& a2, double& a3) {
#pragma unroll
  for (int c = 0; c < 16; c += 2) {
    const double2 xv = *reinterpret_cast<const double2*>(v16 + c);
    a0 = fma(M[c].x, xv.x, a0);
    a1 = fma(M[c].y, xv.x, a1);
    a2 = fma(M[c + 1].x, xv.y, a2);
    a3 = fma(M[c + 1].y, xv.y, a3);
  }
}
// column form: (a0 + a2, a1 + a3) += tile(rows 16 cg ..; columns 2 cl, 2 cl + 1)^T . v[16 cg ..]
__device__ __forceinline__ void trsv_fma_cols(const double2 (&M)[16], const double* v16, double& a0, double& a1, double& a2, double& a3) {
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const double2 xv = *reinterpret_cast<const double2*>(v16 + 2 * k);
    a0 = fma(M[k].x, xv.x, a0);
    a2 = fma(M[k].y, xv.y, a2);
    a1 = fma(M[8 + k].x, xv.x, a1);
    a3 = fma(M[8 + k].y, xv.y, a3);
  }
}

// y = M v (TRANS: M^T v) for a tile held in row form (TRANS: column form): the per-wave partial sums meet in `part`
// (8 x 128 doubles of LDS); returns y[t] in the threads t < 128.  Two barriers; `v` must be visible (a barrier behind its last
// store), `part` free.
template <bool TRANS>
__device__ __forceinline__ double trsv_mv(const double2 (&M)[16], const double* v, double* part, int t) {
  const int ln = t & 63, cg = t >> 6;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (TRANS) trsv_fma_cols(M, v + 16 * cg, a0, a1, a2, a3);
  else trsv_fma_rows(M, v + 16 * cg, a0, a1, a2, a3);
  *reinterpret_cast<double2*>(part + cg * TILE + 2 * ln) = make_double2(a0 + a2, a1 + a3);
  __syncthreads();
  double s = 0.0;
  if (t < TILE) {
#pragma unroll
    for (int w = 0; w < 8; ++w) s += part[w * TILE + t];
  }
  return s;
}

// the refined solve against the diagonal tile: sb = right-hand side (LDS, visible), LI / LD = Linv_i / L_ii in the form TRANS
// asks for; returns x[t] in the threads t < 128
template <bool TRANS>
__device__ __forceinline__ double trsv_diag_solve(const double2 (&LI)[16], const double2 (&LD)[16], const double* sb, double* s1, double* s2,
                                                  double* part, int t, bool refine = true) {
  const double x0 = trsv_mv<TRANS>(LI, sb, part, t);
  if (!refine) return x0;
  if (t < TILE) s1[t] = x0;
  __syncthreads();
  const double lx = trsv_mv<TRANS>(LD, s1, part, t);
  if (t < TILE) s2[t] = sb[t] - lx;
  __syncthreads();
  const double dx = trsv_mv<TRANS>(LI, s2, part, t);
  return x0 + dx;
}

// Register plan (both kernels): Linv_i and L_ii are loaded FIRST and stay in registers (2 x 64); the row strip streams through
// ONE tile buffer (64), re-loaded right behind the product that consumed it.  So when the chain reaches a workgroup -- the poll
// for the last solution block it needs -- the workgroup has NO load in flight: the price of a hand-over sits in the consumer
// CU's own memory queue (MI355X_MICROARCH.md, handoff-1to1: 1.1 us between unloaded CUs, 2.9 between streaming ones; the first
// form of this kernel, which kept prefetching L_ii behind the poll, took 4.5 us per tile row, and neither the refinement's
// two products nor the poll's s_sleep showed in it).  The loop has no conditional around a memory instruction (the last tile is
// peeled): behind a conditional load hipcc falls back to `s_waitcnt vmcnt(0)` everywhere.  All 512 threads poll (four per
// entry): one instruction stream for the eight waves.
__global__ __launch_bounds__(512, 1) void trsv_fwd_resident_kernel(TrsvArgs g) {
  __shared__ __attribute__((aligned(16))) double sx[2][TILE];
  __shared__ __attribute__((aligned(16))) double part[8 * TILE];
  __shared__ __attribute__((aligned(16))) double sb[TILE], s1[TILE], s2[TILE];
  __shared__ int s_ticket;
  const int t = threadIdx.x, rp = t & 63, cg = __builtin_amdgcn_readfirstlane(t >> 6);
  if (t == 0) s_ticket = atomicAdd(g.ticket, 1) + 1;
  __syncthreads();
  const int i = __builtin_amdgcn_readfirstlane(s_ticket);
  if (i >= g.T) return;
  const unsigned lane_bytes = 16u * (unsigned)rp;
  double2 LI[16], LD[16], B[16];
  trsv_load_rows(LI, g.linv + (int64_t)i * TILE * TILE + (int64_t)(16 * cg) * TILE, TILE, lane_bytes);
  trsv_load_rows(LD, g.L + (int64_t)i * TILE * (g.ld + 1) + (int64_t)(16 * cg) * g.ld, g.ld, lane_bytes);
  const double bi = g.b[(int64_t)i * TILE + (t & 127)];
  const double* strip = g.L + (int64_t)i * TILE + (int64_t)(16 * cg) * g.ld;       // tile (i, j) at strip + j * 128 * ld
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int par = 0;
  // y_j arrives, tile (i, j) (in B) is multiplied with it
  auto consume = [&](int j) {
    const double* px = g.x + (int64_t)j * TILE + (t & 127);
    const double xv = trsv_poll(px, trsv_ld(px), g.info, (g.flags & 2) != 0);
    if (t < TILE) sx[par][t] = xv;
    __syncthreads();
    trsv_fma_rows(B, &sx[par][16 * cg], a0, a1, a2, a3);
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));      // (the products stay HERE, in front of the re-load of B)
    par ^= 1;
  };
  if (i > 0) {
    trsv_load_rows(B, strip, g.ld, lane_bytes);
    for (int j = 0; j < i - 1; ++j) {
      consume(j);
      trsv_load_rows(B, strip + (int64_t)(j + 1) * TILE * g.ld, g.ld, lane_bytes);
    }
    consume(i - 1);
  }
  *reinterpret_cast<double2*>(part + cg * TILE + 2 * rp) = make_double2(a0 + a2, a1 + a3);
  __syncthreads();
  if (t < TILE) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += part[w * TILE + t];
    sb[t] = bi - s;
  }
  __syncthreads();
  const double xi = trsv_diag_solve<false>(LI, LD, sb, s1, s2, part, t, !(g.flags & 1));
  if (t < TILE) trsv_st(g.x + (int64_t)i * TILE + t, xi);
}

__global__ __launch_bounds__(512, 1) void trsv_bwd_resident_kernel(TrsvArgs g) {
  __shared__ __attribute__((aligned(16))) double sx[2][TILE];
  __shared__ __attribute__((aligned(16))) double part[8 * TILE];
  __shared__ __attribute__((aligned(16))) double sb[TILE], s1[TILE], s2[TILE];
  __shared__ __attribute__((aligned(16))) double red[TILE * 32];
  __shared__ int s_ticket;
  const int t = threadIdx.x, ln = t & 63, cg = __builtin_amdgcn_readfirstlane(t >> 6);
  const int rq = ln & 31, ch = ln >> 5;          // bulk tiles: lane = (rq, column half ch): rows 2 rq, 2 rq + 1, 64 + 2 rq, 65 + 2 rq; columns 16 cg + 8 ch + c
  if (t == 0) s_ticket = atomicAdd(g.ticket, 1) + 1;
  __syncthreads();
  const int k = __builtin_amdgcn_readfirstlane(s_ticket);
  if (k >= g.T) return;
  const int i = g.T - 1 - k;                     // this workgroup's tile column
  const unsigned lane_bulk = 16u * (unsigned)rq + (unsigned)ch * (unsigned)(8 * g.ld * 8);
  const unsigned lane_linv = (unsigned)(2 * ln) * (unsigned)(TILE * 8), lane_diag = (unsigned)(2 * ln) * (unsigned)(g.ld * 8);
  double2 LI[16], LD[16], B[16];
  trsv_load_cols(LI, g.linv + (int64_t)i * TILE * TILE + 16 * cg, TILE, lane_linv);
  trsv_load_cols(LD, g.L + (int64_t)i * TILE * (g.ld + 1) + 16 * cg, g.ld, lane_diag);
  const double yi = g.b[(int64_t)i * TILE + (t & 127)];
  const double* strip = g.L + ((int64_t)i * TILE + 16 * cg) * g.ld;       // tile (j, i) at strip + j * 128
  auto load_bulk = [&](int j) {
    const __amdgpu_buffer_rsrc_t r = trsv_rsrc(strip + (int64_t)j * TILE);
#pragma unroll
    for (int c = 0; c < 8; ++c) {          // (a load instruction reads 512 contiguous bytes of each of two columns)
      B[2 * c] = trsv_ldb(r, lane_bulk, (int64_t)c * g.ld);
      B[2 * c + 1] = trsv_ldb(r, lane_bulk, (int64_t)c * g.ld + 64);
    }
  };
  double acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.0;
  int par = 0;
  auto consume = [&](int j) {
    const double* px = g.x + (int64_t)j * TILE + (t & 127);
    const double xv = trsv_poll(px, trsv_ld(px), g.info, (g.flags & 2) != 0);
    if (t < TILE) sx[par][t] = xv;
    __syncthreads();
    const double2 xa = *reinterpret_cast<const double2*>(&sx[par][2 * rq]);
    const double2 xb = *reinterpret_cast<const double2*>(&sx[par][64 + 2 * rq]);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      double s = fma(B[2 * c].x, xa.x, acc[c]);
      s = fma(B[2 * c].y, xa.y, s);
      s = fma(B[2 * c + 1].x, xb.x, s);
      acc[c] = fma(B[2 * c + 1].y, xb.y, s);
    }
    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]));
    par ^= 1;
  };
  if (k > 0) {
    load_bulk(g.T - 1);
    for (int j = g.T - 1; j > i + 1; --j) {
      consume(j);
      load_bulk(j - 1);
    }
    consume(i + 1);
  }
  // sum over the 32 lanes rq: red[column][rq], then four lanes per column
#pragma unroll
  for (int c = 0; c < 8; ++c) red[(16 * cg + 8 * ch + c) * 32 + rq] = acc[c];
  __syncthreads();
  {
    const int col = t >> 2, qt = t & 3;
    const double* r8 = red + col * 32 + 8 * qt;
    const double2 u0 = *reinterpret_cast<const double2*>(r8), u1 = *reinterpret_cast<const double2*>(r8 + 2);
    const double2 u2 = *reinterpret_cast<const double2*>(r8 + 4), u3 = *reinterpret_cast<const double2*>(r8 + 6);
    double s = ((u0.x + u0.y) + (u1.x + u1.y)) + ((u2.x + u2.y) + (u3.x + u3.y));
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (qt == 0) part[col] = s;
  }
  __syncthreads();
  if (t < TILE) sb[t] = yi - part[t];
  __syncthreads();
  const double xi = trsv_diag_solve<true>(LI, LD, sb, s1, s2, part, t, !(g.flags & 1));
  if (t < TILE) trsv_st(g.x + (int64_t)i * TILE + t, xi);
}

// v (padded length T * 128, device) <- G^{-1} v on the panel stream; tmp: T * 128 + 2 doubles of scratch (the intermediate
// vector and the two ticket words).  `info`: device status word (zeroed by the caller), INT_MIN after a timed-out hand-over.
int solve_vec_resident(lpgp_ctx* ctx, lpgp_mat* mat, int64_t T64, double* v, double* tmp, int* info) {
  const int T = (int)T64;
  hipStream_t st = ctx->s_main;
  const size_t nb = (size_t)T * TILE * sizeof(double);
  TrsvArgs g;
  g.L = mat->a; g.ld = mat->cap; g.linv = mat->linv; g.info = info; g.T = T;
  static const int dbg_flags = [] { const char* e = std::getenv("LPGP_TRSV_FLAGS"); return e ? std::atoi(e) : 0; }();
  g.flags = dbg_flags;
  // forward: L tmp = v
  LPGP_HIP(hipMemsetAsync(tmp, 0xFF, nb + 2 * sizeof(double), st));           // the sentinel, and both tickets at -1
  g.b = v; g.x = tmp; g.ticket = reinterpret_cast<int*>(tmp + (size_t)T * TILE);
  prof_begin(ctx, st, LPGP_K_TRSM, (double)T * TILE * (double)T * TILE, 4.0 * (double)T * TILE * (double)T * TILE);
  hipLaunchKernelGGL(trsv_fwd_resident_kernel, dim3((unsigned)T), dim3(512), 0, st, g);
  prof_end(ctx, st);
  // backward: L^T v = tmp
  LPGP_HIP(hipMemsetAsync(v, 0xFF, nb, st));
  g.b = tmp; g.x = v; g.ticket = reinterpret_cast<int*>(tmp + (size_t)T * TILE + 1);
  prof_begin(ctx, st, LPGP_K_TRSM, (double)T * TILE * (double)T * TILE, 4.0 * (double)T * TILE * (double)T * TILE);
  hipLaunchKernelGGL(trsv_bwd_resident_kernel, dim3((unsigned)T), dim3(512), 0, st, g);
  prof_end(ctx, st);
  LPGP_HIP(hipGetLastError());
  return 0;
}

}  // namespace lpgp
