// Latency-bound triangular solves against tiles / panels of the factor on fp64 MFMA: the refined tile solve of the
// panel chain and of the forward substitution, and the fused panel step of the forward substitution.  (A translation
// unit of its own: the fully unrolled stage schedules take a minute to compile.)
#include "lpgp_internal.h"
#include "kernel_util.h"
#include "solve_panel.h"

namespace lpgp {

// =========================================================================================
// Tile solve with ONE STEP OF ITERATIVE REFINEMENT, in place -- the latency-bound second kernel of
// every step of the panel chain and the tile step of the multi-RHS forward substitution:
//   KFAST = false:  X[rows, 0:128] <- X L^{-T}            X(i, c) at P[i + c ld]   (panel rows below a factored tile)
//   KFAST = true :  V[0:128, cols] <- L^{-1} V, as X = V^T:  X(i, c) at P[c + i ld]   (i = right-hand-side column)
// with L the 128 x 128 lower-triangular diagonal tile of the factor and Linv its explicit inverse:
//     X0 = A Linv^T;    R = A - X0 L^T;    X = X0 + R Linv^T.
// A product with the explicit inverse alone has a backward error of cond(L) eps (Linv L = I + E,
// |E| ~ cond(L) eps): measured at c3 (cond of the diagonal tiles up to 3e5) the posterior mean was 1.8e-8
// away from LAPACK's, 65x LAPACK's own distance from the long-double-refined solution
// (scratch/parity_diag.py, scratch/tileinv_accuracy.py).  The refinement step squares E away: what is
// left is the rounding of R, i.e. the backward error of a substitution -- at three latency-bound tile
// products instead of one and no dependent chain of 128 steps.
//
// A workgroup of eight waves owns 32 rows: two groups of 16 rows x four waves; wave `cc` of a group owns
// the output fragments u = cc, cc + 4, ..., cc + 28 (fragment u = columns 4u .. 4u + 3 of the 16 rows, in
// the accumulator layout of v_mfma_f64_4x4x4_4b_f64: lane -> (row = lane & 15, column 4u + (lane >> 4))).
// That layout IS the instruction's m-side operand layout for k-step u, so the result fragments of one
// product are the operand fragments of the next: they are exchanged between the four waves of a group
// through a 16-KB LDS image `xa[group][fragment][lane]`, never reshuffled.  The triangular factor is
// streamed by LDS-DMA in 24 stages of 16 k-rows (Linv^T, L^T, Linv^T) through a statically scheduled circular
// buffer (below).  Fragment u needs k-steps g <= u only (lower triangle): stage kt feeds the fragments u >= 4 kt,
// 144 MFMAs per wave and product instead of 256; the fragments of a stage's own diagonal block meet the
// explicit zeros above the diagonal of Linv / L.  The two groups take their classes in opposite order, so
// every SIMD holds a wave with 4 cc and one with 4 (3 - cc) columns beyond the stage's diagonal.
// LDS: 32 KB + 48 KB = 80 KB -- a solve workgroup fits beside ONE 73-KB GEMM workgroup, and two fit on a free CU.
// =========================================================================================

// Stage s = (product s / 8, k-rows 16 kt .. 16 kt + 15, kt = s % 8) holds only the columns the lower triangle
// needs, c >= 16 kt: 16 rows of 128 - 16 kt doubles at a row stride of 136 - 16 kt (2 * stride % 64 is 16 or 48:
// the four k-rows of a replicated n-side fragment read fall into four different bank groups).  The stages are
// placed in ONE circular buffer by a schedule computed at compile time (the whole stage loop is unrolled):
// at the top of stage s, right behind its barrier, every following stage that fits beside the stages still in
// use is issued.  The kernel is bound by the latency of these loads, not by their bytes or the matrix work
// (measured with three fixed 17-KB slots, two in flight: 18 us for one wave of workgroups against 6 us of
// MFMA time): what counts is bytes in flight per byte to fetch, and the trimmed stages put 4-5 of them in
// flight in the same 52 KB.  (Measured after that, scratch/tile_solve_time.py with diagnostic builds: 16 us for one
// wave of workgroups, 14 us with the factor loads removed altogether: what is left is the matrix work itself,
// 3 x 288 MFMAs per SIMD = 6.8 us, plus launch, the loads of A and 26 barriers.)
// Round 4: the tile solve's own ring is 48 KB (6144 doubles; the fused panel kernels keep TSV_RING = 52 KB), so that the
// workgroup takes 80 KB of LDS and TWO fit on a CU: on the CUs the narrow update stream leaves to the panel chain (64 in
// the chain-bound regime: all of c2, the last third of c3) the rows below a tile take two rounds of workgroups instead
// of three or four.
constexpr int TSV_RING_TILE = 6144;
constexpr int tsv_stride(int s) { return 136 - 16 * (s % 8); }
constexpr int tsv_size(int s) { return 16 * tsv_stride(s); }
struct TsvSched {
  int off[TSV_NSTAGE] = {};        // ring offset of stage s (doubles)
  int iss_lo[TSV_NSTAGE + 1] = {}; // stages [iss_lo[t], iss_hi[t]) are issued at time t: t = 0 before the loop,
  int iss_hi[TSV_NSTAGE + 1] = {}; // t = s + 1 at the top of stage s (behind its barrier)
  int wait[TSV_NSTAGE] = {};       // vmcnt at the top of stage s: this wave's DMA instructions of later stages in flight
};
constexpr TsvSched tsv_make_sched() {
  TsvSched S;
  int next = 0, head = 0;
  for (int t = 0; t <= TSV_NSTAGE; ++t) {
    const int live_lo = t == 0 ? 0 : t - 1;          // stages >= live_lo are in use or in flight
    S.iss_lo[t] = next;
    while (next < TSV_NSTAGE) {
      const int sz = tsv_size(next);
      int o = head;
      if (o + sz > TSV_RING_TILE) o = 0;
      bool ok = true;
      for (int l = live_lo; l < next; ++l)
        if (o < S.off[l] + tsv_size(l) && S.off[l] < o + sz) ok = false;
      if (!ok) break;
      S.off[next] = o;
      head = o + sz;
      ++next;
    }
    S.iss_hi[t] = next;
    if (t < TSV_NSTAGE) {
      // stage t must have been issued by now (the ring holds any single stage)
      S.wait[t] = 2 * (next - (t + 1));
    }
  }
  return S;
}
constexpr TsvSched TSV_SCHED = tsv_make_sched();
static_assert(TSV_SCHED.iss_hi[TSV_NSTAGE] == TSV_NSTAGE, "tile solve: a stage was never issued");
constexpr bool tsv_sched_ok() {
  for (int s = 0; s < TSV_NSTAGE; ++s)
    if (TSV_SCHED.iss_hi[s] < s + 1 || TSV_SCHED.wait[s] < 0 || TSV_SCHED.wait[s] > 62) return false;   // issued before it is awaited
  return true;
}
static_assert(tsv_sched_ok(), "tile solve: broken stage schedule");

struct TileSolveArgs {
  double* P;
  int64_t ld;
  const double* linv;                        // 128 x 128, column-major, ld 128, zeros above the diagonal
  const double* L;                           // 128 x 128 diagonal tile of the factor, zeros above the diagonal
  int64_t ldl;
};



template <bool KFAST>
__global__ __launch_bounds__(512, 2) void tile_solve_kernel(TileSolveArgs g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* xa = smem;
  double* ring = smem + TSV_XA;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wu = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wu >> 2, s4 = wu & 3;
  const int cc = rg ? 3 - s4 : s4;
  const int li = lane & 15, lj = lane >> 4;
  const int64_t i0 = (int64_t)blockIdx.x * TSV_ROWS + rg * 16 + li;      // this lane's row
  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) double*)smem);

  // stage s of the factor stream: product s / 8 (0, 2: Linv^T; 1: L^T); element (k, c) of M^T is M[c + k ldm]:
  // one k-row from column 16 kt on = up to 128 contiguous doubles = one DMA wave instruction (lanes beyond the
  // row are masked off), two rows per wave
  auto issue = [&](auto S_) {
    constexpr int s = decltype(S_)::value;
    constexpr int p = s / 8, kt = s % 8, len = 128 - 16 * kt, stride = tsv_stride(s);
    const double* M = (p == 1) ? g.L : g.linv;
    const int64_t ldm = (p == 1) ? g.ldl : (int64_t)TILE;
    double* sb = ring + TSV_SCHED.off[s];
    if (2 * lane < len) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = 2 * wu + h;
        const char* ub = reinterpret_cast<const char*>(M + 16 * kt + ((int64_t)kt * 16 + r) * ldm);
        __builtin_amdgcn_global_load_lds((gptr_t)(ub + (unsigned)lane * 16u), (lptr_t)(sb + r * stride), 16, 0, 0);
      }
    }
  };
  // own fragments of A (fragment u = cc + 4 q: element (row, column 4u + lj)); loaded BEFORE the first factor
  // stages are requested: vector-memory operations return in order, so the wait for these fragments must not
  // include the stages
  double a[8], x[8];
  double* const pbase = KFAST ? g.P + i0 * g.ld + (4 * cc + lj) : g.P + i0 + (int64_t)(4 * cc + lj) * g.ld;
  const int64_t pstep = KFAST ? 16 : 16 * g.ld;                            // fragment q -> q + 1: 16 columns on
#pragma unroll
  for (int q = 0; q < 8; ++q) a[q] = pbase[q * pstep];
  asm volatile("" ::: "memory");
  issue(std::integral_constant<int, 0>{});        // full rows: no lane mask, no branch
  double* const xown = xa + (size_t)(rg * 32 + cc) * 64 + lane;            // fragment q of this wave: xown[q * 256]
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    xown[q * 256] = a[q];
    x[q] = 0.0;
  }
  // (behind the use of a[]: hipcc waits vmcnt(0) at the first use of a loaded value that follows a lane-masked
  //  branch, i.e. it would wait for these stages too)
  asm volatile("" ::: "memory");
  static_for<1, TSV_SCHED.iss_hi[0]>(issue);
  const unsigned mlane = lds_base + 8u * (unsigned)(rg * 2048 + lane);
  const unsigned nlane = lds_base + 8u * (unsigned)(TSV_XA + lj * 136 + (lane & 3) + 4 * cc);
  const unsigned lj128 = (unsigned)lj * 128u;                              // bytes a row stride shrinks per kt, times lj

  // dst[q] += sum over k-steps g <= u of (operand fragment g from xa) x (factor fragment (g, u)), u = cc + 4 q
  auto run_product = [&](auto P_, double(&dst)[8]) {
    constexpr int prod = decltype(P_)::value;
    static_for<0, 8>([&](auto KT_) {
      constexpr int kt = decltype(KT_)::value;
      constexpr int s = prod * 8 + kt;
      constexpr int stride = tsv_stride(s);
      vm_wait_n<TSV_SCHED.wait[s]>();      // this wave's DMA pieces of stage s have landed ...
      TSV_BARRIER();                     // ... everybody's have, stage s - 1 is consumed, xa of this product is written
      static_for<TSV_SCHED.iss_lo[s + 1], TSV_SCHED.iss_hi[s + 1]>(issue);
      const unsigned aN = nlane + (unsigned)TSV_SCHED.off[s] * 8u - (unsigned)kt * lj128;
      double mf[2], nf[2][8];
      asm volatile("" ::: "memory");
      mf[0] = lds_read_async<(4 * kt) * 64>(mlane);
      static_for<kt, 8>([&](auto Q_) {
        constexpr int q = decltype(Q_)::value;
        nf[0][q] = lds_read_async<16 * (q - kt)>(aN);
      });
      static_for<0, 4>([&](auto K_) {
        constexpr int ks = decltype(K_)::value;
        if constexpr (ks + 1 < 4) {
          mf[(ks + 1) & 1] = lds_read_async<(4 * kt + ks + 1) * 64>(mlane);
          static_for<kt, 8>([&](auto Q_) {
            constexpr int q = decltype(Q_)::value;
            nf[(ks + 1) & 1][q] = lds_read_async<(ks + 1) * 4 * stride + 16 * (q - kt)>(aN);
          });
          lds_wait_n<9 - kt>();
        } else {
          lds_wait_n<0>();
        }
        static_for<kt, 8>([&](auto Q_) {
          constexpr int q = decltype(Q_)::value;
          dst[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(nf[ks & 1][q], mf[ks & 1], dst[q], 0, 0, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  };

  run_product(std::integral_constant<int, 0>{}, x);          // x = X0 = A Linv^T
  TSV_BARRIER();                                            // nobody reads the fragments of A any more
#pragma unroll
  for (int q = 0; q < 8; ++q) xown[q * 256] = -x[q];
  run_product(std::integral_constant<int, 1>{}, a);          // a = R = A - X0 L^T
  TSV_BARRIER();
#pragma unroll
  for (int q = 0; q < 8; ++q) xown[q * 256] = a[q];
  run_product(std::integral_constant<int, 2>{}, x);          // x = X0 + R Linv^T
  // in place: this workgroup read exactly the 32 rows it overwrites, all of them before the first barrier
#pragma unroll
  for (int q = 0; q < 8; ++q) pbase[q * pstep] = x[q];
}

int launch_panel_solve_4(lpgp_ctx* ctx, hipStream_t stream, const PanelSolveArgs& a, int64_t cols, bool kfast);    // solve4.hip

template <bool KFAST>
static int launch_panel_any(lpgp_ctx* ctx, hipStream_t stream, double* V, int64_t ldv, const double* linv, const double* L, int64_t ldl,
                            int nt_panel, int64_t count, int prof_kernel) {
  if (count <= 0 || nt_panel <= 0) return 0;
  LPGP_CHECK(nt_panel <= 4, "panel solve: at most 4 tiles per panel (got %d)", nt_panel);
  PanelSolveArgs a;
  a.V = V; a.ldv = ldv; a.linv = linv; a.L = L; a.ldl = ldl;
  // algorithmic flops of the triangular solve of the panel: count x (nt_panel * 128)^2
  if (prof_kernel >= 0) prof_begin(ctx, stream, prof_kernel, (double)count * (double)(nt_panel * TILE) * (double)(nt_panel * TILE), 0.0);
  int rc;
  switch (nt_panel) {
    case 1: rc = launch_panel_solve_nt<1, KFAST>(ctx, stream, a, count); break;
    case 2: rc = launch_panel_solve_nt<2, KFAST>(ctx, stream, a, count); break;
    case 3: rc = launch_panel_solve_nt<3, KFAST>(ctx, stream, a, count); break;
    default: rc = launch_panel_solve_4(ctx, stream, a, count, KFAST); break;
  }
  if (prof_kernel >= 0) prof_end(ctx, stream);
  return rc;
}

// V (nt_rows <= 4 tiles of rows x nt_cols * 128 columns, column-major ldv) <- L_KK^{-1} V in place, L_KK the panel's diagonal block
int launch_trsv_panel(lpgp_ctx* ctx, hipStream_t stream, double* V, int64_t ldv, const double* linv, const double* L, int64_t ldl,
                      int nt_rows, int nt_cols, int prof_kernel) {
  return launch_panel_any<true>(ctx, stream, V, ldv, linv, L, ldl, nt_rows, (int64_t)nt_cols * TILE, prof_kernel);
}

// X (mt * 128 rows x nt_cols <= 4 tiles of columns, column-major ldx) <- X L_KK^{-T} in place: the rows below (or appended
// to) an already factored diagonal block
int launch_trsm_panel(lpgp_ctx* ctx, hipStream_t stream, double* X, int64_t ldx, const double* linv, const double* L, int64_t ldl,
                      int nt_cols, int mt, int prof_kernel) {
  return launch_panel_any<false>(ctx, stream, X, ldx, linv, L, ldl, nt_cols, (int64_t)mt * TILE, prof_kernel);
}

template <bool KFAST>
static int launch_tile_solve(lpgp_ctx* ctx, hipStream_t stream, double* P, int64_t ld, const double* linv, const double* L,
                             int64_t ldl, int64_t rows, int prof_kernel) {
  if (rows <= 0) return 0;
  const size_t shmem = (size_t)(TSV_XA + TSV_RING_TILE) * sizeof(double);   // 81 920 B: two workgroups per CU
  LPGP_TRY_RC(ensure_lds_attr(ctx, reinterpret_cast<const void*>(&tile_solve_kernel<KFAST>), shmem));
  TileSolveArgs a;
  a.P = P; a.ld = ld; a.linv = linv; a.L = L; a.ldl = ldl;
  // algorithmic flops: the triangular solve itself (rows x 128^2), not the three products that form it
  if (prof_kernel >= 0) prof_begin(ctx, stream, prof_kernel, (double)rows * TILE * TILE, 0.0);
  hipLaunchKernelGGL(tile_solve_kernel<KFAST>, dim3((unsigned)(rows / TSV_ROWS)), dim3(512), shmem, stream, a);
  if (prof_kernel >= 0) prof_end(ctx, stream);
  LPGP_HIP(hipGetLastError());
  return 0;
}

// X (mt*128 rows x 128, column-major ldx) <- X L^{-T} in place
int launch_trsm_tile(lpgp_ctx* ctx, hipStream_t stream, double* X, int64_t ldx, const double* linv, const double* L, int64_t ldl,
                     int mt, int prof_kernel) {
  return launch_tile_solve<false>(ctx, stream, X, ldx, linv, L, ldl, (int64_t)mt * TILE, prof_kernel);
}

// V (128 rows x nt*128 columns, column-major ldv) <- L^{-1} V in place
int launch_trsv_tile(lpgp_ctx* ctx, hipStream_t stream, double* V, int64_t ldv, const double* linv, const double* L, int64_t ldl,
                     int nt, int prof_kernel) {
  return launch_tile_solve<true>(ctx, stream, V, ldv, linv, L, ldl, (int64_t)nt * TILE, prof_kernel);
}

}  // namespace lpgp
