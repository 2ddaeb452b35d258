// fp64 MFMA GEMM / SYRK for the Cholesky trailing update, the panel triangular solves
// (as products with explicit 128x128 inverses) and the multi-RHS forward substitution.
//
// Replaces LAPACK dpotrf's trailing DSYRK/DGEMM and dpotrs/dtrtrs's DTRSM that the
// reference reaches through probnum `LinearOperator.cholesky/solve`
// (_conditional.py:44,108,228; linops/_block.py:192-207,233-251).
//
// Design (gfx950): 128x128 block tile, 256 threads = 2x2 waves, 64x64 per wave as 4x4
// v_mfma_f64_16x16x4_f64 accumulators (128 VGPRs), K staged 16 deep through LDS with a
// register prefetch (one barrier per stage).  The MFMA operands are SWAPPED (B fragment as
// the instruction's A operand) so that every accumulator register holds 16 consecutive
// rows of one column of C: C is read and written in 128-byte row segments.  LDS images:
// [k][144] for an operand whose non-contracted index is fastest in memory, [idx][18] for a
// k-fastest operand; both are conflict-free for the one-ds_read_b64-per-fragment pattern.
// Block ids are grouped into 8x8 super-tiles and dealt so that one XCD works on one
// super-tile at a time (its 16 operand panels stay in that XCD's L2).

#include "lpgp_internal.h"

namespace lpgp {

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDM = 144;                 // [k][LDM] image (LDM % 32 == 16)
constexpr int STAGE = BK * LDM;          // doubles per operand per stage (K image uses 128*16 of it)

// ---- staging: global -> LDS by LDS-DMA (global_load_lds_dwordx4) -------------------------
// One wave-instruction moves 64 x 16 B = 1 KiB to a wave-uniform LDS base + lane*16.
//  * operand with its non-contracted index fastest in memory ("M image"): LDS image
//    [k][LDM]; one piece = one k-row of 128 doubles (the row padding LDM-128 is never
//    crossed by a piece);
//  * k-fastest operand ("K image"): LDS image [idx][16] without padding; one piece = 8 rows
//    of 128 B.  Bank conflicts are avoided by an XOR swizzle of the 16-byte chunks,
//    chunk' = chunk ^ ((idx >> 1) & 7), applied on the per-lane SOURCE address (the LDS
//    side of a DMA is always linear) and again on every fragment read.
// Measured reason for DMA instead of register staging: a stage's global loads have
// ~3000 cycles of latency under load and hipcc sinks register loads next to their ds_write
// (exposing that latency every stage); DMA writes LDS, so it cannot be moved below the
// stage's first ds_read and stays at the top of the stage.
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <bool T>
__device__ __forceinline__ void dma_tile(const double* __restrict__ P, int64_t ld, int64_t idx0, int64_t k0,
                                         int lane, int w, double* sdst) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int piece = q * 4 + w;                         // wave-uniform
    if constexpr (!T) {
      const double* src = P + (idx0 + 2 * lane) + (k0 + piece) * ld;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sdst + piece * LDM), 16, 0, 0);
    } else {
      const int r = piece * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      const double* src = P + (k0 + 2 * c) + (idx0 + r) * ld;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sdst + piece * 128), 16, 0, 0);
    }
  }
}

// Fragment element addresses (in doubles, relative to the stage base).
//   m-side fragment:  idx = idx_base + (lane & 15), k = ks*4 + (lane >> 4)
//   n-side fragment:  idx = idx_base + (lane & 3),  k = ks*4 + (lane >> 4)   (replicated over lane bits 2..3:
//                     the "A" operand of v_mfma_f64_4x4x4_4b_f64 with one 4x4 block shared by all four blocks)
template <bool T>
__device__ __forceinline__ int frag_off(int idx, int k) {
  if constexpr (!T) return k * LDM + idx;
  else              return idx * 16 + (((k >> 1) ^ ((idx >> 1) & 7)) << 1) + (k & 1);
}

// LDS read whose completion the COMPILER does not track: with LDS-DMA in flight hipcc turns
// every wait for a ds_read result into s_waitcnt lgkmcnt(0), which also waits for the
// prefetch just issued for the next chunk.  The reads are therefore issued from inline asm
// and retired by hand-counted s_waitcnt lgkmcnt(N) (LDS operations return in order).
__device__ __forceinline__ double lds_read_async(unsigned byte_addr) {
  double d;
  asm volatile("ds_read_b64 %0, %1" : "=v"(d) : "v"(byte_addr));
  return d;
}
#define LDS_WAIT(N) do { asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// MFMA shape: measured on MI355X, v_mfma_f64_16x16x4_f64 sustains only ~48 TFLOP/s chip-wide
// (>= 88 cycles per instruction per SIMD at any occupancy) while v_mfma_f64_4x4x4_4b_f64
// issues every 16.5 cycles = 75 TFLOP/s already at one wave per SIMD (scratch/mfma_probe.hip,
// DESIGN.md section 5).  The 64x64 wave tile is therefore built from 4x4x4_4b instructions:
// per k-step of 4, acc[t][u] (m = 16t + (lane&15), n = 4u + (lane>>4)) += Bfrag4[u] x Afrag[t],
// where the instruction's four 4x4 blocks share the 4 n-columns (replicated "A" operand) and
// cover 16 consecutive rows m ("B" operand) -- the same 128-byte-segment C layout as before.
// TRI: lower-triangular output (symmetric rank-k update): a distinct instantiation so that the
// SYRK launches show up under their own kernel symbol in rocprofv3 --stats.
template <bool TA, bool TB, bool TRI>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  // ---- block -> tile (XCD-aware 8x8 super-tiles) ----
  // Blocks b, b+8, b+16, ... share an XCD (round-robin dispatch): each run of 64 of them
  // is one 8x8 super-tile.  For the triangular case only super-tiles on or below the
  // diagonal are enumerated, so every XCD gets the same number of them.
  // S x S super-tiles, S = 1 << g.sshift in {8, 4, 2}: 8 for large grids (most L2 reuse), smaller
  // when there would be too few super-tiles to balance the 8 XCDs (measured with S = 8 only:
  // a 32x32-tile SYRK ran two rounds on two XCDs and one on the others, 27 instead of ~50 TFLOP/s).
  const int sh = g.sshift, S = 1 << sh, SS = S * S;
  const int SR = (g.mt + S - 1) >> sh, SC = (g.nt + S - 1) >> sh;
  const int b = blockIdx.x;
  const int xcd = b & 7, q = b >> 3;
  const int s = (q / SS) * 8 + xcd, inner = q % SS;
  int sr, sc;
  if (TRI) {
    const int ntri = SC * (SC + 1) / 2;             // super-tiles of the leading SC x SC triangle
    if (s < ntri) {
      sr = (int)((sqrtf(8.0f * (float)s + 1.0f) - 1.0f) * 0.5f);
      while ((sr + 1) * (sr + 2) / 2 <= s) ++sr;
      while (sr * (sr + 1) / 2 > s) --sr;
      sc = s - sr * (sr + 1) / 2;
    } else {
      const int s2 = s - ntri;
      sr = SC + s2 / SC;
      sc = s2 % SC;
    }
  } else {
    sr = s % SR;
    sc = s / SR;
  }
  if (sr >= SR || sc >= SC) return;
  const int tr = (sr << sh) + (inner & (S - 1)), tc = (sc << sh) + (inner >> sh);
  if (tr >= g.mt || tc >= g.nt) return;
  if (TRI && tr < tc) return;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid & 1, wn = wid >> 1;
  const int64_t m0 = (int64_t)tr * BM, n0 = (int64_t)tc * BN;

  double* sA = smem;                  // 2 stages
  double* sB = smem + 2 * STAGE;
  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) double*)smem);

  const int KT = g.k / BK;
  const int wu = __builtin_amdgcn_readfirstlane(wid);    // wave index as a scalar (LDS-DMA base must be uniform)
  dma_tile<TA>(g.A, g.lda, m0, 0, lane, wu, sA);
  dma_tile<TB>(g.B, g.ldb, n0, 0, lane, wu, sB);

  // The accumulators start from (beta/alpha) * C, so that the epilogue is stores only: the 64
  // C loads per lane are all in flight at once, under the first tile's DMA latency, instead
  // of 16 dependent load->fma->store round trips at the end of every tile (measured: ~33 us
  // of a 165 us tile at K = 512).
  const double alpha = g.alpha, beta = g.beta;
  double* const cbase = g.C + (n0 + wn * 64 + (lane >> 4)) * g.ldc + m0 + wm * 64 + (lane & 15);
  double acc[4][16];
  if (beta != 0.0) {
    const double sc = beta / alpha;
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t][u] = sc * cbase[(int64_t)(4 * u) * g.ldc + t * 16];
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[t][u] = 0.0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

#ifdef LPGP_STAMP
  unsigned long long st_load = 0, st_mfma = 0, st_store = 0, st_bar = 0, t_a, t_b;
#define STAMP(var) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); } while (0)
#else
#define STAMP(var) do { } while (0)
#endif
  int cur = 0;
  for (int kt = 0; kt < KT; ++kt) {
    const bool more = (kt + 1 < KT);
#ifdef LPGP_STAMP
    STAMP(t_a);
#endif
    // next stage's tiles -> other LDS buffer (unconditional: the last stage harmlessly
    // re-loads its own k-tile; a branch around the DMA would make hipcc drain vmcnt early)
#ifdef LPGP_EXPERIMENT
    if (!(g.ktrim & 1))
#endif
    {
      const int64_t knext = (int64_t)(more ? kt + 1 : kt) * BK;
      dma_tile<TA>(g.A, g.lda, m0, knext, lane, wu, sA + (cur ^ 1) * STAGE);
      dma_tile<TB>(g.B, g.ldb, n0, knext, lane, wu, sB + (cur ^ 1) * STAGE);
    }
    // 16 chunks of 16 MFMAs per stage (4 k-steps x 4 groups of 4 n-fragments).  Fragments
    // are double-buffered in registers: the LDS reads of chunk c+1 are issued before the
    // MFMAs of chunk c (264 cycles of matrix work cover the LDS latency) and retired with a
    // counted wait that leaves exactly those newer reads in flight.
    const unsigned baseA = lds_base + (unsigned)(cur * STAGE) * 8u;
    const unsigned baseB = lds_base + (unsigned)((2 + cur) * STAGE) * 8u;
    double am[2][4], bn[2][4];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < 4; ++t)
      am[0][t] = lds_read_async(baseA + 8u * (unsigned)frag_off<TA>(wm * 64 + t * 16 + (lane & 15), (lane >> 4)));
#pragma unroll
    for (int v = 0; v < 4; ++v)
      bn[0][v] = lds_read_async(baseB + 8u * (unsigned)frag_off<TB>(wn * 64 + v * 4 + (lane & 3), (lane >> 4)));
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int ks = c >> 2, uc = c & 3, cb = c & 1;
      if (c + 1 < 16) {
        const int ks2 = (c + 1) >> 2, uc2 = (c + 1) & 3;
#pragma unroll
        for (int v = 0; v < 4; ++v)
          bn[cb ^ 1][v] = lds_read_async(
              baseB + 8u * (unsigned)frag_off<TB>(wn * 64 + (uc2 * 4 + v) * 4 + (lane & 3), ks2 * 4 + (lane >> 4)));
        if (uc2 == 0) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            am[ks2 & 1][t] = lds_read_async(
                baseA + 8u * (unsigned)frag_off<TA>(wm * 64 + t * 16 + (lane & 15), ks2 * 4 + (lane >> 4)));
          LDS_WAIT(8);
        } else {
          LDS_WAIT(4);
        }
      } else {
        LDS_WAIT(0);
      }
#pragma unroll
      for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[t][uc * 4 + v] =
              __builtin_amdgcn_mfma_f64_4x4x4f64(bn[cb][v], am[ks & 1][t], acc[t][uc * 4 + v], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#ifdef LPGP_STAMP
    STAMP(t_b); st_mfma += t_b - t_a; t_a = t_b;
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces have landed
#ifdef LPGP_STAMP
    STAMP(t_b); st_store += t_b - t_a; t_a = t_b;
#endif
#ifdef LPGP_EXPERIMENT
    if (!(g.ktrim & 2))
#endif
    __syncthreads();
#ifdef LPGP_STAMP
    STAMP(t_b); st_bar += t_b - t_a;
#endif
    cur ^= 1;
  }
#ifdef LPGP_STAMP
  if (g.stamps && tid == 0) {
    unsigned long long* o = g.stamps + 4 * (size_t)blockIdx.x;
    o[0] = st_load; o[1] = st_mfma; o[2] = st_store; o[3] = st_bar;
  }
#endif

  // ---- epilogue: lane holds C[m = 16t + (l&15)][n = 4u + (l>>4)] in acc[t][u]; stores only ----
#pragma unroll
  for (int u = 0; u < 16; ++u)
#pragma unroll
    for (int t = 0; t < 4; ++t) cbase[(int64_t)(4 * u) * g.ldc + t * 16] = alpha * acc[t][u];
}

template <bool TA, bool TB, bool TRI>
static int launch_impl(lpgp_ctx* ctx, hipStream_t stream, const GemmArgs& g) {
  static bool attr_set = false;
  size_t shmem = (size_t)4 * STAGE * sizeof(double);      // 73 728 B: two workgroups per CU
  // 82 KB: more than half of the 160 KB, so two such workgroups never share a CU, yet one of
  // them still fits beside a 73.7 KB workgroup of a concurrent large update.
  const size_t shmem_solo = 82 * 1024;
  if (!attr_set) {
    LPGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f64_kernel<TA, TB, TRI>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem_solo));
    attr_set = true;
  }
  // Small launches (the latency-bound panel steps): the dispatcher packs two workgroups per CU
  // and leaves other CUs idle; each workgroup then gets half of the matrix pipe.  Spread them.
  {
    const int64_t tiles = TRI ? ((int64_t)g.nt * (g.nt + 1) / 2 + (int64_t)(g.mt - g.nt) * g.nt) : (int64_t)g.mt * g.nt;
    if (ctx->solo_small && tiles <= ctx->cus) shmem = shmem_solo;
  }
  GemmArgs ga = g;
  int nsuper = 0, SS = 64;
  for (int sh = 3; sh >= 0; --sh) {
    const int S = 1 << sh;
    const int SR = (g.mt + S - 1) / S, SC = (g.nt + S - 1) / S;
    nsuper = SR * SC;
    if (TRI) {
      LPGP_CHECK(g.mt >= g.nt, "gemm: triangular update needs mt >= nt");
      nsuper = SC * (SC + 1) / 2 + (SR - SC) * SC;
    }
    ga.sshift = sh;
    SS = S * S;
    if (nsuper >= ctx->min_supertiles) break;      // enough super-tiles per XCD to balance the 8 XCDs
  }
  const int64_t blocks = (int64_t)((nsuper + 7) / 8) * 8 * SS;
  hipLaunchKernelGGL((gemm_f64_kernel<TA, TB, TRI>), dim3((unsigned)blocks), dim3(256), shmem, stream, ga);
  LPGP_HIP(hipGetLastError());
  return 0;
}

int launch_gemm(lpgp_ctx* ctx, hipStream_t stream, int ta, int tb, const GemmArgs& g, int prof_kernel) {
  if (g.mt <= 0 || g.nt <= 0 || g.k <= 0) return 0;
  LPGP_CHECK(g.k % BK == 0, "gemm: k=%d not a multiple of %d", g.k, BK);
  LPGP_CHECK(!g.tri || (!ta && !tb), "gemm: the triangular update exists for the NT form only");
  if (prof_kernel >= 0) {
    if (g.tri) prof_kernel = LPGP_K_SYRK;        // one profiling slot == one kernel symbol
    const double m = (double)g.mt * BM, n = (double)g.nt * BN, k = (double)g.k;
    // algorithmic flops: a symmetric update counts the lower trapezoid (m >= n) only
    const double flops = g.tri ? 2.0 * k * (m * n - 0.5 * n * (n - 1.0)) : 2.0 * m * n * k;
    prof_begin(ctx, stream, prof_kernel, flops, 0.0);
  }
  int rc;
  if (g.tri) rc = launch_impl<false, false, true>(ctx, stream, g);
  else if (!ta && !tb) rc = launch_impl<false, false, false>(ctx, stream, g);
  else if (!ta && tb) rc = launch_impl<false, true, false>(ctx, stream, g);
  else if (ta && !tb) rc = launch_impl<true, false, false>(ctx, stream, g);
  else rc = launch_impl<true, true, false>(ctx, stream, g);
  if (prof_kernel >= 0) prof_end(ctx, stream);
  return rc;
}

}  // namespace lpgp
