// fp64 MFMA GEMM / SYRK: the Cholesky trailing update, the rank-128 updates inside a panel, the updates of the blocked
// multi-RHS forward / backward substitution, V0^T V1.  (The refined tile solves live in solve.hip / solve_panel.h.)
//
// Replaces LAPACK dpotrf's trailing DSYRK/DGEMM and dpotrs/dtrtrs's DTRSM that the
// reference reaches through probnum `LinearOperator.cholesky/solve`
// (_conditional.py:44,108,228; linops/_block.py:192-207,233-251).
//
// Three kernels, one operand machinery:
//   gemm_f64_kernel    128 x 128 tile, 4 waves x (64 x 64) from v_mfma_f64_4x4x4_4b_f64 (16x16x4 never exceeds ~48 TFLOP/s
//                      chip-wide, see below), K staged 16 deep by LDS-DMA (global_load_lds) into a double buffer, fragments
//                      double-buffered in registers with hand-counted lgkmcnt waits, accumulators initialised from C (the
//                      epilogue is stores only); 200 registers, 73.7 KB of LDS: two workgroups per CU.
//   gemm3_f64_kernel   the same tile at THREE workgroups per CU (K staged 8 deep, single-buffered m-fragments, 168 registers,
//                      36.9 KB): used where it pays end to end (forward substitution), see its header.
//   gemm64_f64_kernel  64 x 64 tile, 4-stage LDS-DMA ring: launches that cannot fill the chip with 128 x 128 tiles.
// The MFMA operands are SWAPPED (B fragment as the instruction's A operand) so that every accumulator register holds 16
// consecutive rows of one column of C: C is read and written in 128-byte row segments.  LDS images: [k][144] ("M image")
// for an operand whose non-contracted index is fastest in memory, [idx][16] with an XOR swizzle of the 16-byte chunks
// ("K image") for a k-fastest operand; both conflict-free for the one-ds_read_b64-per-fragment pattern.
// Tiles are enumerated densely, band by band, and dealt to the 8 XCDs in contiguous ranges (map_tile_dense): 64
// consecutive entries are an 8 x 8 block of tiles whose 16 operand panels stay in one XCD's L2, and every XCD gets the same
// number of tiles whatever the shape (triangle, trapezoid, or the staircase of a rank's tiles in the multi-GPU update).

#include "lpgp_internal.h"
#include "kernel_util.h"
#include <algorithm>
#include <type_traits>

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

// Addresses are formed as (wave-uniform base) + (32-bit per-lane byte offset) so that hipcc
// selects the SGPR-base addressing mode: one VGPR of lane offsets instead of a 64-bit
// address pair per piece.
template <bool T>
__device__ __forceinline__ void dma_tile(const double* __restrict__ P, int64_t ld, int64_t idx0, int64_t k0,
                                         int lane, int w, double* sdst) {
  if constexpr (!T) {
    const unsigned voff = (unsigned)lane * 16u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int piece = q * 4 + w;                         // wave-uniform
      const char* ub = reinterpret_cast<const char*>(P + idx0 + (k0 + piece) * ld);
      __builtin_amdgcn_global_load_lds((gptr_t)(ub + voff), (lptr_t)(sdst + piece * LDM), 16, 0, 0);
    }
  } else {
    const unsigned row8 = (unsigned)(lane >> 3);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int piece = q * 4 + w;
      // r = piece*8 + (lane>>3); swizzle term ((r >> 1) & 7) = ((piece*4) + (lane>>4)) & 7
      const unsigned c = ((unsigned)lane & 7u) ^ ((((unsigned)piece << 2) + ((unsigned)lane >> 4)) & 7u);
      const unsigned voff = (c * 2u + row8 * (unsigned)ld) * 8u;
      const char* ub = reinterpret_cast<const char*>(P + k0 + (idx0 + (int64_t)piece * 8) * ld);
      __builtin_amdgcn_global_load_lds((gptr_t)(ub + voff), (lptr_t)(sdst + piece * 128), 16, 0, 0);
    }
  }
}

// Fragment element addresses (in doubles, relative to the stage base).
//   m-side fragment:  idx = idx_base + (lane & 15), k = ks*4 + (lane >> 4)
//   n-side fragment:  idx = idx_base + (lane & 3),  k = ks*4 + (lane >> 4)   (replicated over lane bits 2..3:
//                     the "A" operand of v_mfma_f64_4x4x4_4b_f64 with one 4x4 block shared by all four blocks)
// Every address splits into a per-lane part (held in a VGPR, computed once per kernel) and a
// compile-time part that goes into the ds_read offset field:
//   M image  off = k*LDM + idx                              lane: (lane>>4)*LDM + w*64 + (lane&15 | lane&3)
//                                                           imm : ks*4*LDM + idx_const
//   K image  off = idx*16 + (((k>>1) ^ ((idx>>1)&7))<<1) + (k&1)
//     m-side: the swizzle term (lane&15)>>1 is XORed with ks*2 + (lane>>5): one lane part per ks
//     n-side: (idx>>1)&7 = vv*2 + ((lane&3)>>1) and the XOR separates: imm part ((ks^vv)&3)*4
template <bool T>
__device__ __forceinline__ unsigned frag_lane_m(int lane, int w, int ks) {
  if constexpr (!T) return (unsigned)((lane >> 4) * LDM + w * 64 + (lane & 15));
  else return (unsigned)((w * 64 + (lane & 15)) * 16 + ((((ks * 2) + (lane >> 5)) ^ ((lane & 15) >> 1)) << 1) + ((lane >> 4) & 1));
}
template <bool T>
__device__ __forceinline__ unsigned frag_lane_n(int lane, int w) {
  if constexpr (!T) return (unsigned)((lane >> 4) * LDM + w * 64 + (lane & 3));
  else return (unsigned)((w * 64 + (lane & 3)) * 16 + (((((lane & 3) >> 1) ^ (lane >> 5)) & 1) << 1) + ((lane >> 4) & 1));
}
template <bool T> constexpr int frag_imm_m(int ks, int t) { return T ? t * 256 : ks * 4 * LDM + t * 16; }
template <bool T> constexpr int frag_imm_n(int ks, int nf) {       // nf = n-fragment 0..15 (4 columns each)
  return T ? nf * 64 + (((ks ^ nf) & 3) << 2) : ks * 4 * LDM + nf * 4;
}

// MFMA shape: measured on MI355X, v_mfma_f64_16x16x4_f64 sustains only ~48 TFLOP/s chip-wide
// (>= 88 cycles per instruction per SIMD at any occupancy) while v_mfma_f64_4x4x4_4b_f64
// issues every 16.5 cycles = 75 TFLOP/s already at one wave per SIMD (scratch/mfma_probe.hip,
// DESIGN.md section 5).  The 64x64 wave tile is therefore built from 4x4x4_4b instructions:
// per k-step of 4, acc[t][u] (m = 16t + (lane&15), n = 4u + (lane>>4)) += Bfrag4[u] x Afrag[t],
// where the instruction's four 4x4 blocks share the 4 n-columns (replicated "A" operand) and
// cover 16 consecutive rows m ("B" operand) -- the same 128-byte-segment C layout as before.
// Virtual block id -> tile.  Blocks b, b+8, b+16, ... share an XCD (round-robin dispatch), so
// every run of S*S ids with equal (b & 7) is one S x S super-tile worked on by ONE XCD (its
// operand panels stay in that XCD's L2).  S = 1 << sshift in {8,4,2,1}: 8 for large grids, smaller
// when there would be too few super-tiles to balance the 8 XCDs (measured with S = 8 only: a
// 32x32-tile SYRK ran two rounds on two XCDs and one on the others, 27 instead of ~50 TFLOP/s).
// For the triangular case only super-tiles on or below the diagonal are enumerated.
template <bool TRI>
__device__ __forceinline__ bool map_tile(const GemmArgs& g, int v, int& tr, int& tc) {
  const int sh = g.sshift, S = 1 << sh, SS = S * S;
  const int SR = (g.mt + S - 1) >> sh, SC = (g.nt + S - 1) >> sh;
  const int xcd = v & 7, q = v >> 3;
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
  if (sr >= SR || sc >= SC) return false;
  // (sqrtf went through the vector ALU: bring the wave-uniform results back to scalar registers,
  // or every address derived from them lives in VGPRs)
  tr = __builtin_amdgcn_readfirstlane((sr << sh) + (inner & (S - 1)));
  tc = __builtin_amdgcn_readfirstlane((sc << sh) + (inner >> sh));
  if (tr >= g.mt || tc >= g.nt) return false;
  if (TRI && tr < tc) return false;
  return true;
}

// Dense enumeration: list index -> tile.  Bands of 8 tile rows; inside a band column-major over
// the valid columns (c < nt, and c <= r for the triangular shapes), so that 64 consecutive
// indices are one 8 x 8 block of tiles (8 row panels + 8 column panels in the XCD's L2).  XCD x
// (blocks with v & 7 == x) walks the contiguous range [x * chunk, (x + 1) * chunk) of the list:
// every XCD gets the same number of tiles whatever the shape.  (The super-tile dealing of map_tile
// hands a triangular super-tile on the diagonal -- 36 tiles -- to an XCD with 64 slots; measured
// on SYRK n x n x 512: n = 16384 51.2 -> 51.9, 12288 48.1 -> 49.9, 6144 49.2 -> 51.0 TFLOP/s,
// rectangular shapes unchanged; scratch/dense_ab.py.)
constexpr int BAND = 8;                  // default band height (GemmArgs::band; LPGP_GEMM_BAND = 4 / 8 / 16 for the measurement in DESIGN.md §5)
// Distributed trailing update (GemmArgs::cyc): number of valid local tile columns of local tile row r -- the
// columns whose global tile index does not exceed the row's (non-decreasing in r: a staircase).
__host__ __device__ __forceinline__ int stair_nc(const GemmArgs& g, int r) {
  const int gr = cyc_l2g(g.rowc, g.rt0 + r);
  const int n = cyc_before(g.colc, gr + 1) - g.ct0;
  return n < 0 ? 0 : (n > g.nt ? g.nt : n);
}
// tiles of band b (tile rows r0 .. r0 + R - 1): column-major over the valid (row, column) pairs
__host__ __device__ __forceinline__ int stair_band_count(const GemmArgs& g, int r0, int R) {
  int acc = 0;
  for (int i = 0; i < R; ++i) acc += stair_nc(g, r0 + i);
  return acc;
}
// j-th tile of the band of tile rows r0 .. r0 + R - 1: columns below nc(r0) hold all R rows of the band, a column c
// beyond holds the rows with nc(r) > c (a suffix of the band: nc is non-decreasing)
__host__ __device__ __forceinline__ void stair_decode(const GemmArgs& g, int r0, int R, int j, int& r, int& c) {
  const int cmin = stair_nc(g, r0);
  if (j < cmin * R) {
    c = j / R;
    r = r0 + (j - c * R);
    return;
  }
  int jj = j - cmin * R;
  for (c = cmin;; ++c) {
    int rf = r0;                                           // first row of the band that reaches column c
    while (stair_nc(g, rf) <= c) ++rf;
    const int cnt = r0 + R - rf;
    if (jj < cnt) {
      r = rf + jj;
      return;
    }
    jj -= cnt;
  }
}

template <bool TRI>
__device__ __forceinline__ bool map_tile_dense(const GemmArgs& g, int v, int& tr, int& tc) {
  const int xcd = v & 7, q = v >> 3;
  const int idx = xcd * g.chunk + q;
  if (idx >= g.ntiles) return false;
  int b, j;
  if (!TRI) {
    const int per = g.band * g.nt;
    b = idx / per;
    j = idx - b * per;
  } else {
    int lo = 0, hi = g.nbands;                       // band_prefix[lo] <= idx < band_prefix[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (g.band_prefix[mid] <= idx) lo = mid; else hi = mid;
    }
    b = lo;
    j = idx - g.band_prefix[b];
  }
  const int r0 = b * g.band;
  const int R = (g.mt - r0 < g.band) ? g.mt - r0 : g.band;
  int r, c;
  if (!TRI) {
    c = j / R;
    r = r0 + (j - c * R);
  } else {
    const int cfull = (r0 + 1 < g.nt) ? r0 + 1 : g.nt;       // columns c <= r0: all R rows valid
    if (g.cyc) {
      stair_decode(g, r0, R, j, r, c);
    } else if (j < cfull * R) {
      c = j / R;
      r = r0 + (j - c * R);
    } else {
      int jj = j - cfull * R, t = 0;
      while (jj >= R - 1 - t) { jj -= R - 1 - t; ++t; }      // column r0+1+t has rows r0+1+t .. r0+R-1
      c = r0 + 1 + t;
      r = c + jj;
    }
  }
  tr = __builtin_amdgcn_readfirstlane(r);
  tc = __builtin_amdgcn_readfirstlane(c);
  return true;
}

// TRI != 0: lower-triangular output (symmetric rank-k update).  TRI = 1 is the rank-nb trailing
// update of the blocked Cholesky (the remainder half of the look-ahead split: the bulk of the
// flops), TRI = 3 its look-ahead half (the next panel's columns, on the panel stream, possibly
// underneath the remainder), TRI = 2 the rank-128 update inside a panel: identical code,
// distinct instantiations, so that each shows up under its own kernel symbol in
// rocprofv3 --stats and in its own HIP-event profiling slot.
//
// One 128x128 tile per workgroup, two workgroups per CU.  (A persistent variant -- 2 workgroups
// per CU walking the tile list with cross-tile prefetch and a half-tile stagger between the two
// co-resident workgroups -- was measured and is 1-4 % SLOWER: the dispatcher already overlaps
// one workgroup's prologue with its neighbour's k-loop.  In-kernel stamps (-DLPGP_STAMP,
// scratch/stamp_test.hip) show the k-loop at 16.1 cycles per MFMA per SIMD, i.e. the matrix pipe
// is saturated; what is left is the C prologue (~15k of ~280k cycles per tile at k = 512) and, in
// SHORT measurements, the clock: for the first ~40 ms after the onset of load the chip runs at
// 1.7-2.07 GHz on random data (2.35 GHz on all-zero operands) -- a burst of five launches sees
// 53-55 TFLOP/s at k = 512 -- and then settles at 2.4 GHz, where the same launch sustains 65 TFLOP/s
// (68.5 at k = 2048; 59 on the CU-masked update stream; profiles/r03_clock_power.txt).  Folding C into the
// first nine k-stages instead (accumulators from zero, 8 C values per lane loaded per stage and
// added one stage later, so that the MFMAs start as soon as the first operand stage lands) was
// built and measured: no gain either (51 vs 52 TFLOP/s at k = 512) -- the co-resident workgroup
// does hide the prologue.)
template <bool TA, bool TB, int TRI>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid & 1, wn = wid >> 1;
  double* sA = smem;                  // 2 stages
  double* sB = smem + 2 * STAGE;
  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) double*)smem);
  const int KT = g.k / BK;
  const int wu = __builtin_amdgcn_readfirstlane(wid);    // wave index as a scalar (LDS-DMA base must be uniform)
  const int wum = wu & 1, wun = wu >> 1;
  const double alpha = g.alpha, beta = g.beta;

#ifdef LPGP_STAMP
  const unsigned long long st_rt_start = __builtin_amdgcn_s_memrealtime();
#endif
  int tr, tc;
  if (g.dense) {
    if (!map_tile_dense<(TRI != 0)>(g, (int)blockIdx.x, tr, tc)) return;
  } else {
    if (!map_tile<(TRI != 0)>(g, (int)blockIdx.x, tr, tc)) return;
  }

  // per-lane byte addresses of the fragment reads (stage 0 of each operand; see frag_lane_*)
  unsigned laneM[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) laneM[ks] = lds_base + 8u * frag_lane_m<TA>(lane, wm, TA ? ks : 0);
  const unsigned laneN = lds_base + (unsigned)(2 * STAGE) * 8u + 8u * frag_lane_n<TB>(lane, wn);

  // operand rows: the tile's own row / column -- or, in the distributed update, the rows of the gathered panel
  // that belong to the GLOBAL tiles this local tile stands for
  int64_t aidx = (int64_t)tr * BM, bidx = (int64_t)tc * BN;
  if (TRI != 0 && g.cyc) {
    aidx = (int64_t)(cyc_l2g(g.rowc, g.rt0 + tr) - g.g0) * BM;
    bidx = (int64_t)(cyc_l2g(g.colc, g.ct0 + tc) - g.g0) * BN;
  }
  dma_tile<TA>(g.A, g.lda, aidx, 0, lane, wu, sA);
  dma_tile<TB>(g.B, g.ldb, bidx, 0, lane, wu, sB);

#ifdef LPGP_STAMP
  const unsigned long long st_pro0 = __builtin_amdgcn_s_memtime();
#endif
  // The accumulators start from (beta/alpha) * C, so that the epilogue is stores only: the 64
  // C loads per lane are all in flight at once instead of 16 dependent load->fma->store
  // round trips at the end of the tile.
  // C element (t,u) of this lane: uniform tile/wave origin + uniform (4u*ldc + 16t) + lane offset
  char* const cub = reinterpret_cast<char*>(g.C + ((int64_t)tc * BN + wun * 64) * g.ldc + (int64_t)tr * BM + wum * 64);
  const unsigned cvoff = ((unsigned)(lane >> 4) * (unsigned)g.ldc + (unsigned)(lane & 15)) * 8u;
  double acc[4][16];
  if (beta != 0.0) {
    const double sc_ = beta / alpha;
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[t][u] = sc_ * *reinterpret_cast<const double*>(cub + ((int64_t)(4 * u) * g.ldc + t * 16) * 8 + cvoff);
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[t][u] = 0.0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

#ifdef LPGP_STAMP
  unsigned long long st_acc[4] = {0, 0, 0, 0};
  const unsigned long long st_tile0 = __builtin_amdgcn_s_memtime();
  const unsigned long long st_real0 = __builtin_amdgcn_s_memrealtime();
#endif
  int cur = 0;
  for (int kt = 0; kt < KT; ++kt) {
#ifdef LPGP_STAMP
    const unsigned long long st0 = __builtin_amdgcn_s_memtime();
#endif
    // next stage -> other LDS buffer (unconditional: a branch around the DMA would make hipcc
    // drain vmcnt early; the last stage harmlessly re-loads its own k-tile)
    {
      const int knext = (kt + 1 < KT ? kt + 1 : kt) * BK;
      dma_tile<TA>(g.A, g.lda, aidx, knext, lane, wu, sA + (cur ^ 1) * STAGE);
      dma_tile<TB>(g.B, g.ldb, bidx, knext, lane, wu, sB + (cur ^ 1) * STAGE);
    }
    // 16 chunks of 16 MFMAs per stage (4 k-steps x 4 groups of 4 n-fragments).  Fragments
    // are double-buffered in registers: the LDS reads of chunk c+1 are issued before the
    // MFMAs of chunk c (264 cycles of matrix work cover the LDS latency) and retired with a
    // counted wait that leaves exactly those newer reads in flight.
    const unsigned stoff = (unsigned)(cur * STAGE) * 8u;
    const unsigned aM0 = laneM[0] + stoff, aM1 = laneM[1] + stoff, aM2 = laneM[2] + stoff, aM3 = laneM[3] + stoff;
    const unsigned aN = laneN + stoff;
    double am[2][4], bn[2][4];
    asm volatile("" ::: "memory");
#ifdef LPGP_STAMP
    const unsigned long long st1 = __builtin_amdgcn_s_memtime();
#endif
    static_for<0, 4>([&](auto T_) {
      constexpr int t = decltype(T_)::value;
      am[0][t] = lds_read_async<frag_imm_m<TA>(0, t)>(aM0);
    });
    static_for<0, 4>([&](auto V_) {
      constexpr int vv = decltype(V_)::value;
      bn[0][vv] = lds_read_async<frag_imm_n<TB>(0, vv)>(aN);
    });
    static_for<0, 16>([&](auto C_) {
      constexpr int c = decltype(C_)::value;
      constexpr int ks = c >> 2, uc = c & 3, cb = c & 1;
      if constexpr (c + 1 < 16) {
        constexpr int ks2 = (c + 1) >> 2, uc2 = (c + 1) & 3;
        static_for<0, 4>([&](auto V_) {
          constexpr int vv = decltype(V_)::value;
          bn[cb ^ 1][vv] = lds_read_async<frag_imm_n<TB>(ks2, uc2 * 4 + vv)>(aN);
        });
        if constexpr (uc2 == 0) {
          const unsigned aMk = ks2 == 1 ? aM1 : (ks2 == 2 ? aM2 : aM3);
          static_for<0, 4>([&](auto T_) {
            constexpr int t = decltype(T_)::value;
            am[ks2 & 1][t] = lds_read_async<frag_imm_m<TA>(ks2, t)>(aMk);
          });
          LDS_WAIT(8);
        } else {
          LDS_WAIT(4);
        }
      } else {
        LDS_WAIT(0);
      }
#pragma unroll
      for (int vv = 0; vv < 4; ++vv)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[t][uc * 4 + vv] =
              __builtin_amdgcn_mfma_f64_4x4x4f64(bn[cb][vv], am[ks & 1][t], acc[t][uc * 4 + vv], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
#ifdef LPGP_STAMP
    const unsigned long long st2 = __builtin_amdgcn_s_memtime();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces have landed
#ifdef LPGP_STAMP
    const unsigned long long st3 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
    cur ^= 1;
#ifdef LPGP_STAMP
    const unsigned long long st4 = __builtin_amdgcn_s_memtime();
    st_acc[0] += st1 - st0; st_acc[1] += st2 - st1; st_acc[2] += st3 - st2; st_acc[3] += st4 - st3;
#endif
  }
#ifdef LPGP_STAMP
  if (g.stamps != nullptr && tid == 0) {
    unsigned long long* o = g.stamps + (size_t)blockIdx.x * 8;
    o[0] = st_acc[0]; o[1] = st_acc[1]; o[2] = st_acc[2]; o[3] = st_acc[3];
    o[4] = st_tile0 - st_pro0;                            // prologue (C loads + first DMA wait)
    o[5] = __builtin_amdgcn_s_memtime() - st_tile0;       // whole k loop, core clocks
    o[6] = __builtin_amdgcn_s_memrealtime() - st_real0;   // same in 100 MHz ticks
  }
  const unsigned long long st_rt_kend = __builtin_amdgcn_s_memrealtime();
#endif

  // ---- epilogue: lane holds C[m = 16t + (l&15)][n = 4u + (l>>4)] in acc[t][u]; stores only ----
#pragma unroll
  for (int u = 0; u < 16; ++u)
#pragma unroll
    for (int t = 0; t < 4; ++t)
      *reinterpret_cast<double*>(cub + ((int64_t)(4 * u) * g.ldc + t * 16) * 8 + cvoff) = alpha * acc[t][u];
#ifdef LPGP_STAMP
  if (g.timeline != nullptr && tid == 0) {
    // workgroup life cycle in 100 MHz ticks + where it ran (scratch/timeline.hip):
    // [start, k-loop begin, k-loop end, stores issued, hw id, xcc]
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* o = g.timeline + (size_t)blockIdx.x * 8;
    o[0] = st_rt_start; o[1] = st_real0; o[2] = st_rt_kend; o[3] = __builtin_amdgcn_s_memrealtime();
    o[4] = hw; o[5] = xcc & 0xf;
  }
#endif
}

// =========================================================================================
// THREE workgroups per CU (round 3).  gemm_f64_kernel above holds 200 registers per lane and 73.7 KB of LDS: exactly two
// workgroups per CU, and its own stamps say the matrix pipe is saturated only while both are in their k-loops (64-73 % of
// the time at K = 512: a workgroup spends 13-15 us of its ~145 in the C prologue, and a lone workgroup drives the pipe at
// 82 %).  This variant trades the register double-buffering of the fragments for a third resident workgroup:
//   * K staged 8 deep, double-buffered: 2 x 2 x 8 x 144 x 8 B = 36 864 B of LDS (three fit in 160 KB);
//   * 128 accumulator registers + ONE set of m-fragments (8) + two sets of n-fragments (16) + addresses <= 168, so that
//     __launch_bounds__(256, 3) holds without spills: the m-fragments of the next k-step are re-loaded inside the last chunk
//     of the current one, each right behind the four MFMAs that read it last (t-major order in that chunk); the LDS latency
//     that is left is covered by the two other waves of the SIMD instead of by registers;
//   * same operand image (M image, [k][LDM]), same swapped-operand accumulator layout, same tile enumerations, same epilogue.
// A always has its non-contracted index fastest (M image); B either way: NT form (TB = false: the symmetric rank-nb updates of
// the factorisation) and NN form (TB = true: B k-fastest, the updates of the blocked forward substitution V <- V - L_panel V_p).
// Launches with fewer than `gemm3` tiles (default 768 = three per CU) stay on the two-resident kernel: a workgroup that is
// alone on its CU has nobody to cover its LDS latency (measured: the look-ahead half of the update, 4 tile columns, ran at 20
// instead of 30 TFLOP/s on this kernel).  So do the updates that run BESIDE a panel chain that is not far shorter than they
// are (GemmArgs::occ3, cleared by the schedulers): three resident waves of 168 registers fill a SIMD's register file, and
// the chain's kernels then wait for a whole workgroup to retire before any of their waves fits -- measured in situ, the
// update itself gains 5 % (50.5 -> 53.1 TFLOP/s by HIP events) and the c3 step nothing (56.5 vs 56.5 ms).
// =========================================================================================
constexpr int BK3 = 8;
constexpr int STAGE3 = BK3 * LDM;        // doubles per operand per stage

// T = false: M image [k][LDM] (one piece = one k-row).  T = true (k fastest in memory): image [idx][8] -- one piece = 16
// index rows of 64 B; no swizzle needed: the only reader is the replicated n-side fragment, whose 16 distinct addresses
// (4 indices x 4 k) fall into 32 distinct banks.
template <bool T>
__device__ __forceinline__ void dma_tile3(const double* __restrict__ P, int64_t ld, int64_t idx0, int64_t k0, int lane, int w,
                                          double* sdst) {
  if constexpr (!T) {
    const unsigned voff = (unsigned)lane * 16u;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int piece = q * 4 + w;                         // k-row, wave-uniform
      const char* ub = reinterpret_cast<const char*>(P + idx0 + (k0 + piece) * ld);
      __builtin_amdgcn_global_load_lds((gptr_t)(ub + voff), (lptr_t)(sdst + piece * LDM), 16, 0, 0);
    }
  } else {
    const unsigned voff = (((unsigned)lane & 3u) * 2u + ((unsigned)lane >> 2) * (unsigned)ld) * 8u;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int piece = q * 4 + w;                         // 16 index rows
      const char* ub = reinterpret_cast<const char*>(P + k0 + (idx0 + (int64_t)piece * 16) * ld);
      __builtin_amdgcn_global_load_lds((gptr_t)(ub + voff), (lptr_t)(sdst + piece * 128), 16, 0, 0);
    }
  }
}
template <bool T> __device__ __forceinline__ unsigned frag3_lane_n(int lane, int w) {
  if constexpr (!T) return frag_lane_n<false>(lane, w);
  else return (unsigned)((w * 64 + (lane & 3)) * BK3 + (lane >> 4));
}
template <bool T> constexpr int frag3_imm_n(int ks, int nf) { return T ? nf * 4 * BK3 + ks * 4 : frag_imm_n<false>(ks, nf); }

template <bool TB, int TRI>
__global__ __launch_bounds__(256, 3) void gemm3_f64_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid & 1, wn = wid >> 1;
  double* sA = smem;                  // 2 stages
  double* sB = smem + 2 * STAGE3;
  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) double*)smem);
  const int KT = g.k / BK3;
  const int wu = __builtin_amdgcn_readfirstlane(wid);
  const int wum = wu & 1, wun = wu >> 1;
  const double alpha = g.alpha, beta = g.beta;

  int tr, tc;
  if (g.dense) {
    if (!map_tile_dense<(TRI != 0)>(g, (int)blockIdx.x, tr, tc)) return;
  } else {
    if (!map_tile<(TRI != 0)>(g, (int)blockIdx.x, tr, tc)) return;
  }
  const unsigned laneM = lds_base + 8u * frag_lane_m<false>(lane, wm, 0);
  const unsigned laneN = lds_base + (unsigned)(2 * STAGE3) * 8u + 8u * frag3_lane_n<TB>(lane, wn);

  int64_t aidx = (int64_t)tr * BM, bidx = (int64_t)tc * BN;
  if (TRI != 0 && g.cyc) {
    aidx = (int64_t)(cyc_l2g(g.rowc, g.rt0 + tr) - g.g0) * BM;
    bidx = (int64_t)(cyc_l2g(g.colc, g.ct0 + tc) - g.g0) * BN;
  }
  dma_tile3<false>(g.A, g.lda, aidx, 0, lane, wu, sA);
  dma_tile3<TB>(g.B, g.ldb, bidx, 0, lane, wu, sB);

  char* const cub = reinterpret_cast<char*>(g.C + ((int64_t)tc * BN + wun * 64) * g.ldc + (int64_t)tr * BM + wum * 64);
  const unsigned cvoff = ((unsigned)(lane >> 4) * (unsigned)g.ldc + (unsigned)(lane & 15)) * 8u;
  double acc[4][16];
  if (beta != 0.0) {
    const double sc_ = beta / alpha;
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[t][u] = sc_ * *reinterpret_cast<const double*>(cub + ((int64_t)(4 * u) * g.ldc + t * 16) * 8 + cvoff);
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[t][u] = 0.0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < KT; ++kt) {
    {
      const int knext = (kt + 1 < KT ? kt + 1 : kt) * BK3;      // (unconditional, see gemm_f64_kernel)
      dma_tile3<false>(g.A, g.lda, aidx, knext, lane, wu, sA + (cur ^ 1) * STAGE3);
      dma_tile3<TB>(g.B, g.ldb, bidx, knext, lane, wu, sB + (cur ^ 1) * STAGE3);
    }
    const unsigned stoff = (unsigned)(cur * STAGE3) * 8u;
    const unsigned aM = laneM + stoff, aN = laneN + stoff;
    double am[4], bn[2][4];
    asm volatile("" ::: "memory");
    static_for<0, 4>([&](auto T_) {
      constexpr int t = decltype(T_)::value;
      am[t] = lds_read_async<frag_imm_m<false>(0, t)>(aM);
    });
    static_for<0, 4>([&](auto V_) {
      constexpr int vv = decltype(V_)::value;
      bn[0][vv] = lds_read_async<frag3_imm_n<TB>(0, vv)>(aN);
    });
    // 8 chunks of 16 MFMAs per stage (2 k-steps x 4 groups of 4 n-fragments); LDS operations return in order:
    // at the top of chunk c the n-fragments of chunk c + 1 are issued, and "all but the newest four" covers this chunk's
    // n-fragments and, after a k-step boundary, the re-loaded m-fragments
    static_for<0, 8>([&](auto C_) {
      constexpr int c = decltype(C_)::value;
      constexpr int ks = c >> 2, uc = c & 3, cb = c & 1;
      if constexpr (c + 1 < 8) {
        constexpr int ks2 = (c + 1) >> 2, uc2 = (c + 1) & 3;
        static_for<0, 4>([&](auto V_) {
          constexpr int vv = decltype(V_)::value;
          bn[cb ^ 1][vv] = lds_read_async<frag3_imm_n<TB>(ks2, uc2 * 4 + vv)>(aN);
        });
        LDS_WAIT(4);
      } else {
        LDS_WAIT(0);
      }
      if constexpr (uc == 3 && c + 1 < 8) {
        // last chunk of a k-step: m-fragment by m-fragment, each re-loaded for the next k-step behind its last reader
        static_for<0, 4>([&](auto T_) {
          constexpr int t = decltype(T_)::value;
#pragma unroll
          for (int vv = 0; vv < 4; ++vv)
            acc[t][uc * 4 + vv] = __builtin_amdgcn_mfma_f64_4x4x4f64(bn[cb][vv], am[t], acc[t][uc * 4 + vv], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          am[t] = lds_read_async<frag_imm_m<false>(ks + 1, t)>(aM);
          __builtin_amdgcn_sched_barrier(0);
        });
      } else {
#pragma unroll
        for (int vv = 0; vv < 4; ++vv)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            acc[t][uc * 4 + vv] = __builtin_amdgcn_mfma_f64_4x4x4f64(bn[cb][vv], am[t], acc[t][uc * 4 + vv], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces have landed
    __syncthreads();
    cur ^= 1;
  }
#pragma unroll
  for (int u = 0; u < 16; ++u)
#pragma unroll
    for (int t = 0; t < 4; ++t)
      *reinterpret_cast<double*>(cub + ((int64_t)(4 * u) * g.ldc + t * 16) * 8 + cvoff) = alpha * acc[t][u];
}

// =========================================================================================
// 64 x 64 tile variant for launches that cannot fill the chip with 128 x 128 tiles: the
// latency-bound steps of the panel chain (rank-128 solves and updates with 8 ... 100 tiles)
// and the trailing updates of the last panels.  One 128 x 128 x K tile occupies a CU's four
// matrix pipes for K/512 * 56 us whatever else is idle; a quarter tile takes a quarter of
// that and four times as many CUs work.  Same operand images as above with BK = 16:
//  * M image, 64 indices per k-row: one DMA wave-instruction moves TWO k-rows (lanes 0-31 /
//    32-63), stored as [k/2][2][64] with 16 doubles of padding per pair (LDMP = 144, so that
//    consecutive pairs start 32 banks apart).  The four k-slots of an MFMA step read rows
//    4ks + {0, 2, 1, 3}: slots 0/1 (lanes 0-31) then come from different pairs = different
//    bank halves.  Both operands use the same slot order, so the contraction is unchanged.
//  * K image: [idx][16] with the chunk swizzle of the big kernel (8 pieces of 8 rows).
// Four stages of 16 k in a ring (73.7 KB, two workgroups per CU), three in flight; wave w owns
// the 64 x 16 column slab w of the tile: acc[t][u] = C[16t + (l&15)][16w + 4u + (l>>4)].
// =========================================================================================
constexpr int LDMP = 144;
constexpr int SOPER = 8 * LDMP;          // doubles per operand per stage (K image uses 1024 of them)

template <bool T>
__device__ __forceinline__ void dma_tile64(const double* __restrict__ P, int64_t ld, int64_t idx0, int64_t k0,
                                           int lane, int w, double* sdst) {
  if constexpr (!T) {
    const unsigned voff = ((unsigned)(lane >> 5) * (unsigned)ld + (unsigned)(lane & 31) * 2u) * 8u;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const int q = qq * 4 + w;                            // k-row pair, wave-uniform
      const char* ub = reinterpret_cast<const char*>(P + idx0 + (k0 + 2 * q) * ld);
      __builtin_amdgcn_global_load_lds((gptr_t)(ub + voff), (lptr_t)(sdst + q * LDMP), 16, 0, 0);
    }
  } else {
    const unsigned row8 = (unsigned)(lane >> 3);
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const int q = qq * 4 + w;                            // 8 index rows
      const unsigned c = ((unsigned)lane & 7u) ^ ((((unsigned)q << 2) + ((unsigned)lane >> 4)) & 7u);
      const unsigned voff = (c * 2u + row8 * (unsigned)ld) * 8u;
      const char* ub = reinterpret_cast<const char*>(P + k0 + (idx0 + (int64_t)q * 8) * ld);
      __builtin_amdgcn_global_load_lds((gptr_t)(ub + voff), (lptr_t)(sdst + q * 128), 16, 0, 0);
    }
  }
}

// fragment addresses (doubles, relative to the operand's stage base); slot j = lane >> 4
template <bool T>
__device__ __forceinline__ unsigned frag64_lane_m(int lane, int ks) {
  const int j = lane >> 4, i = lane & 15;
  if constexpr (!T) return (unsigned)((j & 1) * LDMP + (j >> 1) * 64 + i);
  else return (unsigned)(i * 16 + ((((2 * ks) + (j & 1)) ^ (i >> 1)) << 1) + (j >> 1));
}
template <bool T>
__device__ __forceinline__ unsigned frag64_lane_n(int lane, int w) {
  const int j = lane >> 4, i = lane & 3;
  if constexpr (!T) return (unsigned)((j & 1) * LDMP + (j >> 1) * 64 + 16 * w + i);
  else return (unsigned)((16 * w + i) * 16 + ((((j & 1) ^ (i >> 1)) & 1) << 1) + (j >> 1));
}
template <bool T> constexpr int frag64_imm_m(int ks, int t) { return T ? t * 256 : 2 * ks * LDMP + 16 * t; }
template <bool T> constexpr int frag64_imm_n(int ks, int u) { return T ? u * 64 + (((ks ^ u) & 3) << 2) : 2 * ks * LDMP + 4 * u; }

// SS = stages of the ring: 4 (73.7 KB, three stages in flight, two workgroups per CU) or 2 (36.9 KB, four workgroups per
// CU: round 4, for the rank-128 updates inside a panel -- K = 128 is 8 stages, and on the CUs the narrow update stream
// leaves to the chain the launch is bound by how many workgroups fit, not by the depth of its prefetch).
template <bool TA, bool TB, int TRI, int SS>
__global__ __launch_bounds__(256, 2) void gemm64_f64_kernel(GemmArgs g) {
  static_assert(SS == 2 || SS == 4, "gemm64: ring of 2 or 4 stages");
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wu = __builtin_amdgcn_readfirstlane(wid);
  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) double*)smem);
  const int KT = g.k / BK;
  const double alpha = g.alpha, beta = g.beta;
  const int mt = g.mt * 2;                          // 64-row tiles
  const int tr = (int)blockIdx.x % mt, tc = (int)blockIdx.x / mt;
  if (TRI != 0 && tr < tc) return;                  // (upper quarter of a diagonal 128-tile: never read)

  auto sA = [&](int b) { return smem + (size_t)b * 2 * SOPER; };
  auto sB = [&](int b) { return smem + (size_t)b * 2 * SOPER + SOPER; };
  auto issue = [&](int kt) {
    const int b = kt & (SS - 1);
    dma_tile64<TA>(g.A, g.lda, (int64_t)tr * 64, (int64_t)kt * BK, lane, wu, sA(b));
    dma_tile64<TB>(g.B, g.ldb, (int64_t)tc * 64, (int64_t)kt * BK, lane, wu, sB(b));
  };
  // SS - 1 stages in flight before anything else (each wave: 4 DMA instructions per stage)
#pragma unroll
  for (int kt = 0; kt < SS - 1; ++kt)
    if (kt < KT) issue(kt);

  unsigned laneM[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) laneM[ks] = lds_base + 8u * frag64_lane_m<TA>(lane, TA ? ks : 0);
  const unsigned laneN = lds_base + (unsigned)SOPER * 8u + 8u * frag64_lane_n<TB>(lane, wid);

  char* const cub = reinterpret_cast<char*>(g.C + ((int64_t)tc * 64 + wu * 16) * g.ldc + (int64_t)tr * 64);
  const unsigned cvoff = ((unsigned)(lane >> 4) * (unsigned)g.ldc + (unsigned)(lane & 15)) * 8u;
  double acc[4][4];
  if (beta != 0.0) {
    const double sc_ = beta / alpha;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[t][u] = sc_ * *reinterpret_cast<const double*>(cub + ((int64_t)(4 * u) * g.ldc + t * 16) * 8 + cvoff);
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[t][u] = 0.0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // C and the first three stages

  for (int kt = 0; kt < KT; ++kt) {
    // stage kt has landed once at most the stages issued after it are outstanding
    const int later = KT - 1 - kt;
    if (SS == 4 && later >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (SS == 4 && later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                    // ... for every wave; and stage kt-1 is consumed
    if (kt + SS - 1 < KT) issue(kt + SS - 1);           // into the buffer stage kt-1 was read from
    const unsigned stoff = (unsigned)((kt & (SS - 1)) * 2 * SOPER) * 8u;
    const unsigned aM0 = laneM[0] + stoff, aM1 = laneM[1] + stoff, aM2 = laneM[2] + stoff, aM3 = laneM[3] + stoff;
    const unsigned aN = laneN + stoff;
    double am[2][4], bn[2][4];
    asm volatile("" ::: "memory");
    static_for<0, 4>([&](auto T_) {
      constexpr int t = decltype(T_)::value;
      am[0][t] = lds_read_async<frag64_imm_m<TA>(0, t)>(aM0);
    });
    static_for<0, 4>([&](auto U_) {
      constexpr int u = decltype(U_)::value;
      bn[0][u] = lds_read_async<frag64_imm_n<TB>(0, u)>(aN);
    });
    static_for<0, 4>([&](auto K_) {
      constexpr int ks = decltype(K_)::value;
      if constexpr (ks + 1 < 4) {
        const unsigned aMk = ks + 1 == 1 ? aM1 : (ks + 1 == 2 ? aM2 : aM3);
        static_for<0, 4>([&](auto T_) {
          constexpr int t = decltype(T_)::value;
          am[(ks + 1) & 1][t] = lds_read_async<frag64_imm_m<TA>(ks + 1, t)>(aMk);
        });
        static_for<0, 4>([&](auto U_) {
          constexpr int u = decltype(U_)::value;
          bn[(ks + 1) & 1][u] = lds_read_async<frag64_imm_n<TB>(ks + 1, u)>(aN);
        });
        LDS_WAIT(8);
      } else {
        LDS_WAIT(0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[t][u] = __builtin_amdgcn_mfma_f64_4x4x4f64(bn[ks & 1][u], am[ks & 1][t], acc[t][u], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
  }

#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int t = 0; t < 4; ++t)
      *reinterpret_cast<double*>(cub + ((int64_t)(4 * u) * g.ldc + t * 16) * 8 + cvoff) = alpha * acc[t][u];
}

template <bool TA, bool TB, int TRI, int SS>
static int launch_small_ss(lpgp_ctx* ctx, hipStream_t stream, const GemmArgs& g) {
  const size_t shmem = (size_t)SS * 2 * SOPER * sizeof(double);      // 73 728 B / 36 864 B
  LPGP_TRY_RC(ensure_lds_attr(ctx, reinterpret_cast<const void*>(&gemm64_f64_kernel<TA, TB, TRI, SS>), shmem));
  const unsigned blocks = (unsigned)(g.mt * 2) * (unsigned)(g.nt * 2);
  hipLaunchKernelGGL((gemm64_f64_kernel<TA, TB, TRI, SS>), dim3(blocks), dim3(256), shmem, stream, g);
  LPGP_HIP(hipGetLastError());
  return 0;
}
template <bool TA, bool TB, int TRI>
static int launch_small(lpgp_ctx* ctx, hipStream_t stream, const GemmArgs& g) {
  if constexpr (TRI == 2) {
    // the rank-128 update inside a panel: wide (all rows below) and short (8 stages)
    if (ctx->small_ring2 && g.k <= 128 && (int64_t)g.mt * g.nt >= ctx->small_ring2) return launch_small_ss<TA, TB, TRI, 2>(ctx, stream, g);
  }
  return launch_small_ss<TA, TB, TRI, 4>(ctx, stream, g);
}

// the kernel of a launch: the three-resident variant for the NT form when the context asks for it (LPGP_GEMM3)
template <bool TA, bool TB, int TRI>
static void pick_kernel(const lpgp_ctx* ctx, int64_t ntiles, bool occ3, void (**fn)(GemmArgs), size_t* shmem) {
  *fn = gemm_f64_kernel<TA, TB, TRI>;
  *shmem = (size_t)4 * STAGE * sizeof(double);                  // 73 728 B: two workgroups per CU
  if constexpr (!TA && TRI != 2 && !(TB && TRI != 0)) {
    // (symmetric updates -- the factorisation's -- only with gemm3_fact: there the third resident workgroup costs the panel
    //  chain what it gains the update, and a profiling slot stays one kernel symbol)
    if (occ3 && ctx->gemm3 > 0 && ntiles >= ctx->gemm3 && (TRI == 0 || ctx->gemm3_fact)) {
      *fn = gemm3_f64_kernel<TB, TRI>;
      *shmem = (size_t)4 * STAGE3 * sizeof(double);             // 36 864 B: three workgroups per CU
    }
  }
}

template <bool TA, bool TB, int TRI>
static int launch_impl(lpgp_ctx* ctx, hipStream_t stream, const GemmArgs& g) {
  void (*kfn)(GemmArgs) = nullptr;
  size_t shmem = 0;
  {
    int64_t nt_est = (int64_t)g.mt * g.nt;               // (triangular shapes: the lower trapezoid; distributed staircase: an upper bound)
    if (TRI != 0 && !g.cyc) nt_est = (int64_t)g.nt * (g.nt + 1) / 2 + (int64_t)(g.mt - g.nt) * g.nt;
    pick_kernel<TA, TB, TRI>(ctx, nt_est, g.occ3 != 0, &kfn, &shmem);
  }
  LPGP_TRY_RC(ensure_lds_attr(ctx, reinterpret_cast<const void*>(kfn), shmem));
  GemmArgs ga = g;
  const int BANDR = ctx->gemm_band;
  ga.band = BANDR;
  if (ctx->dense_tiles && (g.mt + BANDR - 1) / BANDR <= GemmArgs::MAXB) {
    ga.dense = 1;
    ga.nbands = (g.mt + BANDR - 1) / BANDR;
    if (TRI) {
      LPGP_CHECK(g.cyc || g.mt >= g.nt, "gemm: triangular update needs mt >= nt");
      int acc = 0;
      for (int b = 0; b < ga.nbands; ++b) {
        ga.band_prefix[b] = acc;
        const int r0 = b * BANDR, R = std::min(BANDR, g.mt - r0);
        if (g.cyc) {
          acc += stair_band_count(g, r0, R);
        } else {
          acc += std::min(r0 + 1, g.nt) * R;
          for (int t = 0; t + 1 < R && r0 + 1 + t < g.nt; ++t) acc += R - 1 - t;
        }
      }
      ga.band_prefix[ga.nbands] = acc;
      ga.ntiles = acc;
    } else {
      ga.ntiles = g.mt * g.nt;
    }
    if (ga.ntiles == 0) return 0;                    // (distributed update: no valid tile in this rank's region)
    ga.chunk = (ga.ntiles + 7) / 8;
    hipLaunchKernelGGL(kfn, dim3((unsigned)(8 * ga.chunk)), dim3(256), shmem, stream, ga);
    LPGP_HIP(hipGetLastError());
    return 0;
  }
  LPGP_CHECK(!g.cyc, "gemm: the distributed update needs the dense tile enumeration");
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
  const int64_t nvirtual = (int64_t)((nsuper + 7) / 8) * 8 * SS;
  hipLaunchKernelGGL(kfn, dim3((unsigned)nvirtual), dim3(256), shmem, stream, ga);
  LPGP_HIP(hipGetLastError());
  return 0;
}

// Host replay of the staircase enumeration of a distributed update (tests: every valid local tile exactly once):
// out[2 i], out[2 i + 1] = (local tile row, local tile column) of list index i; returns the number of tiles.
int stair_enumerate_host(const GemmArgs& g, int32_t* out, int64_t cap) {
  const int nbands = (g.mt + BAND - 1) / BAND;
  int64_t n = 0;
  for (int b = 0; b < nbands; ++b) {
    const int r0 = b * BAND, R = std::min(BAND, g.mt - r0);
    const int cnt = stair_band_count(g, r0, R);
    for (int j = 0; j < cnt; ++j) {
      int r, c;
      stair_decode(g, r0, R, j, r, c);
      if (n < cap) { out[2 * n] = r; out[2 * n + 1] = c; }
      ++n;
    }
  }
  return (int)n;
}

// number of 128 x 128 tiles a launch would work on (triangular shapes: the lower trapezoid; distributed update: the staircase
// of this rank's valid tiles) -- what the schedulers price a trailing update with
int64_t gemm_valid_tiles(const GemmArgs& g) {
  if (g.mt <= 0 || g.nt <= 0) return 0;
  if (!g.tri) return (int64_t)g.mt * g.nt;
  if (!g.cyc) return (int64_t)g.nt * (g.nt + 1) / 2 + (int64_t)(g.mt - g.nt) * g.nt;
  int64_t n = 0;
  for (int r0 = 0; r0 < g.mt; r0 += BAND) n += stair_band_count(g, r0, std::min(BAND, g.mt - r0));
  return n;
}

int launch_gemm(lpgp_ctx* ctx, hipStream_t stream, int ta, int tb, const GemmArgs& g, int prof_kernel) {
  if (g.mt <= 0 || g.nt <= 0 || g.k <= 0) return 0;
  LPGP_CHECK(g.k % BK == 0, "gemm: k=%d not a multiple of %d", g.k, BK);
  LPGP_CHECK(!g.tri || (!ta && !tb), "gemm: the triangular update exists for the NT form only");
  // launches with fewer 128 x 128 tiles than CUs go to the 64 x 64-tile kernel (4x the workgroups)
  // (not for in-place products X <- X * B: with 64-column tiles another workgroup would still be
  //  reading the columns of X this one overwrites)
  const int64_t tiles = g.tri ? ((int64_t)g.nt * (g.nt + 1) / 2 + (int64_t)(g.mt - g.nt) * g.nt) : (int64_t)g.mt * g.nt;
  const bool small = tiles <= ctx->small_tiles_max && g.A != g.C && g.B != g.C && !g.cyc;
  if (prof_kernel >= 0) {
    // one profiling slot == one kernel symbol (family)
    if (small) prof_kernel = LPGP_K_GEMM_SMALL;
    else if (g.tri) prof_kernel = g.tri == 2 ? LPGP_K_SYRK_PANEL : (g.tri == 3 ? LPGP_K_SYRK_AHEAD : LPGP_K_SYRK);
    const double m = (double)g.mt * BM, n = (double)g.nt * BN, k = (double)g.k;
    // algorithmic flops: a symmetric update counts the lower trapezoid (m >= n) only
    double flops = g.tri ? 2.0 * k * (m * n - 0.5 * n * (n - 1.0)) : 2.0 * m * n * k;
    if (g.tri && g.cyc) {            // distributed update: the staircase of this rank's valid tiles (diagonal tiles count half)
      double tiles_valid = 0.0;
      for (int r0 = 0; r0 < g.mt; r0 += BAND) tiles_valid += stair_band_count(g, r0, std::min(BAND, g.mt - r0));
      flops = 2.0 * k * (double)BM * BN * tiles_valid;
    }
    prof_begin(ctx, stream, prof_kernel, flops, 0.0);
  }
  int rc;
  if (small) {
    if (g.tri == 2) rc = launch_small<false, false, 2>(ctx, stream, g);
    else if (g.tri) rc = launch_small<false, false, 1>(ctx, stream, g);
    else if (!ta && !tb) rc = launch_small<false, false, 0>(ctx, stream, g);
    else if (!ta && tb) rc = launch_small<false, true, 0>(ctx, stream, g);
    else if (ta && !tb) rc = launch_small<true, false, 0>(ctx, stream, g);
    else rc = launch_small<true, true, 0>(ctx, stream, g);
  } else if (g.tri == 2) rc = launch_impl<false, false, 2>(ctx, stream, g);
  else if (g.tri == 3) rc = launch_impl<false, false, 3>(ctx, stream, g);
  else if (g.tri) rc = launch_impl<false, false, 1>(ctx, stream, g);
  else if (!ta && !tb) rc = launch_impl<false, false, 0>(ctx, stream, g);
  else if (!ta && tb) rc = launch_impl<false, true, 0>(ctx, stream, g);
  else if (ta && !tb) rc = launch_impl<true, false, 0>(ctx, stream, g);
  else rc = launch_impl<true, true, 0>(ctx, stream, g);
  if (prof_kernel >= 0) prof_end(ctx, stream);
  return rc;
}

}  // namespace lpgp
