// The tile Cholesky (128 x 128 tile resident in LDS, factor + explicit inverse) as a device function: potrf_tile_kernel
// (potrf.hip) and the factor role of the resident panel-chain kernel (chain.hip) share it.
#pragma once

#include "lpgp_internal.h"
#include "kernel_util.h"

namespace lpgp {

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));

constexpr int TL = 136;                       // LDS leading dimension of the 128x128 tile (col-major)
constexpr int TILE_LDS_DOUBLES = TILE * TL + 8 * 256;
constexpr int TILE_WAVES = 8;                 // wave 0: diagonal blocks (the serial chain); the others: everything off it

__device__ __forceinline__ double bcast_lane(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// lane K of every 16-lane row of the wave, broadcast to that row (DPP row_newbcast:K, one v_mov_b64_dpp)
template <int K>
__device__ __forceinline__ double bcast16(double v) {
  static_assert(K >= 0 && K < 16, "row_newbcast lane");
  return __builtin_amdgcn_update_dpp(v, v, 0x150 + K, 0xF, 0xF, true);
}

#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)

// In-place Cholesky of one 128x128 SPD tile (lower, column-major, leading dim lda) and the
// explicit inverse of its factor (dense 128x128 column-major, zeros above the diagonal).
//
// One workgroup, tile resident in LDS.  16x16 blocks: the diagonal block is factored and
// inverted by one wave with one matrix row per lane (register resident, broadcasts by
// v_readlane, reciprocal square root instead of sqrt + divide), everything else is
// 16x16x16 block products on v_mfma_f64_4x4x4_4b_f64 (gemm.hip explains the choice of
// shape).  A 16x16 accumulator is four registers acc[q]: element (m = lane&15,
// n = 4q + (lane>>4)); per 4-deep k-step it takes ONE "m-side" fragment
// (lane -> M[m = lane&15][k = lane>>4]) and four replicated "n-side" fragments
// (lane -> N[n = 4q + (lane&3)][k = lane>>4]).
// where the single workgroup of the tile Cholesky ran (XCC id histogram; lpgp_debug_tile_xcc)
static __device__ int g_tile_xcc_hist[8];

// (a device function since round 5: the tile Cholesky kernel below and the factor role of the resident panel chain, chain.hip)
__device__ __forceinline__ void potrf_tile_body(double* __restrict__ a, int64_t lda, double* __restrict__ linv, int* __restrict__ info,
                                                int info_base, double* sm) {
  double* s = sm;                       // s[c*TL + r]
  double* sD = sm + TILE * TL;          // 8 diagonal-block inverses, column-major 16x16: Linv[r][c] at [c*16 + r]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r16 = lane & 15, g = lane >> 4, l3 = lane & 3;
  const int wu = __builtin_amdgcn_readfirstlane(wid);

#ifdef LPGP_TILE_STAMP
  unsigned long long ts_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tA_, tB_;
#define TSTAMP(i) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tB_) :: "memory"); ts_[i] += tB_ - tA_; tA_ = tB_; } while (0)
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tA_) :: "memory");
#else
#define TSTAMP(i) do { } while (0)
#endif
#ifdef LPGP_TILE_DIAG
  if (tid == 0) {                       // (diagnostic builds: which XCD the tile Cholesky ran on, lpgp_debug_tile_xcc)
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    atomicAdd(&g_tile_xcc_hist[xcc & 7], 1);
  }
#endif
  // ---- load tile: one 1-KiB LDS-DMA piece per column ----
  for (int c = wu; c < TILE; c += TILE_WAVES)
    __builtin_amdgcn_global_load_lds((gptr_t)(a + (int64_t)c * lda + 2 * lane), (lptr_t)(s + c * TL), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  TSTAMP(0);

  // ---- block products, shared by the phases below -------------------------------------------
  // trailing update of step jb:  A_ib,kb -= X_ib X_kb^T   (X = column block jb below the diagonal)
  auto update_pair = [&](int jb, int ib, int kb) {
    const int j0 = jb * 16;
    double acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = s[(kb * 16 + 4 * q + g) * TL + ib * 16 + r16];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const double bop = s[(j0 + 4 * ks + g) * TL + ib * 16 + r16];          //  X_ib[r=r16][k]
#pragma unroll
      for (int q = 0; q < 4; ++q)                                             // -X_kb[c=4q+l3][k]
        acc[q] = MFMA4(-s[(j0 + 4 * ks + g) * TL + kb * 16 + 4 * q + l3], bop, acc[q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) s[(kb * 16 + 4 * q + g) * TL + ib * 16 + r16] = acc[q];
  };
  // block (i, j), i > j, of the inverse:  X_ij = -Linv_i (L_ij Linv_j + sum_{j<k<i} L_ik X_kj);
  // X_ij is kept transposed in the strict upper triangle: X_ij[r][c] at s[(i16+r)*TL + j16+c]
  auto inverse_block = [&](int i, int j) {
    double S[4] = {0.0, 0.0, 0.0, 0.0};                  // S[q]: (m = 4q + g, c = r16)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const double bop = sD[j * 256 + r16 * 16 + 4 * ks + g];                 // Linv_j[t][c=r16]
#pragma unroll
      for (int q = 0; q < 4; ++q)                                             // L_ij[m=4q+l3][t]
        S[q] = MFMA4(s[(j * 16 + 4 * ks + g) * TL + i * 16 + 4 * q + l3], bop, S[q]);
    }
    for (int k = j + 1; k < i; ++k) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const double bop = s[(k * 16 + 4 * ks + g) * TL + j * 16 + r16];      // X_kj[t][c=r16]
#pragma unroll
        for (int q = 0; q < 4; ++q)                                           // L_ik[m=4q+l3][t]
          S[q] = MFMA4(s[(k * 16 + 4 * ks + g) * TL + i * 16 + 4 * q + l3], bop, S[q]);
      }
    }
    double X[4] = {0.0, 0.0, 0.0, 0.0};                  // X[u]: (r = 4u + g, c = r16)
#pragma unroll
    for (int q = 0; q < 4; ++q) {                         // contraction index m = 4q + g: S[q] is the m-side operand
#pragma unroll
      for (int u = 0; u < 4; ++u)                         // -Linv_i[r=4u+l3][m=4q+g]
        X[u] = MFMA4(-sD[i * 256 + (4 * q + g) * 16 + 4 * u + l3], S[q], X[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) s[(i * 16 + 4 * u + g) * TL + j * 16 + r16] = X[u];
  };

  // ---- write-out helpers: a finished part of the result leaves for HBM as soon as it is final,
  //      from the waves that are not on the critical path ----
  // block column cb of L (16 columns, zeros above the diagonal); wave slot wi of nw
  auto store_l_columns = [&](int cb, int wi, int nw) {
    for (int c = cb * 16 + wi; c < cb * 16 + 16; c += nw) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = lane + 64 * h;
        a[(int64_t)c * lda + r] = (r >= c) ? s[c * TL + r] : 0.0;
      }
    }
  };
  // block row i of Linv (16 rows x 128 columns): blocks left of the diagonal from the transposed
  // copies in the upper triangle (a strided, bank-conflicting LDS read -- harmless off the
  // critical path), the diagonal block from sD, zeros to the right; 4 columns x 16 rows per store
  auto store_linv_row = [&](int i, int wi, int nw) {
    const int rl = lane & 15, cl = lane >> 4;
    for (int c4 = wi; c4 < 32; c4 += nw) {
      const int c = c4 * 4 + cl, r = i * 16 + rl;
      double v;
      if (c < i * 16) v = s[r * TL + c];
      else if (c < i * 16 + 16) v = sD[i * 256 + (c & 15) * 16 + rl];
      else v = 0.0;
      linv[c * TILE + r] = v;
    }
  };

  // Pipeline over the eight 16-wide block columns.  The diagonal block (factor + inverse, a serial
  // pivot chain on ONE wave, ~60 % of the kernel when everything waits for it) overlaps with the
  // work that is not on the critical path: while wave 0 factors diagonal block jb, the other waves finish
  // the trailing update of step jb-1 (all block columns except jb, which phase C1 did) and row
  // jb-1 of the tile inverse.
  for (int jb = 0; jb < 8; ++jb) {
    const int j0 = jb * 16;
    // ---- (D) diagonal 16x16 block: factor + invert, wave 0, one matrix row per lane ----
    if (wu == 0) {
      double row[16];
      const int i = r16;
#pragma unroll
      for (int k = 0; k < 16; ++k) row[k] = s[(j0 + k) * TL + j0 + i];
      int bad = 0;
      // Factorisation and inversion share one pivot loop AND their broadcasts: lane c builds
      // column c of X = L^{-1} right-looking, x[k] -= L[k][j] x[j] as soon as x[j] is final, with
      // the very L[k][j] that the rank-1 update of the factor broadcasts at that moment.  (Forming
      // x[j] from all earlier L[j][k] at step j instead doubles the v_readlane count and keeps
      // 240 broadcast SGPRs alive: the compiler spilled 254 of them through v_writelane.)
      double x[16];
      const int c = r16;
#pragma unroll
      for (int j = 0; j < 16; ++j) x[j] = (j == c) ? 1.0 : 0.0;
      // Broadcasts by DPP `row_newbcast:k` (gfx90a+: lane k of every 16-lane row to the whole row, one 64-bit
      // v_mov_b64_dpp; the four rows of the wave hold the same matrix rows, so the result equals a wave-wide broadcast of
      // lane k): round 4.  Rounds 1-3 used two v_readlane_b32 per value -- a VGPR -> SGPR -> VALU round trip with its
      // wait states (113 s_nop in the listing) -- and the pivot step was bound by the ISSUE of that sequence (~80
      // instructions, ~450 cycles per pivot); the arithmetic, its order and its rounding are unchanged.
      static_for<0, 16>([&](auto J_) {
        constexpr int j = decltype(J_)::value;
        const double piv = bcast16<j>(row[j]);
        // (no test here: a pivot that is not positive turns into a NaN on the diagonal of L -- rsq of a negative number,
        //  0 * inf -- and every later pivot of the tile with it; the first one is found AFTER the loop, once per block.
        //  Rounds 1-3 tested and replaced the pivot inside the loop: a compare and eight selects per pivot on the wave
        //  whose instruction count IS the duration of the kernel.)
        // inv = piv^{-1/2}: hardware estimate + one Newton step; l = piv*inv refined once more; inv kept consistent with
        // the refined l.  (Round 3 tried scaling the column with the once-refined inv and refining only the diagonal entry,
        // off the pivot-to-pivot chain: four dependent fp64 operations fewer per pivot, 41 -> 39 us per tile.  The tile's own
        // backward error did not move (1.0e-15 of max |A| either way), but tests/test_gpu_random.py seed 107 -- a Gram
        // matrix of condition ~1e9 -- went from inside the 1e-8 bar to 1.3e-8 of the refined posterior with nothing else
        // changed: the last ulp of the column scaling is worth its 2 us.)
        double inv = __builtin_amdgcn_rsq(piv);
        inv = fma(inv, 0.5 * fma(-piv * inv, inv, 1.0), inv);
        double l = piv * inv;
        const double res = fma(-l, l, piv);
        l = fma(0.5 * inv, res, l);
        inv = fma(inv, -0.5 * inv * inv * res, inv);       // keep inv consistent with the refined l
        row[j] = (i == j) ? l : row[j] * inv;
        x[j] = x[j] * inv;                                  // (delta_jc - sum_{k<j} L[j][k] x[k]) / L[j][j]; exactly 0 for j < c
        static_for<j + 1, 16>([&](auto K_) {
          constexpr int k = decltype(K_)::value;
          const double lkj = bcast16<k>(row[j]);           // L[k][j]
          row[k] = fma(-row[j], lkj, row[k]);
          x[k] = fma(-lkj, x[j], x[k]);
          // pin both updates here: left alone, the compiler sinks all updates of x[k] and row[k]
          // down to pivot step k (their first use).
          // (Tried and measured equal or slower in rounds 1-3, all with v_readlane broadcasts: one instruction stream for
          // both recurrences with factor rows on lanes 0-15 and inverse columns on lanes 16-31; s_setprio for this wave;
          // keeping its SIMD free of background waves; two columns per round with their broadcasts issued back to back;
          // 4 x 4 micro-blocks (git 3de0b0c); factor on wave 0 and inverse on wave 1.)
          asm volatile("" : "+v"(x[k]), "+v"(row[k]));
        });
      });
      if (lane < 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          s[(j0 + k) * TL + j0 + i] = (k <= i) ? row[k] : 0.0;
          sD[jb * 256 + c * 16 + k] = x[k];
        }
      }
      {
        // first pivot of the block that was not positive: its diagonal entry of L is not a finite positive number
        const double dg = s[(j0 + i) * TL + j0 + i];
        const unsigned long long notpd = __builtin_amdgcn_ballot_w64(lane < 16 && !(dg > 0.0 && dg < 1.0e300));
        if (notpd != 0ull) bad = j0 + __builtin_ctzll(notpd) + 1;
      }
      if (bad && lane == 0) atomicCAS(info, 0, info_base + bad);
#ifdef LPGP_TILE_STAMP
      { unsigned long long tC_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tC_) :: "memory"); ts_[6] += tC_ - tA_; }
#endif
    } else if (jb >= 1) {
      // ---- background of step jb (waves 1 .. TILE_WAVES-1) ----
      const int wi = wu - 1;
#ifdef LPGP_TILE_STAMP
      unsigned long long tBg_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tBg_) :: "memory");
#endif
      // results that are final leave first (block column jb-1 of L, block row jb-2 of Linv): the
      // barrier at the end of the phase waits for outstanding stores, so they go out before the
      // arithmetic, not after it
      store_l_columns(jb - 1, wi, TILE_WAVES - 1);
      if (jb >= 2) store_linv_row(jb - 2, wi, TILE_WAVES - 1);
      // rest of the trailing update of step jb-1: block columns kb >= jb+1, dealt round-robin
      // (wave-uniform loop control: wi comes from the scalar wave index)
      {
        int turn = 0;
        for (int kb = jb + 1; kb < 8; ++kb)
          for (int ib = kb; ib < 8; ++ib) {
            if (turn == wi) update_pair(jb - 1, ib, kb);
            turn = (turn == TILE_WAVES - 2) ? 0 : turn + 1;
          }
      }
      // row jb-1 of the inverse (its diagonal inverse and all rows above it are complete)
      for (int j = wi; j < jb - 1; j += TILE_WAVES - 1) inverse_block(jb - 1, j);
#ifdef LPGP_TILE_STAMP
      if (tid == 64) { unsigned long long tC_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tC_) :: "memory"); g_stamps[8 + jb] = tC_ - tBg_; }
#endif
    }
    __syncthreads();
    TSTAMP(1);
    // ---- (B) panel below: X_ib = A_ib * D^-T  (in place), D the diagonal block just factored ----
    // A product with the explicit inverse of D, REFINED ONCE against D itself (round 4): X0 = A Dinv^T, R = A - X0 D^T,
    // X = X0 + R Dinv^T.  The bare product has a backward error of eps * cond(D) -- 70-90 x LAPACK's on tiles whose 16 x 16
    // diagonal blocks reach condition 1e5 (scratch/tile_chol_accuracy.py), and a survey of 400 randomised problems found the
    // device 10-37 x LAPACK's distance from the exact posterior on 4 % of them for exactly that reason (MEASUREMENTS.md);
    // with the step it is LAPACK's.  All three products stay in registers: the accumulator layout of a 16 x 16 block IS
    // the m-side operand layout of the next product's k-steps (acc[q] = element (r16, 4q + g) = m-fragment of k-step q), and
    // the four m-fragments of A are A in accumulator layout.
    for (int ib = jb + 1 + wu; ib < 8; ib += TILE_WAVES) {
      double acc[4] = {0.0, 0.0, 0.0, 0.0}, a4[4], dinv[4][4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        a4[ks] = s[(j0 + 4 * ks + g) * TL + ib * 16 + r16];                     // A[r=r16][k]        (m side)
#pragma unroll
        for (int q = 0; q < 4; ++q) {                                           // Dinv[c=4q+l3][k]   (n side)
          dinv[ks][q] = sD[jb * 256 + (4 * ks + g) * 16 + 4 * q + l3];
          acc[q] = MFMA4(dinv[ks][q], a4[ks], acc[q]);
        }
      }
      double rr[4] = {a4[0], a4[1], a4[2], a4[3]};                               // R = A - X0 D^T
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int q = 0; q < 4; ++q)                                             // -D[c=4q+l3][k=4ks+g]  (zeros above the diagonal are stored)
          rr[q] = MFMA4(-s[(j0 + 4 * ks + g) * TL + j0 + 4 * q + l3], acc[ks], rr[q]);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = MFMA4(dinv[ks][q], rr[ks], acc[q]);   // X = X0 + R Dinv^T
#pragma unroll
      for (int q = 0; q < 4; ++q) s[(j0 + 4 * q + g) * TL + ib * 16 + r16] = acc[q];
    }
    __syncthreads();
    TSTAMP(2);
    // ---- (C1) the part of the trailing update the next diagonal block and panel wait for:
    //      block column jb+1 ----
    if (jb + 1 < 8)
      for (int ib = jb + 1 + wu; ib < 8; ib += TILE_WAVES) update_pair(jb, ib, jb + 1);
    __syncthreads();
    TSTAMP(3);
  }
  // ---- last row of the inverse (needs the last diagonal inverse) ----
  for (int j = wu; j < 7; j += TILE_WAVES) inverse_block(7, j);
  __syncthreads();

  TSTAMP(4);
  // ---- what is left to write: the last block column of L, the last two block rows of Linv ----
  store_l_columns(7, wu, TILE_WAVES);
  store_linv_row(6, wu, TILE_WAVES);
  store_linv_row(7, wu, TILE_WAVES);
#ifdef LPGP_TILE_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TSTAMP(5);
  if (tid == 0) for (int i = 0; i < 7; ++i) g_stamps[i] = ts_[i];
#endif
}

}  // namespace lpgp
