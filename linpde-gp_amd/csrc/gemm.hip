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
constexpr int LDK = 18;                  // [idx][LDK] image
constexpr int STAGE = BK * LDM;          // doubles per operand per stage (== 128*LDK)
static_assert(BK * LDM == BM * LDK, "both LDS images have the same size");

template <bool T>
__device__ __forceinline__ void load_tile(const double* __restrict__ P, int64_t ld, int64_t idx0,
                                          int64_t k0, int tid, v2f64 (&reg)[4]) {
  // tile: 128 (idx) x 16 (k)
  if constexpr (!T) {
    // idx fastest in memory: element (idx,k) at P[idx + k*ld]
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = q * 4 + (tid >> 6);
      const int i = (tid & 63) * 2;
      reg[q] = *reinterpret_cast<const v2f64*>(P + (idx0 + i) + (k0 + k) * ld);
    }
  } else {
    // k fastest in memory: element (idx,k) at P[k + idx*ld]
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = q * 32 + (tid >> 3);
      const int k = (tid & 7) * 2;
      reg[q] = *reinterpret_cast<const v2f64*>(P + (k0 + k) + (idx0 + i) * ld);
    }
  }
}

template <bool T>
__device__ __forceinline__ void store_tile(double* __restrict__ s, int tid, const v2f64 (&reg)[4]) {
  if constexpr (!T) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = q * 4 + (tid >> 6);
      const int i = (tid & 63) * 2;
      *reinterpret_cast<v2f64*>(s + k * LDM + i) = reg[q];
    }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = q * 32 + (tid >> 3);
      const int k = (tid & 7) * 2;
      *reinterpret_cast<v2f64*>(s + i * LDK + k) = reg[q];
    }
  }
}

template <bool T>
__device__ __forceinline__ double read_frag(const double* __restrict__ s, int idx_base, int ks, int lane) {
  // fragment element: idx = idx_base + (lane & 15), k = ks*4 + (lane >> 4)
  if constexpr (!T) return s[(ks * 4 + (lane >> 4)) * LDM + idx_base + (lane & 15)];
  else              return s[(idx_base + (lane & 15)) * LDK + ks * 4 + (lane >> 4)];
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  // ---- block -> tile (XCD-aware 8x8 super-tiles) ----
  // Blocks b, b+8, b+16, ... share an XCD (round-robin dispatch): each run of 64 of them
  // is one 8x8 super-tile.  For the triangular case only super-tiles on or below the
  // diagonal are enumerated, so every XCD gets the same number of them.
  const int SR = (g.mt + 7) >> 3, SC = (g.nt + 7) >> 3;
  const int b = blockIdx.x;
  const int xcd = b & 7, q = b >> 3;
  const int s = (q >> 6) * 8 + xcd, inner = q & 63;
  int sr, sc;
  if (g.tri) {
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
  const int tr = sr * 8 + (inner & 7), tc = sc * 8 + (inner >> 3);
  if (tr >= g.mt || tc >= g.nt) return;
  if (g.tri && tr < tc) return;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid & 1, wn = wid >> 1;
  const int64_t m0 = (int64_t)tr * BM, n0 = (int64_t)tc * BN;

  double* sA = smem;                  // 2 stages
  double* sB = smem + 2 * STAGE;

  v4f64 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};

  const int KT = g.k / BK;
  v2f64 ra[4], rb[4];
  load_tile<TA>(g.A, g.lda, m0, 0, tid, ra);
  load_tile<TB>(g.B, g.ldb, n0, 0, tid, rb);
  store_tile<TA>(sA, tid, ra);
  store_tile<TB>(sB, tid, rb);
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < KT; ++kt) {
    const bool more = (kt + 1 < KT);
    if (more) {
      load_tile<TA>(g.A, g.lda, m0, (int64_t)(kt + 1) * BK, tid, ra);
      load_tile<TB>(g.B, g.ldb, n0, (int64_t)(kt + 1) * BK, tid, rb);
    }
    const double* cA = sA + cur * STAGE;
    const double* cB = sB + cur * STAGE;
    double fa[2][4], fb[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[0][i] = read_frag<TA>(cA, wm * 64 + i * 16, 0, lane);
      fb[0][i] = read_frag<TB>(cB, wn * 64 + i * 16, 0, lane);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = ks & 1, nx = c ^ 1;
      if (ks < 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          fa[nx][i] = read_frag<TA>(cA, wm * 64 + i * 16, ks + 1, lane);
          fb[nx][i] = read_frag<TB>(cB, wn * 64 + i * 16, ks + 1, lane);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[c][j], fa[c][i], acc[i][j], 0, 0, 0);
    }
    if (more) {
      store_tile<TA>(sA + (cur ^ 1) * STAGE, tid, ra);
      store_tile<TB>(sB + (cur ^ 1) * STAGE, tid, rb);
    }
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: lane holds C[m = l&15][n = (l>>4) + 4r] of each 16x16 sub-tile ----
  const double alpha = g.alpha, beta = g.beta;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t n = n0 + wn * 64 + j * 16 + (lane >> 4) + 4 * r;
      double* col = g.C + n * g.ldc + m0 + wm * 64 + (lane & 15);
      if (beta == 0.0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) col[i * 16] = alpha * acc[i][j][r];
      } else {
        double old[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) old[i] = col[i * 16];
#pragma unroll
        for (int i = 0; i < 4; ++i) col[i * 16] = fma(alpha, acc[i][j][r], beta * old[i]);
      }
    }
  }
}

template <bool TA, bool TB>
static int launch_impl(lpgp_ctx* ctx, hipStream_t stream, const GemmArgs& g) {
  static bool attr_set = false;
  const size_t shmem = (size_t)4 * STAGE * sizeof(double);
  if (!attr_set) {
    LPGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f64_kernel<TA, TB>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    attr_set = true;
  }
  const int SR = (g.mt + 7) / 8, SC = (g.nt + 7) / 8;
  int nsuper = SR * SC;
  if (g.tri) {
    LPGP_CHECK(g.mt >= g.nt, "gemm: triangular update needs mt >= nt");
    nsuper = SC * (SC + 1) / 2 + (SR - SC) * SC;
  }
  const int64_t blocks = (int64_t)((nsuper + 7) / 8) * 8 * 64;
  hipLaunchKernelGGL((gemm_f64_kernel<TA, TB>), dim3((unsigned)blocks), dim3(256), shmem, stream, g);
  LPGP_HIP(hipGetLastError());
  return 0;
}

int launch_gemm(lpgp_ctx* ctx, hipStream_t stream, int ta, int tb, const GemmArgs& g, int prof_kernel) {
  if (g.mt <= 0 || g.nt <= 0 || g.k <= 0) return 0;
  LPGP_CHECK(g.k % BK == 0, "gemm: k=%d not a multiple of %d", g.k, BK);
  if (prof_kernel >= 0) {
    const double m = (double)g.mt * BM, n = (double)g.nt * BN, k = (double)g.k;
    // algorithmic flops: symmetric update counts the lower triangle only
    // algorithmic flops: a symmetric update counts the lower trapezoid (m >= n) only
    const double flops = g.tri ? 2.0 * k * (m * n - 0.5 * n * (n - 1.0)) : 2.0 * m * n * k;
    prof_begin(ctx, stream, prof_kernel, flops, 0.0);
  }
  int rc;
  if (!ta && !tb) rc = launch_impl<false, false>(ctx, stream, g);
  else if (!ta && tb) rc = launch_impl<false, true>(ctx, stream, g);
  else if (ta && !tb) rc = launch_impl<true, false>(ctx, stream, g);
  else rc = launch_impl<true, true>(ctx, stream, g);
  if (prof_kernel >= 0) prof_end(ctx, stream);
  return rc;
}

}  // namespace lpgp
