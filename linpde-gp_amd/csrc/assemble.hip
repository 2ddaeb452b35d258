// Pairwise evaluation of differentiated covariance blocks (L0 k L1'^*)(X0, X1).
//
// Replaces the NumPy evaluation chain of the reference
//   covfuncs/linfuncops/diffops/_tensor_product.py:84-119  (_compute_res / _evaluate)
//   covfuncs/linfuncops/diffops/_matern.py:64-86,300-318,403-410,476-483,558-571
//   covfuncs/linfuncops/diffops/_expquad.py:45-57,106-122,187-201,280-312,390-410
//   functions/_polynomial.py:61-68 (Horner)
// (~40-60 full-matrix temporaries per block) by ONE fused kernel: point coordinates of
// the column tile staged in LDS, one exp per entry, one coalesced fp64 store per entry.
//
// Host part: lower_kdesc() turns the term list  sum_t c_t prod_d d^{n0} d'^{n1} k_d  into a
// per-parity-class dense polynomial in r_d = |a_d (x_d - x'_d)| (exact integer tables for
// the Matern derivative polynomials, _matern.py:613-639, and Hermite polynomials).

#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>

#include "lpgp_internal.h"
#include "eval_entries.h"

namespace lpgp {

// ---------------------------------------------------------------------------------------
// device kernel
// ---------------------------------------------------------------------------------------
constexpr int AT = 64;          // tile: 64 rows x 64 cols per workgroup (256 threads)

struct AsmArgs {
  const double* x0;             // SoA rows
  const double* x1;             // SoA cols
  int64_t n0, n1;               // valid rows / cols
  int64_t n0_pad, n1_pad;       // SoA strides
  double* out;                  // column-major
  int64_t ld;
  int64_t row_off, col_off;
  int32_t lower_only;           // symmetric diagonal block: skip tiles strictly above diagonal
  int32_t tiles_r, tiles_c;
  int32_t ct = 1;               // assemble_fast_kernel: column tiles per workgroup (consecutive tiles of one tile row)
  int32_t flags = 0;            // bit 0: per-point exponential factors (lpgp_ctx::asm_factors); measurement aids (LPGP_ASM_DIAG): bit 1 = no evaluation (stores only), bit 2 = no stores (evaluation only)
  Layout2D lay;                 // where element (row_off + i, col_off + j) lives on this rank (multi-GPU: only the owned tiles are written)
};

// A block ROW in one launch (round 5): up to ASM_MAXJ blocks that share ONE descriptor -- the cross blocks of a conditioning against
// the earlier blocks (crosscov/linfunctls/_evaluation.py:163-173, _conditional.py:270), the rows of the cross-covariance
// (_conditional.py:140-153) -- each with its own point sets and place in the output.  A workgroup finds its job by the prefix
// of workgroup counts (a scalar search over the kernel arguments) and then runs exactly the single-block code.
constexpr int ASM_MAXJ = 8;
struct AsmJobDev {
  const double* x0;
  const double* x1;
  int64_t n0, n1, n0_pad, n1_pad, row_off, col_off;
  int32_t lower_only, tiles_r, tiles_c, ct;
};
struct AsmBatch {
  int32_t njobs = 0;             // 0 / 1: the single block of AsmArgs
  int32_t wg0[ASM_MAXJ + 1] = {};
  AsmJobDev job[ASM_MAXJ];
};

// Per-point exponential factors of the Matern dimensions (eval_entries.h: `Fac`): for group g, dimension j and point p of
// the tile, E+ = e^{-a (x_p - x0)} and E- = e^{+a (x_p - x0)} with the tile's own origin x0 (its first column point), rows
// and columns in LDS: [g][j][sign][AT].  The arguments are carried in double-double (lpgp_exp_factors), so the accuracy of an
// entry does not depend on its distance from the origin; FACT_TMAX only keeps e^{+-t} far from overflow and the dropped
// second-order term (t_lo^2 ~ (32 eps)^2) negligible.  Tiles whose points spread further -- dozens of length scales inside
// one 64 x 64 tile -- fall back to one exp per entry.
constexpr double FACT_TMAX = 32.0;
constexpr int AEK = 4;          // entries per thread per pass in the kernels below (1 row x 4 columns; 16 columns per thread)

template <int D>
struct LdsFactors {
  static constexpr bool enabled = true;
  const double* rowf;           // row factors, already offset by this lane's row
  const double* colf;           // column factors, already offset by the first column of the pass
  __device__ __forceinline__ double pair(int g, int j, int e) const {
    const double* r = rowf + (size_t)((g * D + j) * 2) * AT;
    const double* c = colf + (size_t)((g * D + j) * 2) * AT + e;
    return fmin(r[0] * c[AT], r[AT] * c[0]);      // E+(row) E-(col)  vs  E-(row) E+(col)
  }
};

constexpr int FACT_MAXG = 4;     // the factor tables hold this many summands; sums with more take the table exponential

// factors of the point this lane stands for (coordinate xp[j], `valid` false: treated as the origin) for every Matern
// dimension of every product-form group; items (g, j) are dealt to the four waves.  Clears *fast if the range bound fails.
template <int D>
__device__ __forceinline__ void stage_factors(const DevDesc* __restrict__ desc, const double (&xp)[D], const double (&x0)[D], bool valid,
                                              double* dst /* [g][j][2][AT] */, int lane, int w, int wstep, int* fast) {
  const int nitems = desc->ngroups * D;
  for (int it = w; it < nitems; it += wstep) {
    const int g = it / D, j = it - g * D;
    const DevGroup& G = desc->g[g];
    if (G.iso || G.expkind[j] != 1) continue;
    double x = 0.0, o = 0.0;
#pragma unroll
    for (int jj = 0; jj < D; ++jj)
      if (jj == j) { x = xp[jj]; o = x0[jj]; }
    double ep, em, tabs;
    lpgp_exp_factors(G.a[j], valid ? x : o, o, ep, em, tabs);
    if (!(tabs <= FACT_TMAX)) *fast = 0;
    dst[(size_t)((g * D + j) * 2) * AT + lane] = ep;
    dst[(size_t)((g * D + j) * 2 + 1) * AT + lane] = em;
  }
}

// FACTORS: the per-point exponential factors (`asm_factors`, off by default) need 2 x FACT_MAXG * D * 2 * 64 doubles of LDS (32 KB
// at D = 4); the default instantiation carries the 4-KB exponential table only (the generic kernel is the one that
// takes everything the specialised kernels reject -- D >= 3, several groups, isotropic -- and ran at 4 / 3 workgroups per CU)
template <int D, bool FACTORS>
__global__ __launch_bounds__(256) void assemble_kernel(const DevDesc* __restrict__ desc, AsmArgs a) {
  __shared__ double sx1[D][AT];
  __shared__ double sfr[FACTORS ? FACT_MAXG * D * 2 * AT : 1], sfc[FACTORS ? FACT_MAXG * D * 2 * AT : 1];
  __shared__ __attribute__((aligned(16))) double s_exp[2 * EXP_TAB_N];      // table of lpgp_exp_neg (eval_entries.h)
  __shared__ int s_fast;
  const int tr = blockIdx.x % a.tiles_r;   // row tile fastest: consecutive blocks write neighbouring rows
  const int tc = blockIdx.x / a.tiles_r;
  const int64_t r0 = (int64_t)tr * AT, c0 = (int64_t)tc * AT;
  if (a.lower_only && c0 > r0 + AT - 1) return;
  // a 64 x 64 tile lies inside ONE 128-tile of the padded matrix (offsets are multiples of 128): one owner
  const int64_t lrow0 = cyc_local(a.lay.rows, a.row_off + r0), lcol0 = cyc_local(a.lay.cols, a.col_off + c0);
  if (lrow0 < 0 || lcol0 < 0) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wu = __builtin_amdgcn_readfirstlane(w);
  // column coordinates: every wave reads them (cached), wave 0 stages them for the entry loop
  const int64_t col = c0 + lane;
  const int64_t row = r0 + lane;
  double xr[D], xc[D], x0[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    xr[j] = (row < a.n0) ? a.x0[j * a.n0_pad + row] : 0.0;
    xc[j] = (col < a.n1) ? a.x1[j * a.n1_pad + col] : 0.0;
    x0[j] = a.x1[j * a.n1_pad + c0];                     // the tile's origin: its first column point (c0 < n1)
  }
  if (threadIdx.x < AT) {
#pragma unroll
    for (int j = 0; j < D; ++j) sx1[j][threadIdx.x] = xc[j];
  }
  if (threadIdx.x == 0) s_fast = FACTORS ? (a.flags & 1) : 0;
  s_exp[threadIdx.x] = g_exp_table[threadIdx.x], s_exp[threadIdx.x + 256] = g_exp_table[threadIdx.x + 256];         // (256 threads, 256 entries)
  const ExpTab etab{s_exp};
  __syncthreads();
  if (FACTORS && (a.flags & 1)) {
    // waves 0,1: row factors; waves 2,3: column factors
    if (wu < 2) stage_factors<D>(desc, xr, x0, row < a.n0, sfr, lane, wu, 2, &s_fast);
    else stage_factors<D>(desc, xc, x0, col < a.n1, sfc, lane, wu - 2, 2, &s_fast);
    __syncthreads();
  }
  const bool fast = FACTORS && s_fast != 0;
  double sink = 0.0;
#pragma unroll 1
  for (int pass = 0; pass < 16 / AEK; ++pass) {
    const int cb = w * 16 + pass * AEK;
    double dx[D][AEK], res[AEK];
#pragma unroll
    for (int j = 0; j < D; ++j)
#pragma unroll
      for (int e = 0; e < AEK; ++e) dx[j][e] = xr[j] - sx1[j][cb + e];
    if (a.flags & 2) {
#pragma unroll
      for (int e = 0; e < AEK; ++e) res[e] = dx[0][e];
    } else if (fast) {
      LdsFactors<D> fac{sfr + lane, sfc + cb};
      eval_entries<D, AEK, LdsFactors<D>>(desc, dx, res, etab, fac);
    } else {
      eval_entries<D, AEK>(desc, dx, res, etab);
    }
    if (a.flags & 4) {
#pragma unroll
      for (int e = 0; e < AEK; ++e) sink += res[e];
    } else if (row < a.n0) {
#pragma unroll
      for (int e = 0; e < AEK; ++e) {
        int64_t c = c0 + cb + e;
        if (c < a.n1) a.out[(lrow0 + lane) + (lcol0 + cb + e) * a.ld] = res[e];
      }
    }
  }
  if ((a.flags & 4) && sink == 0.12345 && row < a.n0) a.out[lrow0 + lane + lcol0 * a.ld] = sink;
}

// ---------------------------------------------------------------------------------------
// The common descriptor shapes, specialised (round 3).  assemble_kernel above is written once for every D <= 4, any
// number of groups, parity classes and degrees: its evaluation is a nest of tiny run-time loops (4 FMAs between a scalar
// compare-and-branch per coefficient, scalar loads of group parameters per pass), and on the 4096 x 16384 cross-covariance
// of c3 it spends 0.195 ms evaluating against 0.110 ms for the stores and the tile set-up.  Almost
// every block of the BASELINE workloads has ONE product-form group, D <= 2, at most two parity classes and degrees <= 4
// per dimension: for those the polynomial degrees become template parameters (fully unrolled nested Horner), the
// coefficients, scales and exponent kinds travel BY VALUE in the kernel arguments (scalar registers for the whole
// kernel: no load inside the entry loop), eight entries per thread run side by side, and the store address advances by one
// add per entry.  The arithmetic per entry is the generic kernel's, operation for operation -- same association, same
// exp -- so both paths give bit-identical blocks (tests/test_gpu_kernels.py::test_specialised_assembly_is_bit_identical).
// ---------------------------------------------------------------------------------------
// (arguments of the matrix-free product kernels, below)
constexpr int MV_R = MV_RHS;

struct MvArgs {
  const double* x0;
  const double* x1;
  int64_t n0, n1, n0_pad, n1_pad;
  const double* v;               // device, [r][v_stride]
  int64_t v_stride;              // doubles between right-hand sides of v (>= n1)
  double* part;                  // device, [split][r][n0_pad]
  int32_t nr;                    // right-hand sides in this pass (<= MV_R)
  int32_t factors;               // per-point exponential factors (lpgp_ctx::asm_factors)
  int32_t tiles_r, tiles_c, splits;
};

constexpr int FAST_MAXC = 32;   // coefficient doubles over all classes
struct FastDesc {
  double scale;
  double a[2];
  int32_t kind[2];              // 1: exp(-r), 2: exp(-r^2/2)
  int32_t ncls;
  int32_t parity[2];
  double coef[FAST_MAXC];       // class c at c * N0 * N1: [i0 * N1 + i1]
};

// FE entries of one row: columns cb .. cb + FE - 1 of the staged column tile.  res[e] = scale * exp(-expo) * sum_c sgn Poly_c
// LIN: every dimension decays like e^{-r} (a product of Matern factors): the run-time choice between r and r^2/2 per entry
// and dimension (two multiplies and a 64-bit select) is compiled out.  EVEN (only with LIN): no parity class flips a sign
// (even derivative orders in every dimension, e.g. Laplacians): the sign extraction and the sign products are compiled out.
// Both only remove operations whose result the general form discards, so all three forms give the same bits.
template <int D, int N0, int N1, int FE, bool LIN, bool EVEN>
__device__ __forceinline__ void fast_entries(const FastDesc& fd, const double (&xr)[D], const double (*sx1)[AT], int cb,
                                             const ExpTab& etab, double (&res)[FE]) {
  const unsigned m0 = (fd.parity[0] & 1) ? 0x80000000u : 0u, m1 = (D > 1 && (fd.parity[0] & 2)) ? 0x80000000u : 0u;
  const unsigned n0m = (fd.parity[1] & 1) ? 0x80000000u : 0u, n1m = (D > 1 && (fd.parity[1] & 2)) ? 0x80000000u : 0u;
  double r[D][FE], expo[FE], tot[FE];
  unsigned sg[D][FE];
#pragma unroll
  for (int e = 0; e < FE; ++e) expo[e] = 0.0;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const double aj = fd.a[j];
    const bool lin = LIN || fd.kind[j] == 1;
#pragma unroll
    for (int e = 0; e < FE; ++e) {
      const double v = aj * (xr[j] - sx1[j][cb + e]);
      sg[j][e] = EVEN ? 0u : (lpgp_hi32(v) & 0x80000000u);
      r[j][e] = fabs(v);
      expo[e] += lin ? r[j][e] : 0.5 * r[j][e] * r[j][e];
    }
  }
#pragma unroll
  for (int e = 0; e < FE; ++e) tot[e] = 0.0;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    if (c < fd.ncls) {
      const double* cf = fd.coef + c * N0 * N1;
      double acc0[FE];
#pragma unroll
      for (int e = 0; e < FE; ++e) acc0[e] = 0.0;
#pragma unroll
      for (int i0 = N0 - 1; i0 >= 0; --i0) {
        if constexpr (D == 1) {
#pragma unroll
          for (int e = 0; e < FE; ++e) acc0[e] = fma(acc0[e], r[0][e], cf[i0]);
        } else {
          double acc1[FE];
#pragma unroll
          for (int e = 0; e < FE; ++e) acc1[e] = 0.0;
#pragma unroll
          for (int i1 = N1 - 1; i1 >= 0; --i1)
#pragma unroll
            for (int e = 0; e < FE; ++e) acc1[e] = fma(acc1[e], r[D - 1][e], cf[i0 * N1 + i1]);
#pragma unroll
          for (int e = 0; e < FE; ++e) acc0[e] = fma(acc0[e], r[0][e], acc1[e]);
        }
      }
      const unsigned q0 = c == 0 ? m0 : n0m, q1 = c == 0 ? m1 : n1m;
#pragma unroll
      for (int e = 0; e < FE; ++e) {
        if constexpr (EVEN) {
          tot[e] += acc0[e];
        } else {
          unsigned sgn = sg[0][e] & q0;
          if constexpr (D > 1) sgn ^= sg[D - 1][e] & q1;
          tot[e] += lpgp_xor_sign(acc0[e], sgn);
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < FE; ++e) res[e] = fma(fd.scale * lpgp_exp_neg(expo[e], etab), tot[e], 0.0);
}

template <int D, int N0, int N1, int MODE>       // MODE 0: general, 1: LIN, 2: LIN + EVEN (fast_entries)
__global__ __launch_bounds__(256) void assemble_fast_kernel(FastDesc fd, AsmArgs a, AsmBatch bt) {
  constexpr int FE = 8;         // entries per thread per pass
  __shared__ double sx1[2][D][AT];
  __shared__ __attribute__((aligned(16))) double s_exp[2 * EXP_TAB_N];
  int bx = blockIdx.x;
  if (bt.njobs > 1) {
    // which block of the row this workgroup belongs to (uniform: scalar loads from the kernel arguments)
    int j = 0;
    while (j + 1 < bt.njobs && bx >= bt.wg0[j + 1]) ++j;
    bx -= bt.wg0[j];
    const AsmJobDev& jb = bt.job[j];
    a.x0 = jb.x0; a.x1 = jb.x1; a.n0 = jb.n0; a.n1 = jb.n1; a.n0_pad = jb.n0_pad; a.n1_pad = jb.n1_pad;
    a.row_off = jb.row_off; a.col_off = jb.col_off; a.lower_only = jb.lower_only;
    a.tiles_r = jb.tiles_r; a.tiles_c = jb.tiles_c; a.ct = jb.ct;
  }
  // a workgroup owns `ct` consecutive column tiles of one tile row: the row coordinates and the exponential's table (4 KB
  // against 32 KB of output per tile) are set up once, the column coordinates of the next tile are staged while this one
  // is evaluated
  const int tr = bx % a.tiles_r;
  const int tc0 = (bx / a.tiles_r) * a.ct;
  const int tc1 = tc0 + a.ct < a.tiles_c ? tc0 + a.ct : a.tiles_c;
  const int64_t r0 = (int64_t)tr * AT;
  if (a.lower_only && (int64_t)tc0 * AT > r0 + AT - 1) return;
  const int64_t lrow0 = cyc_local(a.lay.rows, a.row_off + r0);
  if (lrow0 < 0) return;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  auto stage_cols = [&](int tc, int buf) {
    if (threadIdx.x < AT) {
      const int64_t c = (int64_t)tc * AT + threadIdx.x;
#pragma unroll
      for (int j = 0; j < D; ++j) sx1[buf][j][threadIdx.x] = (c < a.n1) ? a.x1[j * a.n1_pad + c] : 0.0;
    }
  };
  stage_cols(tc0, 0);
  s_exp[threadIdx.x] = g_exp_table[threadIdx.x], s_exp[threadIdx.x + 256] = g_exp_table[threadIdx.x + 256];
  const ExpTab etab{s_exp};
  const int64_t row = r0 + lane;
  double xr[D];
#pragma unroll
  for (int j = 0; j < D; ++j) xr[j] = (row < a.n0) ? a.x0[j * a.n0_pad + row] : 0.0;
  for (int tc = tc0; tc < tc1; ++tc) {
    const int buf = (tc - tc0) & 1;
    __syncthreads();                                   // this tile's coordinates staged; the other buffer's readers are done
    if (tc + 1 < tc1) stage_cols(tc + 1, buf ^ 1);
    const int64_t c0 = (int64_t)tc * AT;
    if (a.lower_only && c0 > r0 + AT - 1) break;       // (tiles further right lie above the diagonal too)
    const int64_t lcol0 = cyc_local(a.lay.cols, a.col_off + c0);
    if (lcol0 < 0) continue;
#pragma unroll 1
    for (int pass = 0; pass < 16 / FE; ++pass) {
      const int cb = w * 16 + pass * FE;
      double res[FE];
      fast_entries<D, N0, N1, FE, MODE >= 1, MODE == 2>(fd, xr, sx1[buf], cb, etab, res);
      if (row < a.n0) {
        double* op = a.out + (lrow0 + lane) + (lcol0 + cb) * a.ld;
        const int ncols = (int)(a.n1 - c0 < AT ? a.n1 - c0 : AT);        // valid columns of this tile: a scalar 32-bit compare per entry
#pragma unroll
        for (int e = 0; e < FE; ++e) {
          if (cb + e < ncols) *op = res[e];
          op += a.ld;
        }
      }
    }
  }
}

// the matrix-free product on the same specialised evaluation (matvec_kernel's structure: 64 rows per workgroup, a range of
// column tiles, MV_R right-hand sides per evaluation, partial sums per split)
template <int D, int N0, int N1, int MODE>
__global__ __launch_bounds__(256) void matvec_fast_kernel(FastDesc fd, MvArgs a) {
  constexpr int FE = 8;
  __shared__ double sx1[D][AT];
  __shared__ __attribute__((aligned(16))) double s_exp[2 * EXP_TAB_N];
  __shared__ double sv[MV_R][AT];
  __shared__ double red[3][MV_R][AT];
  const int tr = blockIdx.x % a.tiles_r, sp = blockIdx.x / a.tiles_r;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t row = (int64_t)tr * AT + lane;
  s_exp[threadIdx.x] = g_exp_table[threadIdx.x], s_exp[threadIdx.x + 256] = g_exp_table[threadIdx.x + 256];        // (visible behind the first barrier of the tile loop)
  const ExpTab etab{s_exp};
  double xr[D];
#pragma unroll
  for (int j = 0; j < D; ++j) xr[j] = (row < a.n0) ? a.x0[j * a.n0_pad + row] : 0.0;
  double y[MV_R];
#pragma unroll
  for (int r = 0; r < MV_R; ++r) y[r] = 0.0;
  const int per = (a.tiles_c + a.splits - 1) / a.splits;
  const int tc_end = (sp + 1) * per < a.tiles_c ? (sp + 1) * per : a.tiles_c;
  for (int tc = sp * per; tc < tc_end; ++tc) {
    __syncthreads();
    {
      const int64_t c = (int64_t)tc * AT + lane;
      for (int j = w; j < D; j += 4) sx1[j][lane] = (c < a.n1) ? a.x1[j * a.n1_pad + c] : 0.0;
      for (int r = w; r < MV_R; r += 4) sv[r][lane] = (c < a.n1 && r < a.nr) ? a.v[(int64_t)r * a.v_stride + c] : 0.0;
    }
    __syncthreads();
#pragma unroll 1
    for (int pass = 0; pass < 16 / FE; ++pass) {
      const int cb = w * 16 + pass * FE;
      double res[FE];
      fast_entries<D, N0, N1, FE, MODE >= 1, MODE == 2>(fd, xr, sx1, cb, etab, res);
#pragma unroll
      for (int e = 0; e < FE; ++e)
#pragma unroll
        for (int r = 0; r < MV_R; ++r) y[r] = fma(res[e], sv[r][cb + e], y[r]);
    }
  }
  __syncthreads();
  if (w > 0) {
#pragma unroll
    for (int r = 0; r < MV_R; ++r) red[w - 1][r][lane] = y[r];
  }
  __syncthreads();
  if (w == 0 && row < a.n0) {
#pragma unroll
    for (int r = 0; r < MV_R; ++r)
      if (r < a.nr) a.part[((int64_t)sp * MV_R + r) * a.n0_pad + row] = ((y[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane];
  }
}

// host: does the lowered descriptor have the shape assemble_fast_kernel covers?  (D <= 2, one product-form group, <= 2
// parity classes, <= 5 coefficients per dimension, <= FAST_MAXC coefficients in total)
static bool fast_shape(const DevDesc& d, FastDesc* fd, int* n0, int* n1) {
  if (d.d < 1 || d.d > 2 || d.ngroups != 1) return false;
  const DevGroup& G = d.g[0];
  if (G.iso || G.ncls < 1 || G.ncls > 2) return false;
  const int N0 = G.deg[0] + 1, N1 = d.d > 1 ? G.deg[1] + 1 : 1;
  if (N0 < 1 || N0 > 5 || N1 < 1 || N1 > 5 || G.ncls * N0 * N1 > FAST_MAXC) return false;
  fd->scale = G.scale;
  fd->ncls = G.ncls;
  for (int j = 0; j < 2; ++j) {
    fd->a[j] = j < d.d ? G.a[j] : 0.0;
    fd->kind[j] = j < d.d ? G.expkind[j] : 1;
    fd->parity[j] = j < G.ncls ? G.parity[j] : 0;
  }
  for (int i = 0; i < FAST_MAXC; ++i) fd->coef[i] = 0.0;
  for (int c = 0; c < G.ncls; ++c)
    for (int i = 0; i < N0 * N1; ++i) fd->coef[c * N0 * N1 + i] = d.coef[G.coef_off[c] + i];
  *n0 = N0;
  *n1 = N1;
  return true;
}

// dispatch on the polynomial sizes (N0, N1 in 1..5; N1 = 1 for D = 1): KIND 0 = assembly, 1 = matrix-free product
// which of fast_entries' forms a descriptor allows: 2 = every dimension e^{-r} and no sign-flipping parity class, 1 = every
// dimension e^{-r}, 0 = anything
static int fast_mode(const FastDesc& fd, int d) {
  for (int j = 0; j < d; ++j)
    if (fd.kind[j] != 1) return 0;
  const int mask = (1 << d) - 1;
  for (int c = 0; c < fd.ncls; ++c)
    if (fd.parity[c] & mask) return 1;
  return 2;
}
static thread_local const AsmBatch* g_asm_batch = nullptr;      // the job table of the assembly launch being dispatched (launch_fast below)
template <int KIND, int D, int N0, int N1, int MODE, class Args>
static void launch_fast_mode(dim3 grid, hipStream_t stream, const FastDesc& fd, const Args& a) {
  if constexpr (KIND == 0) {
    static const AsmBatch none{};
    hipLaunchKernelGGL((assemble_fast_kernel<D, N0, N1, MODE>), grid, dim3(256), 0, stream, fd, a, g_asm_batch ? *g_asm_batch : none);
  } else {
    hipLaunchKernelGGL((matvec_fast_kernel<D, N0, N1, MODE>), grid, dim3(256), 0, stream, fd, a);
  }
}
template <int KIND, int D, int N0, int N1, class Args>
static void launch_fast_one(dim3 grid, hipStream_t stream, const FastDesc& fd, const Args& a) {
  switch (fast_mode(fd, D)) {
    case 2: launch_fast_mode<KIND, D, N0, N1, 2>(grid, stream, fd, a); break;
    case 1: launch_fast_mode<KIND, D, N0, N1, 1>(grid, stream, fd, a); break;
    default: launch_fast_mode<KIND, D, N0, N1, 0>(grid, stream, fd, a); break;
  }
}
template <int KIND, int D, int N0, class Args>
static void launch_fast_n0(int n1, dim3 grid, hipStream_t stream, const FastDesc& fd, const Args& a) {
  if constexpr (D == 1) {
    launch_fast_one<KIND, 1, N0, 1>(grid, stream, fd, a);
  } else {
    switch (n1) {
      case 1: launch_fast_one<KIND, D, N0, 1>(grid, stream, fd, a); break;
      case 2: launch_fast_one<KIND, D, N0, 2>(grid, stream, fd, a); break;
      case 3: launch_fast_one<KIND, D, N0, 3>(grid, stream, fd, a); break;
      case 4: launch_fast_one<KIND, D, N0, 4>(grid, stream, fd, a); break;
      default: launch_fast_one<KIND, D, N0, 5>(grid, stream, fd, a); break;
    }
  }
}
template <int KIND, int D, class Args>
static void launch_fast(int n0, int n1, dim3 grid, hipStream_t stream, const FastDesc& fd, const Args& a) {
  switch (n0) {
    case 1: launch_fast_n0<KIND, D, 1>(n1, grid, stream, fd, a); break;
    case 2: launch_fast_n0<KIND, D, 2>(n1, grid, stream, fd, a); break;
    case 3: launch_fast_n0<KIND, D, 3>(n1, grid, stream, fd, a); break;
    case 4: launch_fast_n0<KIND, D, 4>(n1, grid, stream, fd, a); break;
    default: launch_fast_n0<KIND, D, 5>(n1, grid, stream, fd, a); break;
  }
}

__global__ void add_diag_kernel(double* a, int64_t ld, int64_t off, int64_t n, const double* v, double scalar, Layout2D lay) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t lr = cyc_local(lay.rows, off + i), lc = cyc_local(lay.cols, off + i);
  if (lr >= 0 && lc >= 0) a[lr + lc * ld] += (v ? v[i] : 0.0) + scalar;
}

__global__ void add_dense_lower_kernel(double* a, int64_t ld, int64_t off, int64_t n, const double* b /* n x n C-order */,
                                       Layout2D lay) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;   // row
  int64_t j = blockIdx.y;                                       // col
  if (i >= n || j > i) return;
  const int64_t lr = cyc_local(lay.rows, off + i), lc = cyc_local(lay.cols, off + j);
  if (lr >= 0 && lc >= 0) a[lr + lc * ld] += b[i * n + j];
}

// Copy a lowered descriptor into the next slot of the context's ring (pinned host -> device,
// asynchronous); the caller records slot.done behind the kernel that reads it.
static int stage_desc(lpgp_ctx* ctx, hipStream_t stream, const DevDesc& host_desc, lpgp_ctx::DescSlot** out) {
  lpgp_ctx::DescSlot& slot = ctx->desc_ring[ctx->desc_next];
  ctx->desc_next = (ctx->desc_next + 1) % lpgp_ctx::DESC_RING;
  if (slot.used) LPGP_HIP(hipEventSynchronize(slot.done));
  // only the used prefix of the coefficient table travels
  int ncoef = 0;
  for (int gi = 0; gi < host_desc.ngroups; ++gi) {
    const DevGroup& G = host_desc.g[gi];
    for (int c = 0; c < G.ncls; ++c) {
      int len = 1;
      for (int dd = 0; dd < host_desc.d; ++dd) len *= G.deg[dd] + 1;
      if (G.coef_off[c] + len > ncoef) ncoef = G.coef_off[c] + len;
    }
  }
  const size_t bytes = offsetof(DevDesc, coef) + (size_t)ncoef * sizeof(double);
  std::memcpy(slot.h, &host_desc, bytes);
  LPGP_HIP(hipMemcpyAsync(slot.d, slot.h, bytes, hipMemcpyHostToDevice, stream));
  *out = &slot;
  return 0;
}

int launch_assemble(lpgp_ctx* ctx, hipStream_t stream, const DevDesc& host_desc, const double* x0,
                    int64_t n0, int64_t n0_pad, const double* x1, int64_t n1, int64_t n1_pad,
                    double* out, int64_t ld, int64_t row_off, int64_t col_off, int lower_only,
                    const Layout2D& lay) {
  AsmArgs a;
  a.x0 = x0; a.x1 = x1; a.n0 = n0; a.n1 = n1; a.n0_pad = n0_pad; a.n1_pad = n1_pad;
  a.out = out; a.ld = ld; a.row_off = row_off; a.col_off = col_off; a.lower_only = lower_only;
  a.lay = lay;
  a.tiles_r = (int)((n0 + AT - 1) / AT);
  a.tiles_c = (int)((n1 + AT - 1) / AT);
  if (a.tiles_r == 0 || a.tiles_c == 0) return 0;
  static const int diag = [] { const char* e = std::getenv("LPGP_ASM_DIAG"); return e ? std::atoi(e) : 0; }();
  a.flags = ((ctx->asm_factors && host_desc.ngroups <= FACT_MAXG) ? 1 : 0) | ((diag & 3) << 1);
  dim3 grid((unsigned)((int64_t)a.tiles_r * a.tiles_c));
  double entries = lower_only ? 0.5 * (double)n0 * ((double)n0 + 1.0) : (double)n0 * (double)n1;
  {
    // the common shapes on the specialised kernel (descriptor by value: no staging, no slot)
    FastDesc fd;
    int N0 = 0, N1 = 0;
    if (ctx->asm_fast && a.flags == 0 && fast_shape(host_desc, &fd, &N0, &N1)) {
      prof_begin(ctx, stream, LPGP_K_ASSEMBLE, 0.0, 8.0 * entries);
      // column tiles per workgroup: as many as leave >= 16 workgroups per CU in the launch (at most ctx->asm_ct)
      // (rectangular blocks only: on a lower-triangle launch the groups that straddle the diagonal do one to four tiles and
      //  the launch loses its balance -- Gram block of 16 384 scattered points: 3.6 TB/s with four tiles per workgroup, 4.3 with
      //  one; scratch/assemble_lower.py)
      int ct = 1;
      while (!lower_only && ct < ctx->asm_ct && (int64_t)a.tiles_r * ((a.tiles_c + 2 * ct - 1) / (2 * ct)) >= 16 * (int64_t)(ctx->cus > 0 ? ctx->cus : 256)) ct *= 2;
      a.ct = ct;
      grid = dim3((unsigned)((int64_t)a.tiles_r * ((a.tiles_c + ct - 1) / ct)));
      if (host_desc.d == 1) launch_fast<0, 1>(N0, N1, grid, stream, fd, a);
      else launch_fast<0, 2>(N0, N1, grid, stream, fd, a);
      prof_end(ctx, stream);
      LPGP_HIP(hipGetLastError());
      return 0;
    }
  }
  lpgp_ctx::DescSlot* slotp = nullptr;
  int rc_ = stage_desc(ctx, stream, host_desc, &slotp);
  if (rc_ != 0) return rc_;
  lpgp_ctx::DescSlot& slot = *slotp;
  const DevDesc* d_desc = slot.d;
  prof_begin(ctx, stream, LPGP_K_ASSEMBLE, 0.0, 8.0 * entries);
  switch (host_desc.d) {
    case 1: if (a.flags & 1) hipLaunchKernelGGL((assemble_kernel<1, true>), grid, dim3(256), 0, stream, d_desc, a); else hipLaunchKernelGGL((assemble_kernel<1, false>), grid, dim3(256), 0, stream, d_desc, a); break;
    case 2: if (a.flags & 1) hipLaunchKernelGGL((assemble_kernel<2, true>), grid, dim3(256), 0, stream, d_desc, a); else hipLaunchKernelGGL((assemble_kernel<2, false>), grid, dim3(256), 0, stream, d_desc, a); break;
    case 3: if (a.flags & 1) hipLaunchKernelGGL((assemble_kernel<3, true>), grid, dim3(256), 0, stream, d_desc, a); else hipLaunchKernelGGL((assemble_kernel<3, false>), grid, dim3(256), 0, stream, d_desc, a); break;
    case 4: if (a.flags & 1) hipLaunchKernelGGL((assemble_kernel<4, true>), grid, dim3(256), 0, stream, d_desc, a); else hipLaunchKernelGGL((assemble_kernel<4, false>), grid, dim3(256), 0, stream, d_desc, a); break;
    default: LPGP_CHECK(false, "assemble: d=%d", host_desc.d);
  }
  prof_end(ctx, stream);
  LPGP_HIP(hipGetLastError());
  LPGP_HIP(hipEventRecord(slot.done, stream));
  slot.used = true;
  return 0;
}

// column tiles per workgroup of a rectangular block (see launch_assemble)
static int fast_ct(const lpgp_ctx* ctx, int tiles_r, int tiles_c, int lower_only) {
  int ct = 1;
  while (!lower_only && ct < ctx->asm_ct && (int64_t)tiles_r * ((tiles_c + 2 * ct - 1) / (2 * ct)) >= 16 * (int64_t)(ctx->cus > 0 ? ctx->cus : 256)) ct *= 2;
  return ct;
}

bool assemble_same_fast(const DevDesc& p, const DevDesc& q) {
  FastDesc fp, fq;
  std::memset(&fp, 0, sizeof(fp));          // (the comparison below is over the bytes, padding included)
  std::memset(&fq, 0, sizeof(fq));
  int a0, a1, b0, b1;
  if (p.d != q.d || !fast_shape(p, &fp, &a0, &a1) || !fast_shape(q, &fq, &b0, &b1) || a0 != b0 || a1 != b1) return false;
  return std::memcmp(&fp, &fq, sizeof(FastDesc)) == 0;
}

// Several blocks that share `host_desc` in as few launches as the job table allows (ASM_MAXJ per launch); descriptors outside the
// specialised kernel's shapes, per-point factors or the measurement aids: one launch_assemble per block, as before.
int launch_assemble_batch(lpgp_ctx* ctx, hipStream_t stream, const DevDesc& host_desc, const AsmJob* jobs, int njobs, double* out, int64_t ld,
                          const Layout2D& lay) {
  FastDesc fd;
  int N0 = 0, N1 = 0;
  static const int diag = [] { const char* e = std::getenv("LPGP_ASM_DIAG"); return e ? std::atoi(e) : 0; }();
  const bool fast = ctx->asm_fast && ctx->asm_batch && diag == 0 && !(ctx->asm_factors && host_desc.ngroups <= FACT_MAXG) && fast_shape(host_desc, &fd, &N0, &N1);
  if (!fast || njobs <= 1) {
    for (int j = 0; j < njobs; ++j)
      LPGP_TRY_RC(launch_assemble(ctx, stream, host_desc, jobs[j].x0, jobs[j].n0, jobs[j].n0_pad, jobs[j].x1, jobs[j].n1, jobs[j].n1_pad, out, ld,
                                  jobs[j].row_off, jobs[j].col_off, jobs[j].lower_only, lay));
    return 0;
  }
  for (int j0 = 0; j0 < njobs; j0 += ASM_MAXJ) {
    AsmBatch bt;
    double entries = 0.0;
    int wgs = 0;
    for (int j = j0; j < njobs && j < j0 + ASM_MAXJ; ++j) {
      const AsmJob& J = jobs[j];
      const int tr = (int)((J.n0 + AT - 1) / AT), tc = (int)((J.n1 + AT - 1) / AT);
      if (tr == 0 || tc == 0) continue;
      AsmJobDev& d = bt.job[bt.njobs];
      d.x0 = J.x0; d.x1 = J.x1; d.n0 = J.n0; d.n1 = J.n1; d.n0_pad = J.n0_pad; d.n1_pad = J.n1_pad;
      d.row_off = J.row_off; d.col_off = J.col_off; d.lower_only = J.lower_only;
      d.tiles_r = tr; d.tiles_c = tc; d.ct = fast_ct(ctx, tr, tc, J.lower_only);
      bt.wg0[bt.njobs] = wgs;
      wgs += tr * ((tc + d.ct - 1) / d.ct);
      ++bt.njobs;
      entries += J.lower_only ? 0.5 * (double)J.n0 * ((double)J.n0 + 1.0) : (double)J.n0 * (double)J.n1;
    }
    if (bt.njobs == 0) continue;
    bt.wg0[bt.njobs] = wgs;
    AsmArgs a;                       // the first job's fields (a launch with ONE job runs on them alone)
    const AsmJobDev& f = bt.job[0];
    a.x0 = f.x0; a.x1 = f.x1; a.n0 = f.n0; a.n1 = f.n1; a.n0_pad = f.n0_pad; a.n1_pad = f.n1_pad;
    a.out = out; a.ld = ld; a.row_off = f.row_off; a.col_off = f.col_off; a.lower_only = f.lower_only; a.lay = lay;
    a.tiles_r = f.tiles_r; a.tiles_c = f.tiles_c; a.ct = f.ct; a.flags = 0;
    prof_begin(ctx, stream, LPGP_K_ASSEMBLE, 0.0, 8.0 * entries);
    g_asm_batch = &bt;
    if (host_desc.d == 1) launch_fast<0, 1>(N0, N1, dim3((unsigned)wgs), stream, fd, a);
    else launch_fast<0, 2>(N0, N1, dim3((unsigned)wgs), stream, fd, a);
    g_asm_batch = nullptr;
    prof_end(ctx, stream);
    LPGP_HIP(hipGetLastError());
  }
  return 0;
}

// ---------------------------------------------------------------------------------------
// Matrix-free product  Y = K(X0, X1) V : every entry is evaluated on the fly and consumed in
// registers, nothing of the n0 x n1 matrix touches HBM.  This is the slot the reference fills
// with KeOps lazy tensors (`_keops_lazy_tensor`, diffops/_matern.py:112-135,231-264;
// experiments/cpu.py:214-229) for iterative solvers beyond dense-memory N.
// Bound: fp64 VALU (one exp + the Horner polynomial per entry; 2*nrhs flops of "useful" work).
// A workgroup owns 64 rows and one of `splits` column ranges; lane = row, wave = 16-column
// slab of each 64-column tile, up to MV_R right-hand sides ride along per evaluation.
// Partial sums per split go to `part`; mv_reduce_kernel adds them in a fixed order.
// ---------------------------------------------------------------------------------------
template <int D, bool FACTORS>
__global__ __launch_bounds__(256) void matvec_kernel(const DevDesc* __restrict__ desc, MvArgs a) {
  __shared__ double sx1[D][AT];
  __shared__ double sv[MV_R][AT];
  __shared__ double red[3][MV_R][AT];
  // per-point exponential factors (see assemble_kernel): origin = the first row of this workgroup's row tile; the row
  // factors are computed once, the column factors per column tile, and a tile whose points lie more than FACT_TMAX scaled
  // units from the origin falls back to one exp per entry
  __shared__ double sfr[FACTORS ? FACT_MAXG * D * 2 * AT : 1], sfc[FACTORS ? FACT_MAXG * D * 2 * AT : 1];
  __shared__ int s_fast_r, s_fast_c[2];
  __shared__ __attribute__((aligned(16))) double s_exp[2 * EXP_TAB_N];
  const int tr = blockIdx.x % a.tiles_r, sp = blockIdx.x / a.tiles_r;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wu = __builtin_amdgcn_readfirstlane(w);
  const int64_t row = (int64_t)tr * AT + lane;
  double xr[D], x0[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    xr[j] = (row < a.n0) ? a.x0[j * a.n0_pad + row] : 0.0;
    x0[j] = a.x0[j * a.n0_pad + (int64_t)tr * AT];
  }
  double y[MV_R];
#pragma unroll
  for (int r = 0; r < MV_R; ++r) y[r] = 0.0;
  const int per = (a.tiles_c + a.splits - 1) / a.splits;
  const int tc_end = (sp + 1) * per < a.tiles_c ? (sp + 1) * per : a.tiles_c;
  if (threadIdx.x == 0) { s_fast_r = FACTORS ? a.factors : 0; s_fast_c[0] = 1; s_fast_c[1] = 1; }
  s_exp[threadIdx.x] = g_exp_table[threadIdx.x], s_exp[threadIdx.x + 256] = g_exp_table[threadIdx.x + 256];
  const ExpTab etab{s_exp};
  __syncthreads();
  if (FACTORS && a.factors) stage_factors<D>(desc, xr, x0, row < a.n0, sfr, lane, wu, 4, &s_fast_r);

  for (int tc = sp * per; tc < tc_end; ++tc) {
    __syncthreads();                                   // previous tile consumed (first tile: flags initialised)
    {
      const int64_t c = (int64_t)tc * AT + lane;       // wave w stages coordinate / vector row w, w+4, ...
      for (int j = w; j < D; j += 4) sx1[j][lane] = (c < a.n1) ? a.x1[j * a.n1_pad + c] : 0.0;
      for (int r = w; r < MV_R; r += 4) sv[r][lane] = (c < a.n1 && r < a.nr) ? a.v[(int64_t)r * a.v_stride + c] : 0.0;
      double xc[D];
#pragma unroll
      for (int j = 0; j < D; ++j) xc[j] = (c < a.n1) ? a.x1[j * a.n1_pad + c] : 0.0;
      if (FACTORS && a.factors) stage_factors<D>(desc, xc, x0, c < a.n1, sfc, lane, wu, 4, &s_fast_c[tc & 1]);
      if (threadIdx.x == 0) s_fast_c[(tc + 1) & 1] = 1;       // the next tile's flag (read two barriers from now)
    }
    __syncthreads();
    const bool fast = FACTORS && s_fast_r != 0 && s_fast_c[tc & 1] != 0;
#pragma unroll 1
    for (int pass = 0; pass < 16 / AEK; ++pass) {
      const int cb = w * 16 + pass * AEK;
      double dx[D][AEK], res[AEK];
#pragma unroll
      for (int j = 0; j < D; ++j)
#pragma unroll
        for (int e = 0; e < AEK; ++e) dx[j][e] = xr[j] - sx1[j][cb + e];
      if (fast) {
        LdsFactors<D> fac{sfr + lane, sfc + cb};
        eval_entries<D, AEK, LdsFactors<D>>(desc, dx, res, etab, fac);
      } else {
        eval_entries<D, AEK>(desc, dx, res, etab);
      }
#pragma unroll
      for (int e = 0; e < AEK; ++e)
#pragma unroll
        for (int r = 0; r < MV_R; ++r) y[r] = fma(res[e], sv[r][cb + e], y[r]);   // columns >= n1 carry v = 0
    }
  }
  __syncthreads();
  if (w > 0) {
#pragma unroll
    for (int r = 0; r < MV_R; ++r) red[w - 1][r][lane] = y[r];
  }
  __syncthreads();
  if (w == 0 && row < a.n0) {
#pragma unroll
    for (int r = 0; r < MV_R; ++r)
      if (r < a.nr) a.part[((int64_t)sp * MV_R + r) * a.n0_pad + row] = ((y[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane];
  }
}

// out[r][out_stride] (accumulate: +=) the sum over the splits, in a fixed order
__global__ void mv_reduce_kernel(const double* __restrict__ part, double* __restrict__ out, int64_t n0, int64_t n0_pad,
                                 int splits, int nr, int64_t out_stride, int accumulate) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (i >= n0 || r >= nr) return;
  double s = 0.0;
  for (int sp = 0; sp < splits; ++sp) s += part[((int64_t)sp * MV_R + r) * n0_pad + i];
  double* o = out + (int64_t)r * out_stride + i;
  *o = accumulate ? *o + s : s;
}

// out[r][n0_pad] = sum_j K(x0_i, x1_j) v[r][j] for r < nr <= MV_R; `part` holds splits*MV_R*n0_pad doubles
int launch_matvec(lpgp_ctx* ctx, hipStream_t stream, const DevDesc& host_desc, const double* x0, int64_t n0,
                  int64_t n0_pad, const double* x1, int64_t n1, int64_t n1_pad, const double* v, int nr,
                  double* part, int splits, double* out, int64_t v_stride, int64_t out_stride, int accumulate) {
  if (v_stride <= 0) v_stride = n1_pad;
  if (out_stride <= 0) out_stride = n0_pad;
  MvArgs a;
  a.x0 = x0; a.x1 = x1; a.n0 = n0; a.n1 = n1; a.n0_pad = n0_pad; a.n1_pad = n1_pad;
  a.v = v; a.v_stride = v_stride; a.part = part; a.nr = nr;
  a.factors = (ctx->asm_factors && host_desc.ngroups <= FACT_MAXG) ? 1 : 0;
  a.tiles_r = (int)((n0 + AT - 1) / AT);
  a.tiles_c = (int)((n1 + AT - 1) / AT);
  a.splits = splits;
  dim3 grid((unsigned)((int64_t)a.tiles_r * splits));
  {
    FastDesc fd;
    int N0 = 0, N1 = 0;
    if (ctx->asm_fast && !a.factors && fast_shape(host_desc, &fd, &N0, &N1)) {      // the common shapes: specialised evaluation
      prof_begin(ctx, stream, LPGP_K_MATVEC, 2.0 * (double)n0 * (double)n1 * nr, 0.0);
      if (host_desc.d == 1) launch_fast<1, 1>(N0, N1, grid, stream, fd, a);
      else launch_fast<1, 2>(N0, N1, grid, stream, fd, a);
      prof_end(ctx, stream);
      LPGP_HIP(hipGetLastError());
      hipLaunchKernelGGL(mv_reduce_kernel, dim3((unsigned)((n0 + 255) / 256), (unsigned)nr), dim3(256), 0, stream,
                         (const double*)part, out, n0, n0_pad, splits, nr, out_stride, accumulate);
      LPGP_HIP(hipGetLastError());
      return 0;
    }
  }
  lpgp_ctx::DescSlot* slot = nullptr;
  int rc = stage_desc(ctx, stream, host_desc, &slot);
  if (rc != 0) return rc;
  prof_begin(ctx, stream, LPGP_K_MATVEC, 2.0 * (double)n0 * (double)n1 * nr, 0.0);
  switch (host_desc.d) {
    case 1: if (a.factors) hipLaunchKernelGGL((matvec_kernel<1, true>), grid, dim3(256), 0, stream, slot->d, a); else hipLaunchKernelGGL((matvec_kernel<1, false>), grid, dim3(256), 0, stream, slot->d, a); break;
    case 2: if (a.factors) hipLaunchKernelGGL((matvec_kernel<2, true>), grid, dim3(256), 0, stream, slot->d, a); else hipLaunchKernelGGL((matvec_kernel<2, false>), grid, dim3(256), 0, stream, slot->d, a); break;
    case 3: if (a.factors) hipLaunchKernelGGL((matvec_kernel<3, true>), grid, dim3(256), 0, stream, slot->d, a); else hipLaunchKernelGGL((matvec_kernel<3, false>), grid, dim3(256), 0, stream, slot->d, a); break;
    case 4: if (a.factors) hipLaunchKernelGGL((matvec_kernel<4, true>), grid, dim3(256), 0, stream, slot->d, a); else hipLaunchKernelGGL((matvec_kernel<4, false>), grid, dim3(256), 0, stream, slot->d, a); break;
    default: LPGP_CHECK(false, "matvec: d=%d", host_desc.d);
  }
  prof_end(ctx, stream);
  LPGP_HIP(hipGetLastError());
  LPGP_HIP(hipEventRecord(slot->done, stream));
  slot->used = true;
  hipLaunchKernelGGL(mv_reduce_kernel, dim3((unsigned)((n0 + 255) / 256), (unsigned)nr), dim3(256), 0, stream,
                     (const double*)part, out, n0, n0_pad, splits, nr, out_stride, accumulate);
  LPGP_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------
// Tensor-grid ("Kronecker") assembly.  Every kernel this library evaluates is a product over
// input dimensions, so on point sets that are tensor grids (rows = grid of factors F0[0..D),
// columns = grid of F1[0..D), C order) a block is a sum of Kronecker products of 1-D matrices:
//   G[(i_0..i_{D-1}), (j_0..j_{D-1})] = sum_t c_t prod_d M_{u(t,d)}[i_d, j_d],
//   M_u = [d^{n0} d'^{n1} k_d](F0[d], F1[d])   (each distinct (d, n0, n1) of a group once).
// This is the structure the reference exposes as `TensorProduct.linop` /
// `TensorProduct_LinDiffOp_LinDiffOp.linop` (covfuncs/_tensor_product.py:64-82,
// diffops/_tensor_product.py:140-156).  The 1-D matrices come from assemble_kernel<1>
// (O(T d n^2) kernel evaluations instead of O(N^2)); kron_expand_kernel then writes the block
// with a handful of cached loads and multiplies per entry: no exp, no polynomial.
// ---------------------------------------------------------------------------------------
constexpr int KR_MAXT = 48;      // terms over all groups
constexpr int KR_MAXU = 16;      // distinct 1-D matrices per dimension

struct KronArgs {
  int32_t nterms;
  int32_t nuniq[LPGP_MAXD];
  int32_t n0d[LPGP_MAXD], n1d[LPGP_MAXD];     // grid extents per dimension (rows / columns)
  int32_t ldu[LPGP_MAXD];                     // leading dimension of the 1-D matrices of dimension d
  const double* u[LPGP_MAXD];                 // dimension d: nuniq[d] matrices, n1d[d] * ldu[d] doubles apart
  double coef[KR_MAXT];
  uint8_t which[KR_MAXT][LPGP_MAXD];          // term t uses matrix which[t][d] of dimension d
  int64_t n0, n1;
  double* out;
  int64_t ld, row_off, col_off;
  int32_t lower_only, tiles_r, tiles_c;
  Layout2D lay;
};

// NU: compile-time bound on the number of distinct fast-dimension matrices (2, 4, 8 or 16), so
// that their weights q[] live in registers and the 8 x NU loads of a pass are issued together.
template <int D, int NU>
__global__ __launch_bounds__(256) void kron_expand_kernel(KronArgs a) {
  const int tr = blockIdx.x % a.tiles_r, tc = blockIdx.x / a.tiles_r;
  const int64_t r0 = (int64_t)tr * AT, c0 = (int64_t)tc * AT;
  if (a.lower_only && c0 > r0 + AT - 1) return;
  const int64_t lrow0 = cyc_local(a.lay.rows, a.row_off + r0), lcol0 = cyc_local(a.lay.cols, a.col_off + c0);
  if (lrow0 < 0 || lcol0 < 0) return;                 // (one owner per 64 x 64 tile, see assemble_kernel)
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t row = r0 + lane;
  // row multi-index (C order: last dimension fastest) as offsets into the 1-D matrices
  int roff[D];
  {
    unsigned rem = (unsigned)(row < a.n0 ? row : a.n0 - 1);
#pragma unroll
    for (int d = D - 1; d >= 0; --d) {
      const unsigned q = rem / (unsigned)a.n0d[d];
      roff[d] = (int)(rem - q * (unsigned)a.n0d[d]);
      rem = q;
    }
  }
  // column multi-index of the first column of this wave's 16-column slab; advanced incrementally
  // (a division per entry would cost more than the entry)
  const int64_t cfirst = c0 + w * 16;
  int j[D];
  {
    unsigned rem = (unsigned)(cfirst < a.n1 ? cfirst : 0);
#pragma unroll
    for (int d = D - 1; d >= 0; --d) {
      const unsigned q = rem / (unsigned)a.n1d[d];
      j[d] = (int)(rem - q * (unsigned)a.n1d[d]);
      rem = q;
    }
  }
  const int nfast = a.n1d[D - 1], ldf = a.ldu[D - 1];
  const int nu = a.nuniq[D - 1];
  const int64_t sfast = (int64_t)nfast * ldf;
  // fast-dimension matrices beyond nu alias the last one and get weight 0
  const double* ufast[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) ufast[u] = a.u[D - 1] + (int64_t)(u < nu ? u : nu - 1) * sfast + roff[D - 1];
  double* const outp = a.out + (lrow0 + lane) + (lcol0 - c0) * a.ld;       // column c of the block at outp[c * ld]
  const bool row_ok = row < a.n0;
  double q[NU];                  // weight of each fast-dimension matrix for the current slow column index
  bool fresh = true;
  int e = 0;
  while (e < 16 && cfirst + e < a.n1) {
    if (fresh) {
      // q[u] = sum over the terms that use fast matrix u of  coef * prod_{d < D-1} M[i_d, j_d]
#pragma unroll
      for (int u = 0; u < NU; ++u) q[u] = 0.0;
      for (int t = 0; t < a.nterms; ++t) {
        double p = a.coef[t];
#pragma unroll
        for (int d = 0; d < D - 1; ++d)
          p *= a.u[d][(int64_t)a.which[t][d] * a.n1d[d] * a.ldu[d] + (int64_t)j[d] * a.ldu[d] + roff[d]];
        const int uu = a.which[t][D - 1];
#pragma unroll
        for (int u = 0; u < NU; ++u) q[u] += (uu == u) ? p : 0.0;
      }
      fresh = false;
    }
    // columns until the fastest index wraps (at most 8 per pass)
    int seg = nfast - j[D - 1];
    if (seg > 16 - e) seg = 16 - e;
    if ((int64_t)seg > a.n1 - (cfirst + e)) seg = (int)(a.n1 - (cfirst + e));
    if (seg >= 8) {
      seg = 8;
      double acc[8];
#pragma unroll
      for (int x = 0; x < 8; ++x) acc[x] = 0.0;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const double* col = ufast[u] + (int64_t)j[D - 1] * ldf;
#pragma unroll
        for (int x = 0; x < 8; ++x) acc[x] = fma(q[u], col[x * ldf], acc[x]);
      }
      if (row_ok) {
#pragma unroll
        for (int x = 0; x < 8; ++x) outp[(cfirst + e + x) * a.ld] = acc[x];
      }
    } else {
      for (int x = 0; x < seg; ++x) {
        double acc = 0.0;
#pragma unroll
        for (int u = 0; u < NU; ++u) acc = fma(q[u], ufast[u][(int64_t)(j[D - 1] + x) * ldf], acc);
        if (row_ok) outp[(cfirst + e + x) * a.ld] = acc;
      }
    }
    e += seg;
    // advance the multi-index; a carry out of the fastest dimension changes q
    j[D - 1] += seg;
    if (j[D - 1] == nfast) {
      j[D - 1] = 0;
      fresh = true;
#pragma unroll
      for (int d = D - 2; d >= 0; --d) {
        if (++j[d] < a.n1d[d]) break;
        j[d] = 0;
      }
    }
  }
}

// Two-dimensional grids, NU <= 8: the roles are turned around.  A workgroup keeps a 64 x 32
// sub-block of every FAST-dimension matrix in registers (lane = fast row index, wave = 8 fast
// columns) and walks over KR_PAIRS pairs (i_s, j_s) of SLOW indices; per pair the weights q[u]
// come from a few wave-uniform loads of the slow-dimension matrices and every output entry costs
// NU fused multiply-adds and its store -- no vector loads in the loop, so the kernel runs at the
// rate of the stores (measured with scratch/store_probe.hip: this store pattern alone reaches
// 5.9 TB/s, the per-entry evaluation 2.9 TB/s).
constexpr int KR_PAIRS = 16;
// DIST: multi-GPU layout (ownership and local index per row / column); the single-GPU instantiation keeps the
// store addressing free of the per-pair index mapping (measured: 4.07 TB/s against 3.70 with the mapping in)
template <int NU, bool DIST>
__global__ __launch_bounds__(256) void kron2_kernel(KronArgs a, int ftr, int ftc) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  int b = blockIdx.x;
  const int tr = b % ftr; b /= ftr;
  const int tcf = b % ftc;
  const int chunk = b / ftc;
  const int n0s = a.n0d[0], n1s = a.n1d[0], n0f = a.n0d[1], n1f = a.n1d[1];
  const int ifast = tr * 64 + lane;                    // fast row index of this lane
  const int jf0 = tcf * 32 + w * 8;                    // first fast column of this wave
  const int nu = a.nuniq[1];
  const int64_t sfast = (int64_t)n1f * a.ldu[1];
  const bool row_ok = ifast < n0f;
  double vals[NU][8];
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const int jf = jf0 + x;
      vals[u][x] = (u < nu && row_ok && jf < n1f) ? a.u[1][(int64_t)u * sfast + (int64_t)jf * a.ldu[1] + ifast] : 0.0;
    }
  const int64_t sslow = (int64_t)n1s * a.ldu[0];
  const int64_t npairs = (int64_t)n0s * n1s;
  int64_t p = (int64_t)chunk * KR_PAIRS;
  int is = (int)(p % n0s), js = (int)(p / n0s);
  for (int it = 0; it < KR_PAIRS && p < npairs; ++it, ++p) {
    const int64_t rtile = (int64_t)is * n0f + tr * 64;            // first global row / column of the
    const int64_t ctile = (int64_t)js * n1f + tcf * 32;           // 64 x 32 tile of this pair
    const bool skip = a.lower_only && ctile > rtile + 63;
    // multi-GPU: a tile of this kernel is not aligned to the 128-tiles of the padded matrix, so ownership is
    // decided per row (lane) and per column
    int64_t lrow = 0, lcol[8];
    unsigned mine = skip ? 0u : 0xffu;
    if (DIST && !skip) {
      lrow = cyc_local(a.lay.rows, a.row_off + (int64_t)is * n0f + ifast);
      mine = 0;
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        lcol[x] = cyc_local(a.lay.cols, a.col_off + (int64_t)js * n1f + jf0 + x);
        if (lcol[x] >= 0) mine |= 1u << x;
      }
    }
    if (!skip && mine != 0) {
      double q[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) q[u] = 0.0;
      for (int t = 0; t < a.nterms; ++t) {
        const double pv = a.coef[t] * a.u[0][(int64_t)a.which[t][0] * sslow + (int64_t)js * a.ldu[0] + is];
        const int uu = a.which[t][1];
#pragma unroll
        for (int u = 0; u < NU; ++u) q[u] += (uu == u) ? pv : 0.0;
      }
      double* outp = a.out + (a.row_off + (int64_t)is * n0f + ifast) + (a.col_off + (int64_t)js * n1f + jf0) * a.ld;
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        double acc = 0.0;
#pragma unroll
        for (int u = 0; u < NU; ++u) acc = fma(q[u], vals[u][x], acc);
        if (DIST) {
          if (row_ok && lrow >= 0 && jf0 + x < n1f && ((mine >> x) & 1u)) a.out[lrow + lcol[x] * a.ld] = acc;
        } else {
          if (row_ok && jf0 + x < n1f) outp[(int64_t)x * a.ld] = acc;
        }
      }
    }
    if (++is == n0s) {
      is = 0;
      ++js;
    }
  }
}

// The same tile walk with 16-BYTE stores (round 5): a lane owns TWO consecutive fast rows, a wave stores 128 rows x 1 column = 1 KB per
// instruction, a workgroup a 128 x 32 tile per pair of slow indices -- half the store instructions and address arithmetic per byte
// of kron2_kernel.  Single GPU, even fast extent (the rows of a pair then start on a 16-byte boundary), at most 4 distinct fast
// matrices (two rows of each in registers: 64 doubles).
template <int NU>
__global__ __launch_bounds__(256) void kron2w_kernel(KronArgs a, int ftr, int ftc) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  int b = blockIdx.x;
  const int tr = b % ftr; b /= ftr;
  const int tcf = b % ftc;
  const int chunk = b / ftc;
  const int n0s = a.n0d[0], n1s = a.n1d[0], n0f = a.n0d[1], n1f = a.n1d[1];
  const int if0 = tr * 128 + 2 * lane;                 // first of this lane's two fast rows (n0f even: both valid or neither)
  const int jf0 = tcf * 32 + w * 8;
  if (a.lower_only) {
    // a diagonal block: half of the pairs of slow indices lie above the diagonal -- a workgroup whose pairs ALL do leaves before
    // it loads its 64 values per lane (scalar test over <= 16 pairs)
    const int64_t np_ = (int64_t)n0s * n1s;
    bool any = false;
    for (int64_t pp = (int64_t)chunk * KR_PAIRS; pp < np_ && pp < (int64_t)(chunk + 1) * KR_PAIRS; ++pp) {
      const int64_t is_ = pp % n0s, js_ = pp / n0s;
      any = any || !(js_ * n1f + tcf * 32 > is_ * n0f + tr * 128 + 127);
    }
    if (!any) return;
  }
  const int nu = a.nuniq[1];
  const int64_t sfast = (int64_t)n1f * a.ldu[1];
  const bool row_ok = if0 < n0f;
  double v0[NU][8], v1[NU][8];
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const int jf = jf0 + x;
      const bool ok = u < nu && row_ok && jf < n1f;
      const double* src = a.u[1] + (int64_t)u * sfast + (int64_t)jf * a.ldu[1] + if0;
      v0[u][x] = ok ? src[0] : 0.0;
      v1[u][x] = ok ? src[1] : 0.0;
    }
  const int64_t sslow = (int64_t)n1s * a.ldu[0];
  const int64_t npairs = (int64_t)n0s * n1s;
  int64_t p = (int64_t)chunk * KR_PAIRS;
  int is = (int)(p % n0s), js = (int)(p / n0s);
  for (int it = 0; it < KR_PAIRS && p < npairs; ++it, ++p) {
    const int64_t rtile = (int64_t)is * n0f + tr * 128;
    const int64_t ctile = (int64_t)js * n1f + tcf * 32;
    const bool skip = a.lower_only && ctile > rtile + 127;
    if (!skip) {
      double q[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) q[u] = 0.0;
      for (int t = 0; t < a.nterms; ++t) {
        const double pv = a.coef[t] * a.u[0][(int64_t)a.which[t][0] * sslow + (int64_t)js * a.ldu[0] + is];
        const int uu = a.which[t][1];
#pragma unroll
        for (int u = 0; u < NU; ++u) q[u] += (uu == u) ? pv : 0.0;
      }
      double* outp = a.out + (a.row_off + (int64_t)is * n0f + if0) + (a.col_off + (int64_t)js * n1f + jf0) * a.ld;
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          a0 = fma(q[u], v0[u][x], a0);
          a1 = fma(q[u], v1[u][x], a1);
        }
        if (row_ok && jf0 + x < n1f) *reinterpret_cast<double2*>(outp + (int64_t)x * a.ld) = make_double2(a0, a1);
      }
    }
    if (++is == n0s) {
      is = 0;
      ++js;
    }
  }
}

template <int D>
static void launch_kron_nu(dim3 grid, hipStream_t stream, const KronArgs& a) {
  const int nu = a.nuniq[D - 1];
  if (nu <= 2) hipLaunchKernelGGL((kron_expand_kernel<D, 2>), grid, dim3(256), 0, stream, a);
  else if (nu <= 4) hipLaunchKernelGGL((kron_expand_kernel<D, 4>), grid, dim3(256), 0, stream, a);
  else if (nu <= 8) hipLaunchKernelGGL((kron_expand_kernel<D, 8>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((kron_expand_kernel<D, KR_MAXU>), grid, dim3(256), 0, stream, a);
}

// Does the Kronecker path hold this sum (its term table and its per-dimension tables of distinct 1-D matrices are fixed-size
// kernel arguments)?  Callers fall back to the entry-wise assembly of the flattened grids otherwise (lpgp_kron_fits).
bool kron_fits(const lpgp_kdesc* kd, int ngroups) {
  if (!kd || ngroups < 1 || ngroups > LPGP_MAXG) return false;
  const int D = kd[0].d;
  if (D < 1 || D > LPGP_MAXD) return false;
  struct Key { int g, n0, n1; };
  std::vector<Key> keys[LPGP_MAXD];
  int nterms = 0;
  for (int g = 0; g < ngroups; ++g) {
    const lpgp_kdesc& K = kd[g];
    if (K.d != D || K.family[0] == LPGP_MATERN_ISO || K.nterms < 0 || K.nterms > LPGP_MAXT) return false;
    nterms += K.nterms;
    if (nterms > KR_MAXT) return false;
    for (int t = 0; t < K.nterms; ++t)
      for (int d = 0; d < D; ++d) {
        const int n0 = K.terms[t].n0[d], n1 = K.terms[t].n1[d];
        bool found = false;
        for (const Key& q : keys[d]) found = found || (q.g == g && q.n0 == n0 && q.n1 == n1);
        if (!found) {
          if ((int)keys[d].size() >= KR_MAXU) return false;
          keys[d].push_back({g, n0, n1});
        }
      }
  }
  return true;
}

// F0[d] / F1[d]: device coordinate arrays of the grid factors (n0d[d] / n1d[d] points); work:
// device scratch of at least kron_work_doubles(...) doubles.
int launch_assemble_kron(lpgp_ctx* ctx, hipStream_t stream, const lpgp_kdesc* kd, int ngroups,
                         const double* const* F0, const int64_t* n0d, const double* const* F1, const int64_t* n1d,
                         double* work, size_t work_doubles, double* out, int64_t ld, int64_t row_off,
                         int64_t col_off, int lower_only, const Layout2D& lay) {
  const int D = kd[0].d;
  KronArgs a;
  std::memset(&a, 0, sizeof(a));
  a.n0 = 1; a.n1 = 1;
  size_t woff = 0;
  size_t dim_off[LPGP_MAXD];
  for (int d = 0; d < D; ++d) {
    a.n0d[d] = (int32_t)n0d[d];
    a.n1d[d] = (int32_t)n1d[d];
    a.ldu[d] = (int32_t)round_up(n0d[d], 2);
    a.n0 *= n0d[d];
    a.n1 *= n1d[d];
    dim_off[d] = woff;
    woff += (size_t)KR_MAXU * a.ldu[d] * n1d[d];
  }
  LPGP_CHECK(woff <= work_doubles, "kron assembly: scratch too small");
  LPGP_CHECK(a.n0 < (int64_t)1 << 31 && a.n1 < (int64_t)1 << 31, "kron assembly: grid too large");
  // distinct 1-D matrices: key (group, n0, n1) per dimension
  struct Key { int g, n0, n1; };
  std::vector<Key> keys[LPGP_MAXD];
  for (int g = 0; g < ngroups; ++g) {
    const lpgp_kdesc& K = kd[g];
    LPGP_CHECK(K.d == D, "kron assembly: group %d has d=%d != %d", g, K.d, D);
    LPGP_CHECK(K.family[0] != LPGP_MATERN_ISO, "kron assembly: an isotropic Matern is not a product over dimensions");
    for (int t = 0; t < K.nterms; ++t) {
      LPGP_CHECK(a.nterms < KR_MAXT, "kron assembly: more than %d terms (lpgp_kron_fits tells beforehand)", KR_MAXT);
      const int ti = a.nterms++;
      a.coef[ti] = K.scale * K.terms[t].coef;
      for (int d = 0; d < D; ++d) {
        const int n0 = K.terms[t].n0[d], n1 = K.terms[t].n1[d];
        int found = -1;
        for (size_t q = 0; q < keys[d].size(); ++q)
          if (keys[d][q].g == g && keys[d][q].n0 == n0 && keys[d][q].n1 == n1) found = (int)q;
        if (found < 0) {
          LPGP_CHECK((int)keys[d].size() < KR_MAXU, "kron assembly: more than %d distinct 1-D factors", KR_MAXU);
          keys[d].push_back({g, n0, n1});
          found = (int)keys[d].size() - 1;
        }
        a.which[ti][d] = (uint8_t)found;
      }
    }
  }
  // evaluate the 1-D matrices with the ordinary assembly kernel (D = 1, scale 1, one term)
  const Layout2D none;
  for (int d = 0; d < D; ++d) {
    a.nuniq[d] = (int32_t)keys[d].size();
    a.u[d] = work + dim_off[d];
    for (size_t q = 0; q < keys[d].size(); ++q) {
      const lpgp_kdesc& K = kd[keys[d][q].g];
      lpgp_kdesc k1;
      std::memset(&k1, 0, sizeof(k1));
      k1.d = 1;
      k1.family[0] = K.family[d];
      k1.p[0] = K.p[d];
      k1.lengthscale[0] = K.lengthscale[d];
      k1.scale = 1.0;
      k1.nterms = 1;
      k1.terms[0].coef = 1.0;
      k1.terms[0].n0[0] = keys[d][q].n0;
      k1.terms[0].n1[0] = keys[d][q].n1;
      DevDesc dd;
      int rc = lower_kdesc(&k1, 1, &dd);
      if (rc != 0) return rc;
      double* dst = work + dim_off[d] + q * (size_t)a.ldu[d] * n1d[d];
      rc = launch_assemble(ctx, stream, dd, F0[d], n0d[d], n0d[d], F1[d], n1d[d], n1d[d], dst, a.ldu[d], 0, 0, 0, none);
      if (rc != 0) return rc;
    }
  }
  a.out = out; a.ld = ld; a.row_off = row_off; a.col_off = col_off; a.lower_only = lower_only;
  a.lay = lay;
  a.tiles_r = (int)((a.n0 + AT - 1) / AT);
  a.tiles_c = (int)((a.n1 + AT - 1) / AT);
  if (a.tiles_r == 0 || a.tiles_c == 0) return 0;
  dim3 grid((unsigned)((int64_t)a.tiles_r * a.tiles_c));
  const double entries = lower_only ? 0.5 * (double)a.n0 * ((double)a.n0 + 1.0) : (double)a.n0 * (double)a.n1;
  prof_begin(ctx, stream, LPGP_K_ASSEMBLE_GRID, 0.0, 8.0 * entries);
  if (D == 2 && a.nuniq[1] <= 8 && a.n0d[1] >= 32 && a.n1d[1] >= 16) {
    const bool dist = lay.rows.P > 1 || lay.cols.P > 1;
    const int ftr = (a.n0d[1] + 63) / 64, ftc = (a.n1d[1] + 31) / 32;
    const int64_t chunks = ((int64_t)a.n0d[0] * a.n1d[0] + KR_PAIRS - 1) / KR_PAIRS;
    dim3 g2((unsigned)((int64_t)ftr * ftc * chunks));
    if (!dist && ctx->kron_wide && a.n0d[1] % 2 == 0 && a.nuniq[1] <= 4 && ((a.row_off | a.ld) & 1) == 0) {
      // 16-byte stores: a lane owns two consecutive fast rows (kron2w_kernel)
      const int ftrw = (a.n0d[1] + 127) / 128;
      hipLaunchKernelGGL((kron2w_kernel<4>), dim3((unsigned)((int64_t)ftrw * ftc * chunks)), dim3(256), 0, stream, a, ftrw, ftc);
    } else if (a.nuniq[1] <= 4) {
      if (dist) hipLaunchKernelGGL((kron2_kernel<4, true>), g2, dim3(256), 0, stream, a, ftr, ftc);
      else hipLaunchKernelGGL((kron2_kernel<4, false>), g2, dim3(256), 0, stream, a, ftr, ftc);
    } else {
      if (dist) hipLaunchKernelGGL((kron2_kernel<8, true>), g2, dim3(256), 0, stream, a, ftr, ftc);
      else hipLaunchKernelGGL((kron2_kernel<8, false>), g2, dim3(256), 0, stream, a, ftr, ftc);
    }
  } else
  switch (D) {
    case 1: launch_kron_nu<1>(grid, stream, a); break;
    case 2: launch_kron_nu<2>(grid, stream, a); break;
    case 3: launch_kron_nu<3>(grid, stream, a); break;
    case 4: launch_kron_nu<4>(grid, stream, a); break;
    default: LPGP_CHECK(false, "kron assembly: d=%d", D);
  }
  prof_end(ctx, stream);
  LPGP_HIP(hipGetLastError());
  return 0;
}

size_t kron_work_doubles(int D, const int64_t* n0d, const int64_t* n1d) {
  size_t w = 0;
  for (int d = 0; d < D; ++d) w += (size_t)KR_MAXU * (size_t)round_up(n0d[d], 2) * (size_t)n1d[d];
  return w;
}

int launch_add_diag(hipStream_t stream, double* a, int64_t ld, int64_t off, int64_t n, const double* v, double scalar,
                    const Layout2D& lay) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(add_diag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, ld, off, n, v, scalar, lay);
  LPGP_HIP(hipGetLastError());
  return 0;
}

int launch_add_dense(hipStream_t stream, double* a, int64_t ld, int64_t off, int64_t n, const double* b, const Layout2D& lay) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(add_dense_lower_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)n), dim3(256), 0, stream, a, ld, off, n, b, lay);
  LPGP_HIP(hipGetLastError());
  return 0;
}

}  // namespace lpgp
