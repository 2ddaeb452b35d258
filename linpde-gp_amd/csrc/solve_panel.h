// The fused panel step of the forward substitution (kernel template + launcher), instantiated in solve.hip (NT = 1, 2, 3),
// solve4.hip (NT = 4) and solve4p.hip (the variants with the fused look-ahead update, NP = 4): the fully unrolled stage
// schedule of NT = 4 alone takes over a minute to compile.
#pragma once

#include "lpgp_internal.h"
#include "kernel_util.h"

namespace lpgp {

constexpr int TSV_NSTAGE = 24;               // 3 products x 8 stages of 16 k
constexpr int TSV_XA = 2 * 32 * 64;          // doubles: two row groups x 32 fragments x 64 lanes
constexpr int TSV_ROWS = 32;                 // rows (KFAST: right-hand-side columns) per workgroup
constexpr int TSV_RING = 6656;               // doubles: byte-granular ring of factor stages (52 KB)

template <int N>
__device__ __forceinline__ void lds_wait_n() {
  static_assert(N >= 0 && N <= 15, "lgkmcnt field");
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
template <int N>
__device__ __forceinline__ void vm_wait_n() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Workgroup barrier WITHOUT the fence of __syncthreads(): hipcc turns that fence into s_waitcnt vmcnt(0) whenever
// LDS-DMA is in flight (a DMA writes LDS), which would serialise the stage prefetch.  What a barrier of this
// kernel needs is waited for explicitly: the wave's own LDS reads / writes (lgkmcnt) here, its DMA pieces of the
// stage about to be read by the counted vmcnt wait in front of it.
#define TSV_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// =========================================================================================
// Forward substitution of a WHOLE PANEL of the factor in ONE launch:  V[panel rows, cols] <- L_KK^{-1} V, with
// L_KK the NT x NT tile diagonal block (NT <= 4: a 512-panel).  The columns of a right-hand side are independent,
// so a workgroup that owns 32 of them can run the entire panel chain for them without ever meeting another
// workgroup: NT refined tile solves (the three products of tile_solve_kernel each) with the rank-128 updates of the
// panel rows below in between -- round 1 / the per-tile path launch {tile solve, update} NT times, 2 NT dependent
// launches of ~16 + 12 us.  Same machinery as tile_solve_kernel<true> (X = V^T: element (i = column, c = panel row) at
// V[c + i ldv]; fragments, xa image, LDS-DMA ring with a compile-time schedule); the chain is a static sequence of
// products, 8 stages each:
//     for j < NT:   x  = A_j Linv_j^T        (kind 0, triangular: stage kt feeds fragments u >= 4 kt)
//                   A_j -= x L_jj^T          (kind 1, triangular)      -> residual
//                   x += A_j Linv_j^T        (kind 2, triangular)      -> X_j
//                   A_i -= X_j L_ij^T, i > j (kind 3, full tile)
// with the fragments of all NT tiles of the panel in registers (8 NT doubles per lane).
// =========================================================================================
// Round 4, NP > 0: the kernel ALSO applies the PREVIOUS panel (NP tiles = 128 NP solved rows of V right above this panel) to
// its own rows first -- the look-ahead update (a) of the blocked forward substitution, until round 3 a launch of its own in
// front of this one -- as NP x NT more products at the head of the chain:
//     for j < NP:   for i < NT:   A_i -= Vprev_j Lprev_ij^T     (kind 4, full tile; Lprev = the factor's block left of the
//                                                                 panel's diagonal block)
// One launch instead of two, and above all ONE wait: the fused kernel becomes runnable at the very moment the previous
// remainder update ends -- together with the next remainder update -- and takes its slots on the drained chip (measured,
// kernel trace: a panel solve launched beside a starting update completes in 120 us; launched 60 us later, behind (a), it
// starves until that update drains and the update stream then idles 57-95 us per panel, 2 ms of the c3 prediction).
constexpr int psv_nprod(int NT, int NP = 0) { return NP * NT + 3 * NT + NT * (NT - 1) / 2; }
struct PsvProd { int kind, j, i; };
constexpr PsvProd psv_prod(int NT, int p, int NP = 0) {
  if (p < NP * NT) return {4, p / NT, p % NT};
  p -= NP * NT;
  int q = 0;
  for (int j = 0; j < NT; ++j) {
    for (int k = 0; k < 3; ++k, ++q)
      if (q == p) return {k, j, j};
    for (int i = j + 1; i < NT; ++i, ++q)
      if (q == p) return {3, j, i};
  }
  return {-1, 0, 0};
}
constexpr int psv_first_prod(int NT, int j, int NP = 0) {          // index of product (kind 0, j)
  int q = NP * NT;
  for (int jj = 0; jj < j; ++jj) q += 3 + (NT - 1 - jj);
  return q;
}
template <int NT, int NP = 0> constexpr bool psv_tri(int s) { return psv_prod(NT, s / 8, NP).kind < 3; }
template <int NT, int NP = 0> constexpr int psv_stride(int s) { return psv_tri<NT, NP>(s) ? 136 - 16 * (s % 8) : 136; }
template <int NT, int NP = 0> constexpr int psv_size(int s) { return 16 * psv_stride<NT, NP>(s); }
template <int NT, int NP = 0>
struct PsvSched {
  static constexpr int NS = 8 * psv_nprod(NT, NP);
  int off[NS] = {};
  int iss_lo[NS + 1] = {};
  int iss_hi[NS + 1] = {};
  int wait[NS] = {};
};
template <int NT, int NP = 0>
constexpr PsvSched<NT, NP> psv_make_sched(int dma_per_stage) {          // same placement rule as tsv_make_sched
  PsvSched<NT, NP> S;
  constexpr int NS = PsvSched<NT, NP>::NS;
  int next = 0, head = 0;
  for (int t = 0; t <= NS; ++t) {
    const int live_lo = t == 0 ? 0 : t - 1;
    S.iss_lo[t] = next;
    while (next < NS) {
      const int sz = psv_size<NT, NP>(next);
      int o = head;
      if (o + sz > TSV_RING) o = 0;
      bool ok = true;
      for (int l = live_lo; l < next; ++l)
        if (o < S.off[l] + psv_size<NT, NP>(l) && S.off[l] < o + sz) ok = false;
      if (!ok) break;
      S.off[next] = o;
      head = o + sz;
      ++next;
    }
    S.iss_hi[t] = next;
    if (t < NS) S.wait[t] = dma_per_stage * (next - (t + 1));
  }
  return S;
}
template <int NT, int RG, int NP = 0>
constexpr bool psv_sched_ok() {
  constexpr PsvSched<NT, NP> S = psv_make_sched<NT, NP>(4 / RG);
  if (S.iss_hi[PsvSched<NT, NP>::NS] != PsvSched<NT, NP>::NS) return false;
  for (int s = 0; s < PsvSched<NT, NP>::NS; ++s)
    if (S.iss_hi[s] < s + 1 || S.wait[s] < 0 || S.wait[s] > 62) return false;
  return true;
}

struct PanelSolveArgs {
  double* V;                 // top row of the panel: NT * 128 rows x all columns, column-major, leading dimension ldv
  int64_t ldv;
  const double* linv;        // tile inverses of the panel's NT diagonal tiles, contiguous (128 x 128 each, ld 128)
  const double* L;           // the panel's diagonal block of the factor (NT x NT tiles), leading dimension ldl
  int64_t ldl;
  const double* Lprev = nullptr;   // NP > 0: the factor's NT x NP tile block left of L (same leading dimension); the previous
                                   // panel's solved rows are the NP * 128 rows of V right above V
};

// RG = 16-column groups per workgroup (2: eight waves, 32 columns; 1: four waves, 16 columns -- half the matrix work per
// workgroup on twice as many, for right-hand-side blocks that would not fill the chip otherwise)
// KFAST = true: the forward substitution (above).  KFAST = false: the same chain for ROWS below an already factored
// diagonal block, X[rows, panel columns] <- X L_KK^{-T} with X(i, c) at V[i + c ldv] -- the panel solve of the multi-GPU
// factorisation (the diagonal block arrives first there) and of a block append (old panels, new rows).
template <int NT, int RG, bool KFAST, int NP = 0>
__global__ __launch_bounds__(256 * RG, 1) void panel_solve_kernel(PanelSolveArgs g) {
  static_assert(psv_sched_ok<NT, RG, NP>(), "panel solve: broken stage schedule");
  static_assert(NP == 0 || KFAST, "panel solve: the fused look-ahead update exists for the forward substitution only");
  constexpr PsvSched<NT, NP> SCH = psv_make_sched<NT, NP>(4 / RG);
  constexpr int XA = RG * 32 * 64;                       // doubles of the fragment image
  constexpr int NS = PsvSched<NT, NP>::NS;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* xa = smem;
  double* ring = smem + XA;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wu = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wu >> 2, s4 = wu & 3;
  const int cc = rg ? 3 - s4 : s4;
  const int li = lane & 15, lj = lane >> 4;
  const int64_t i0 = (int64_t)blockIdx.x * (16 * RG) + rg * 16 + li;      // this lane's right-hand-side column (KFAST) / row
  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) double*)smem);

  auto issue = [&](auto S_) {
    constexpr int s = decltype(S_)::value;
    constexpr PsvProd pd = psv_prod(NT, s / 8, NP);
    constexpr int kt = s % 8, stride = psv_stride<NT, NP>(s);
    constexpr bool tri = pd.kind < 3;
    constexpr int len = tri ? 128 - 16 * kt : 128, col0 = tri ? 16 * kt : 0;
    // element (k, c) of M^T is M[c + k ldm].  The base pointers and the leading dimension pass through an empty asm per
    // stage: the unrolled chain is straight-line code, and the compiler otherwise keeps the scalar base of EVERY factor tile
    // (up to 26 pointer pairs) alive from the first stage on -- 37 (NP = 0) to 78 (NP = 4) SGPRs spilled to VGPR lanes;
    // recomputing a base costs three scalar instructions on an idle scalar unit
    const double* lbase = (pd.kind == 0 || pd.kind == 2) ? g.linv : (pd.kind == 4 ? g.Lprev : g.L);
    int64_t ldl = g.ldl;
    asm volatile("" : "+s"(lbase), "+s"(ldl));
    const double* M = (pd.kind == 0 || pd.kind == 2) ? lbase + (int64_t)pd.j * TILE * TILE
                                                      : lbase + (int64_t)pd.i * TILE + (int64_t)pd.j * TILE * ldl;
    const int64_t ldm = (pd.kind == 0 || pd.kind == 2) ? (int64_t)TILE : ldl;
    double* sb = ring + SCH.off[s];
    if (2 * lane < len) {
#pragma unroll
      for (int h = 0; h < 4 / RG; ++h) {
        const int r = (4 / RG) * wu + h;
        const char* ub = reinterpret_cast<const char*>(M + col0 + ((int64_t)kt * 16 + r) * ldm);
        __builtin_amdgcn_global_load_lds((gptr_t)(ub + (unsigned)lane * 16u), (lptr_t)(sb + r * stride), 16, 0, 0);
      }
    }
  };

  // fragments of the NT tiles of the panel: fragment (t, q) = element (column i0, panel row 128 t + 4 (cc + 4 q) + lj)
  double a[NT][8], x[8];
  double* const pbase = KFAST ? g.V + i0 * g.ldv + (4 * cc + lj) : g.V + i0 + (int64_t)(4 * cc + lj) * g.ldv;
  const int64_t cstep = KFAST ? 1 : g.ldv;                  // address step per column c of X
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 8; ++q) a[t][q] = pbase[(t * TILE + q * 16) * cstep];
  // (NP > 0) the first operand image is the previous panel's first solved tile, negated: the chain opens with A_i -= Vprev_j Lprev_ij^T
  double vp[8];
  if constexpr (NP > 0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) vp[q] = pbase[(-(NP * TILE) + q * 16) * cstep];
  }
  asm volatile("" ::: "memory");
  issue(std::integral_constant<int, 0>{});
  double* const xown = xa + (size_t)(rg * 32 + cc) * 64 + lane;
#pragma unroll
  for (int q = 0; q < 8; ++q) xown[q * 256] = NP > 0 ? -vp[q] : a[0][q];
  asm volatile("" ::: "memory");
  static_for<1, SCH.iss_hi[0]>(issue);
  const unsigned mlane = lds_base + 8u * (unsigned)(rg * 2048 + lane);
  const unsigned nlane = lds_base + 8u * (unsigned)(XA + lj * 136 + (lane & 3) + 4 * cc);
  const unsigned lj128 = (unsigned)lj * 128u;

  // dst[q] += sum over the k-steps of (operand fragment from xa) x (factor fragment), product P of the chain
  auto run_product = [&](auto P_, double(&dst)[8]) {
    constexpr int prod = decltype(P_)::value;
    constexpr bool tri = psv_prod(NT, prod, NP).kind < 3;
    static_for<0, 8>([&](auto KT_) {
      constexpr int kt = decltype(KT_)::value;
      constexpr int s = prod * 8 + kt;
      constexpr int stride = psv_stride<NT, NP>(s);
      constexpr int q0 = tri ? kt : 0;                       // first fragment this stage feeds
      vm_wait_n<SCH.wait[s]>();
      TSV_BARRIER();
      static_for<SCH.iss_lo[s + 1], SCH.iss_hi[s + 1]>(issue);
      const unsigned aN = nlane + (unsigned)SCH.off[s] * 8u - (tri ? (unsigned)kt * lj128 : 0u);
      double mf[2], nf[2][8];
      asm volatile("" ::: "memory");
      mf[0] = lds_read_async<(4 * kt) * 64>(mlane);
      static_for<q0, 8>([&](auto Q_) {
        constexpr int q = decltype(Q_)::value;
        nf[0][q] = lds_read_async<16 * (q - q0)>(aN);
      });
      static_for<0, 4>([&](auto K_) {
        constexpr int ks = decltype(K_)::value;
        if constexpr (ks + 1 < 4) {
          mf[(ks + 1) & 1] = lds_read_async<(4 * kt + ks + 1) * 64>(mlane);
          static_for<q0, 8>([&](auto Q_) {
            constexpr int q = decltype(Q_)::value;
            nf[(ks + 1) & 1][q] = lds_read_async<(ks + 1) * 4 * stride + 16 * (q - q0)>(aN);
          });
          lds_wait_n<9 - q0>();
        } else {
          lds_wait_n<0>();
        }
        static_for<q0, 8>([&](auto Q_) {
          constexpr int q = decltype(Q_)::value;
          dst[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(nf[ks & 1][q], mf[ks & 1], dst[q], 0, 0, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  };

  // ---- (NP > 0) the fused look-ahead update: the previous panel applied to this panel's rows ----
  static_for<0, NP>([&](auto JP_) {
    constexpr int jp = decltype(JP_)::value;
    if constexpr (jp > 0) {
      TSV_BARRIER();                                           // the products of tile jp - 1 have read the fragment image
#pragma unroll
      for (int q = 0; q < 8; ++q) xown[q * 256] = -vp[q];
    }
    if constexpr (jp + 1 < NP) {
      // the next solved tile of the previous panel, in flight while the products of this one run
#pragma unroll
      for (int q = 0; q < 8; ++q) vp[q] = pbase[(-(NP * TILE) + (jp + 1) * TILE + q * 16) * cstep];
    }
    static_for<0, NT>([&](auto I_) {
      constexpr int i = decltype(I_)::value;
      run_product(std::integral_constant<int, jp * NT + i>{}, a[i]);          // A_i -= Vprev_jp Lprev_{i,jp}^T
    });
  });
  static_for<0, NT>([&](auto J_) {
    constexpr int j = decltype(J_)::value;
    constexpr int p0 = psv_first_prod(NT, j, NP);
    if constexpr (j > 0 || NP > 0) {
      TSV_BARRIER();                                           // the updates by X_{j-1} have read xa
#pragma unroll
      for (int q = 0; q < 8; ++q) xown[q * 256] = a[j][q];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = 0.0;
    run_product(std::integral_constant<int, p0>{}, x);         // x = X0 = A_j Linv_j^T
    TSV_BARRIER();
#pragma unroll
    for (int q = 0; q < 8; ++q) xown[q * 256] = -x[q];
    run_product(std::integral_constant<int, p0 + 1>{}, a[j]);  // a_j = R = A_j - X0 L_jj^T
    TSV_BARRIER();
#pragma unroll
    for (int q = 0; q < 8; ++q) xown[q * 256] = a[j][q];
    run_product(std::integral_constant<int, p0 + 2>{}, x);     // x = X_j
#pragma unroll
    for (int q = 0; q < 8; ++q) a[j][q] = x[q];
    if constexpr (j + 1 < NT) {
      TSV_BARRIER();
#pragma unroll
      for (int q = 0; q < 8; ++q) xown[q * 256] = -x[q];
      static_for<j + 1, NT>([&](auto I_) {
        constexpr int i = decltype(I_)::value;
        run_product(std::integral_constant<int, p0 + 3 + (i - j - 1)>{}, a[i]);   // A_i -= X_j L_ij^T
      });
    }
  });
  // in place: this workgroup read exactly the 32 columns of the panel it overwrites, all of them before the first barrier
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 8; ++q) pbase[(t * TILE + q * 16) * cstep] = a[t][q];
  (void)NS;
}

template <int NT, int RG, bool KFAST, int NP = 0>
int launch_panel_solve_rg(lpgp_ctx* ctx, hipStream_t stream, const PanelSolveArgs& a, int64_t cols) {
  // ctx->panel_lds_extra: LDS the launch asks for beyond what the kernel uses (69 632 B).  With it (20 480 B: 90 112 B in all) a
  // workgroup no longer fits into the room ONE retiring gemm3 workgroup leaves on a CU (36.9 KB + the 49 KB free beside three of
  // them), so beside a long three-resident update the chain waits for that update's tail instead of running inside it -- see
  // trsm_lower_two_level (round 4: without the 78 spilled SGPRs the kernel takes 152 registers, fits such a hole, and c4's
  // prediction got 3 % SLOWER: the chain ran beside the update at a fraction of its speed and took the update's slots with it).
  const size_t shmem = (size_t)(RG * 32 * 64 + TSV_RING) * sizeof(double);
  LPGP_TRY_RC(ensure_lds_attr(ctx, reinterpret_cast<const void*>(&panel_solve_kernel<NT, RG, KFAST, NP>), shmem + 20480));
  hipLaunchKernelGGL((panel_solve_kernel<NT, RG, KFAST, NP>), dim3((unsigned)(cols / (16 * RG))), dim3(256 * RG), shmem + (size_t)ctx->panel_lds_extra,
                     stream, a);
  LPGP_HIP(hipGetLastError());
  return 0;
}
// 16-column workgroups (RG = 1) throughout: per column the 32-column variant is no faster (measured: 264 workgroups of
// either kind take 117 us at NT = 4 for 4224 resp. 205 us for 8448 columns) and the chain of a narrow block is half as long
template <int NT, bool KFAST, int NP = 0>
int launch_panel_solve_nt(lpgp_ctx* ctx, hipStream_t stream, const PanelSolveArgs& a, int64_t cols) {
  return launch_panel_solve_rg<NT, 1, KFAST, NP>(ctx, stream, a, cols);
}

}  // namespace lpgp
