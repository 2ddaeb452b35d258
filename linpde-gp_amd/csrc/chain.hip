// The RESIDENT PANEL CHAIN (round 5): the whole chain of one panel of the blocked Cholesky -- four tile Choleskys, the refined
// solves of every row below against each of them, the rank-128 updates of the rest of the panel in between -- in ONE
// launch whose workgroups hand over through device flags, instead of twelve dependent launches (potrf_tile_kernel ->
// tile_solve_kernel -> in-panel update, four times) with 6-15 us between every two of them.
//
// Replaces, for panels of four tiles with at most `chain_resident_max_rows` tile rows below them (the regime where the chain,
// not the trailing update, bounds the factorisation: all of c2, the last third of c3, every small problem), the tile loop of
// `factor_columns` (potrf.hip) -- i.e. LAPACK dpotrf's panel factorisation behind `gram.solve` (_conditional.py:44) and
// `BlockMatrix2x2._cholesky` (linops/_block.py:233-242).  Same arithmetic, product for product, as the kernels it replaces
// (the row workgroups run the fused panel chain of solve_panel.h, the factor workgroup the tile Cholesky of potrf_tile.h).
//
// Roles (one launch, 512 threads and 152 KB of LDS per workgroup, one workgroup per CU):
//   block 0          FACTOR: for tile column j = 0..3: wait until the four row workgroups of tile j have applied every earlier
//                    column to the diagonal tile (U_j), factor it (L_jj, Linv_j), publish F_j.
//   blocks 1..12     IN-BLOCK ROWS: 32 rows each of tiles 1..3 of the diagonal block.  For every column j left of their own tile t:
//                    wait F_j, solve their rows against tile j (three products: one refinement step), store them, publish
//                    S_tj; wait for the sub-diagonal tiles of column j down to their own (S_ij, i <= t), update their rows of
//                    columns j + 1 .. t; after column t - 1 store their strip of the diagonal tile and publish U_t.
//   blocks 13..      ROWS BELOW the diagonal block, 32 each: per column j wait until the column is complete (F_j and S_ij for
//                    all i), then solve and update exactly as panel_solve_kernel<4, 2, false> does.
// The critical path of a tile step is  tile Cholesky (35 us) -> F -> solve of ONE tile's rows (3 x 8 stages) -> S -> one
// update product (8 stages) -> U -> next tile Cholesky: three hand-overs of ~3 us (release, flag, poll, acquire) where the
// launches had three kernel boundaries.
//
// The row workgroups keep the statically scheduled LDS-DMA ring of solve_panel.h, which prefetches factor stages 4-5 stages
// ahead -- across the points where a column becomes available.  At such a point the workgroup drains its DMA, waits for the
// flag, acquires, and RE-ISSUES the stages that were in flight (same ring slots, same order, so every counted vmcnt wait of
// the schedule still holds): the early copies may have read the factor before it was final and are simply overwritten.
//
// Hand-overs follow MI355X_MICROARCH.md "inter-workgroup visibility": producer -- every storing wave's vmcnt(0), workgroup
// barrier, lane 0: agent release, vmcnt(0), relaxed agent-scope flag store / add; consumer -- lane 0 polls (relaxed agent
// load, s_sleep), agent acquire, every wave's vmcnt(0), workgroup barrier, then plain loads.  Every poll is BOUNDED: a flag
// that does not arrive (a dispatch order this design does not expect) sets the launch's abort word and a negative status
// instead of hanging the device; the host reports it as an error.

#include <climits>

#include "lpgp_internal.h"
#include "kernel_util.h"
#include "potrf_tile.h"
#include "solve_panel.h"

namespace lpgp {

constexpr int CH_SLOT_INTS = 32;          // F[0..3] | S(k, j) at 4 + 3 (k - 1) + j, k = 1..3, j < k | U_k at 12 + k, k = 1..3 | abort at 16 | U_0 at 17
constexpr int CH_SLOTS = 64;
constexpr int CH_SPIN_LIMIT = 4000000;    // polls of ~0.25 us each: one second

struct ChainArgs {
  double* a;               // element (first row, first column) of the panel's diagonal block
  int64_t ld;
  double* linv;            // the panel's four tile inverses, contiguous
  int* slot;               // this launch's flag slot (zeroed)
  int* slot_clear;         // a slot that is not in use: zeroed here for a later launch
  int* info;
  int32_t info_base;
  int32_t n_below;         // workgroups of rows below the diagonal block (32 rows each)
};

__device__ __forceinline__ int ch_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// every thread of the workgroup calls it: returns when cond() holds (lane 0 polls), acquired
template <class Cond>
__device__ __forceinline__ void ch_wait(const ChainArgs& g, Cond cond) {
  if (threadIdx.x == 0) {
    int spins = 0;
    while (!cond()) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > CH_SPIN_LIMIT || ch_load(g.slot + 16) != 0) {
        __hip_atomic_store(g.slot + 16, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // everybody stops waiting
        atomicCAS(g.info, 0, INT_MIN);                                                          // status < 0: not a pivot, a broken hand-over
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TSV_BARRIER();
}

// every thread calls it after its stores: the workgroup's stores are visible to whoever sees the flag
__device__ __forceinline__ void ch_publish(int* word, bool add) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TSV_BARRIER();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (add) __hip_atomic_fetch_add(word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__device__ __forceinline__ int* ch_S(const ChainArgs& g, int k, int j) { return g.slot + 4 + 3 * (k - 1) + j; }
__device__ __forceinline__ int* ch_U(const ChainArgs& g, int k) { return k == 0 ? g.slot + 17 : g.slot + 12 + k; }      // (U_0: only with the fused look-ahead)

// ---- rows: the fused panel chain of solve_panel.h (NT = 4, 32 rows per workgroup, row form) with hand-overs ----
// tr: tile of the diagonal block the rows lie in (1..3), or 4 for rows below it; row_rel: first row relative to the block's first
// base / lane_step / cstep: element (workgroup-local row i, panel column c) of X at base[i * lane_step + c * cstep] -- rows of the
// matrix: (g.a + first row, 1, g.ld); columns of a right-hand side V (X = V^T, the forward substitution's panel step): (V + first
// column * ldv, ldv, 1)
// RG = 2: eight waves, 32 rows per workgroup (one per CU beside the factor workgroup's 152 KB); RG = 1: four waves, 16 rows, 68 KB of
// LDS -- two workgroups per CU (panel_chain_rows_kernel: the rows below a WIDE panel, round 6)
// NP = 4 (round 6): the LOOK-AHEAD update by the previous panel -- A_i -= X_prev,j L_prev,ij^T for this workgroup's rows, the launch
// factor_columns otherwise puts between two chains -- rides in front of the chain (sixteen more products; the form of
// panel_solve_kernel<NT, RG, true, 4>).  It needs nothing of THIS panel, so it runs while the factor workgroup waits for the four
// workgroups of tile 0's rows (tr = 0: a role that exists only with NP > 0), which publish U_0.
template <int RG, int NP = 0>
__device__ __forceinline__ void chain_rows_role(const ChainArgs& g, const int tr, double* base, const int64_t lane_step, const int64_t cstep,
                                                double* smem) {
  constexpr int NT = 4;
  constexpr PsvSched<NT, NP> SCH = psv_make_sched<NT, NP>(4 / RG);
  static_assert(psv_sched_ok<NT, RG, NP>(), "resident chain: broken stage schedule");
  constexpr int XA = RG * 32 * 64;
  double* xa = smem;
  double* ring = smem + XA;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wu = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wu >> 2, s4 = wu & 3;
  const int cc = rg ? 3 - s4 : s4;
  const int li = lane & 15, lj = lane >> 4;
  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) double*)smem);
  const bool below = tr >= NT;
  const double* Lblk = g.a;
  const double* linv = g.linv;
  const int64_t ldl = g.ld;

  auto issue = [&](auto S_) {
    constexpr int s = decltype(S_)::value;
    constexpr PsvProd pd = psv_prod(NT, s / 8, NP);
    constexpr int kt = s % 8, stride = psv_stride<NT, NP>(s);
    constexpr bool tri = pd.kind < 3;
    constexpr int len = tri ? 128 - 16 * kt : 128, col0 = tri ? 16 * kt : 0;
    int64_t ldl_ = ldl;
    // (kind 4: the factor's block LEFT of the diagonal block -- this panel's rows, the previous panel's columns)
    const double* lbase = (pd.kind == 0 || pd.kind == 2) ? linv : (pd.kind == 4 ? Lblk - (int64_t)NP * TILE * ldl : Lblk);
    asm volatile("" : "+s"(lbase), "+s"(ldl_));
    // (the look-ahead products run i-MAJOR here -- product p < NP * NT is tile row i = p / NP of the panel against the previous
    //  panel's column tile j = p % NP -- so that a workgroup that owns rows of tile tr is done with them after (tr + 1) * NP
    //  products; all kind-4 stages have one size, the stage schedule does not care)
    constexpr int pi = pd.kind == 4 ? (s / 8) / (NP > 0 ? NP : 1) : pd.i, pj = pd.kind == 4 ? (s / 8) % (NP > 0 ? NP : 1) : pd.j;
    const double* M = (pd.kind == 0 || pd.kind == 2) ? lbase + (int64_t)pj * TILE * TILE
                                                      : lbase + (int64_t)pi * TILE + (int64_t)pj * TILE * ldl_;
    const int64_t ldm = (pd.kind == 0 || pd.kind == 2) ? (int64_t)TILE : ldl_;
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

  // column j of the diagonal block is ready for THIS workgroup's solve against tile j
  auto wait_column = [&](int j) {
    if (below) {
      ch_wait(g, [&] {
        if (ch_load(g.slot + j) == 0) return false;
        for (int i = j + 1; i < NT; ++i)
          if (ch_load(ch_S(g, i, j)) < 4) return false;
        return true;
      });
    } else {
      ch_wait(g, [&] { return ch_load(g.slot + j) != 0; });
    }
  };

  // fragments: fragment (t, q) = element (row, panel column 128 t + 4 (cc + 4 q) + lj)
  double a[NT][8], x[8];
  double* const pbase = base + (int64_t)(rg * 16 + li) * lane_step + (int64_t)(4 * cc + lj) * cstep;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 8; ++q) a[t][q] = pbase[(t * TILE + q * 16) * cstep];
  // (NP > 0) the first operand image is the previous panel's first solved tile of these rows, negated
  double vp[8];
  if constexpr (NP > 0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) vp[q] = pbase[(-(NP * TILE) + q * 16) * cstep];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TSV_BARRIER();
  } else {
    wait_column(0);                                         // (the loads above are waited for inside)
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

  auto run_product = [&](auto P_, double(&dst)[8], const bool active) {
    constexpr int prod = decltype(P_)::value;
    constexpr bool tri = psv_prod(NT, prod, NP).kind < 3;
    static_for<0, 8>([&](auto KT_) {
      constexpr int kt = decltype(KT_)::value;
      constexpr int s = prod * 8 + kt;
      constexpr int stride = psv_stride<NT, NP>(s);
      constexpr int q0 = tri ? kt : 0;
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
        if (active) {
          static_for<q0, 8>([&](auto Q_) {
            constexpr int q = decltype(Q_)::value;
            dst[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(nf[ks & 1][q], mf[ks & 1], dst[q], 0, 0, 0);
          });
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  };

  // ---- (NP > 0) the fused look-ahead update: the previous panel applied to this workgroup's rows of the panel's columns ----
  // i-major: tile column i of the panel gets its NP products in a row; rows of the diagonal block's tile tr stop after column tr
  // (columns beyond lie above the diagonal) and go straight to their chain -- the stages of the skipped products that are already
  // in flight land in the ring unread, the hand-over below drains and re-issues as at every hand-over.
  static_for<0, NT>([&](auto I_) {
    constexpr int i = decltype(I_)::value;
    if (below || i <= tr) {
      static_for<0, NP>([&](auto JP_) {
        constexpr int jp = decltype(JP_)::value;
        if constexpr (i > 0 || jp > 0) {
          TSV_BARRIER();                                       // the previous product has read the fragment image
#pragma unroll
          for (int q = 0; q < 8; ++q) xown[q * 256] = -vp[q];
        }
        {                                                      // the next operand (the same rows' next solved tile of the previous panel), in flight under this product
          constexpr int jn = (jp + 1) % NP;
          int64_t cs = cstep;
          asm volatile("" : "+s"(cs));                         // (re-formed per product: hoisted, the 32 addresses of the four operand tiles cost 64 registers)
#pragma unroll
          for (int q = 0; q < 8; ++q) vp[q] = pbase[(-(NP * TILE) + jn * TILE + q * 16) * cs];
        }
        run_product(std::integral_constant<int, i * NP + jp>{}, a[i], true);      // A_i -= Xprev_jp Lprev_{i,jp}^T
      });
    }
  });
  if constexpr (NP > 0) {
    if (tr == 0) {
      // rows of the panel's FIRST tile: their strip of the diagonal tile is complete, the factor workgroup may start
#pragma unroll
      for (int q = 0; q < 8; ++q) pbase[q * 16 * cstep] = a[0][q];
      ch_publish(ch_U(g, 0), true);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (stages in flight land in this workgroup's LDS: not beyond its end)
      return;
    }
  }
  static_for<0, NT>([&](auto J_) {
    constexpr int j = decltype(J_)::value;
    constexpr int p0 = psv_first_prod(NT, j, NP);
    const bool solve = j < tr;                               // rows of tile tr are solved against the columns left of it
    if constexpr (j == 0 && NP > 0) {
      // column 0 becomes available (nothing of this panel was needed so far): wait, acquire, re-issue the stages in flight
      wait_column(0);
      static_for<8 * p0, SCH.iss_hi[8 * p0]>(issue);
      TSV_BARRIER();
#pragma unroll
      for (int q = 0; q < 8; ++q) xown[q * 256] = a[0][q];
    }
    if constexpr (j > 0) {
      if (solve) {
        // column j becomes available: drain, wait, acquire, re-issue the stages that were prefetched before it was final
        wait_column(j);
        static_for<8 * p0, SCH.iss_hi[8 * p0]>(issue);
      }
      TSV_BARRIER();                                         // the updates by X_{j-1} have read xa
#pragma unroll
      for (int q = 0; q < 8; ++q) xown[q * 256] = a[j][q];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = 0.0;
    run_product(std::integral_constant<int, p0>{}, x, solve);         // x = X0 = A_j Linv_j^T
    TSV_BARRIER();
#pragma unroll
    for (int q = 0; q < 8; ++q) xown[q * 256] = -x[q];
    run_product(std::integral_constant<int, p0 + 1>{}, a[j], solve);  // a_j = R = A_j - X0 L_jj^T
    TSV_BARRIER();
#pragma unroll
    for (int q = 0; q < 8; ++q) xown[q * 256] = a[j][q];
    run_product(std::integral_constant<int, p0 + 2>{}, x, solve);     // x = X_j
    if (solve) {
#pragma unroll
      for (int q = 0; q < 8; ++q) a[j][q] = x[q];
    }
    if constexpr (j + 1 < NT) {
      if (!below && solve) {
        // in-block rows: L_{tr, j} is final -- out it goes, the rows below and the sibling workgroups of this tile wait for it;
        // then this workgroup's own updates need the sub-diagonal tiles of column j down to its own tile
#pragma unroll
        for (int q = 0; q < 8; ++q) pbase[(j * TILE + q * 16) * cstep] = x[q];
        ch_publish(ch_S(g, tr, j), true);
        ch_wait(g, [&] {
          for (int i = j + 1; i <= tr; ++i)
            if (ch_load(ch_S(g, i, j)) < 4) return false;
          return true;
        });
        static_for<8 * (p0 + 3), SCH.iss_hi[8 * (p0 + 3)]>(issue);
      }
      TSV_BARRIER();
#pragma unroll
      for (int q = 0; q < 8; ++q) xown[q * 256] = -x[q];
      static_for<j + 1, NT>([&](auto I_) {
        constexpr int i = decltype(I_)::value;
        run_product(std::integral_constant<int, p0 + 3 + (i - j - 1)>{}, a[i], solve && i <= tr);   // A_i -= X_j L_ij^T
        if (!below && i == tr && j == tr - 1) {
          // the strip of the diagonal tile is complete: the factor workgroup may start on tile tr
#pragma unroll
          for (int q = 0; q < 8; ++q) pbase[(i * TILE + q * 16) * cstep] = a[i][q];
          ch_publish(ch_U(g, tr), true);
        }
      });
    }
  });
  if (below) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 8; ++q) pbase[(t * TILE + q * 16) * cstep] = a[t][q];
  }
}

// AHEAD = 4: the look-ahead update by the PREVIOUS panel (four tiles wide) rides in front of every row workgroup's chain, four more
// workgroups own the rows of the panel's first tile (blocks 1..4), and the factor workgroup waits for them (U_0) before tile 0.
template <int AHEAD>
__global__ __launch_bounds__(512, 1) void panel_chain_kernel(ChainArgs g) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int b = blockIdx.x;
  constexpr int NB0 = AHEAD ? 4 : 0;                          // workgroups of tile 0's rows
  if (b == 0) {
    // ---- factor role ----
    if (threadIdx.x < CH_SLOT_INTS) g.slot_clear[threadIdx.x] = 0;
    for (int j = 0; j < 4; ++j) {
      if (j > 0 || AHEAD) ch_wait(g, [&] { return ch_load(ch_U(g, j)) >= 4; });
      potrf_tile_body(g.a + (int64_t)j * TILE * (g.ld + 1), g.ld, g.linv + (int64_t)j * TILE * TILE, g.info, g.info_base + j * TILE, sm);
      ch_publish(g.slot + j, false);
      TSV_BARRIER();                                          // (the tile's LDS image is reused by the next load)
    }
    return;
  }
  if (b <= NB0) {
    chain_rows_role<2, AHEAD>(g, 0, g.a + 32 * (b - 1), 1, g.ld, sm);
  } else if (b <= NB0 + 12) {
    const int tr = 1 + (b - NB0 - 1) / 4, w = (b - NB0 - 1) & 3;
    chain_rows_role<2, AHEAD>(g, tr, g.a + (int64_t)tr * TILE + 32 * w, 1, g.ld, sm);
  } else {
    chain_rows_role<2, AHEAD>(g, 4, g.a + (int64_t)4 * TILE + 32 * (int64_t)(b - NB0 - 13), 1, g.ld, sm);
  }
}

// The panel step of the forward substitution that rides inside the factorisation (potrf.hip: ride_panel), for a panel the
// resident chain factors: V[panel rows, 32 columns per workgroup] <- L_KK^{-1} V, the chain of panel_solve_kernel<4, 2, true> with
// the hand-overs of the rows BELOW the block -- it follows the factor workgroup tile by tile through the SAME flags instead of
// waiting for the whole chain kernel, so it ends one tile solve (~15 us) after the chain does, not one launch and a whole panel
// chain (12 + 60 us) after it: the tail of every step of a small problem (N_tot = 1 152: 1.33 ms per step, two such panels).
// Launched AFTER its chain kernel in host order, on the substitution's stream; reads the flags only.
__global__ __launch_bounds__(512, 1) void panel_chain_v_kernel(ChainArgs g, double* V, int64_t ldv) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  chain_rows_role<2>(g, 4, V + (int64_t)blockIdx.x * 32 * ldv, ldv, 1, sm);
}

// The rows BELOW the diagonal block of a panel whose factor and in-block workgroups run in panel_chain_kernel on ANOTHER stream
// (round 6: the two-kernel form of the resident chain for panels with more rows below than one workgroup per CU can hold at
// 152 KB each): 16 rows per workgroup, 68 KB of LDS, two per CU; reads the chain's flags, publishes nothing.
__global__ __launch_bounds__(256, 2) void panel_chain_rows_kernel(ChainArgs g) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  chain_rows_role<1>(g, 4, g.a + (int64_t)4 * TILE + 16 * (int64_t)blockIdx.x, 1, g.ld, sm);
}

// The chain of panel [p0, p0 + 4) of the padded matrix, all rows down to tile T, on `stream`.
// rows_here = false: the factor workgroup and the twelve in-block row workgroups only; the rows below follow through the flags
// in a launch of their own (launch_panel_chain_rows).
// ahead = true: the look-ahead update by the previous panel (columns [p0 - 4, p0), all rows from p0 on) is part of the launch.
int launch_panel_chain(lpgp_ctx* ctx, hipStream_t stream, lpgp_mat* mat, int p0, int T, int* d_info, bool rows_here, bool ahead) {
  LPGP_CHECK(!ahead || (p0 >= 4 && rows_here), "resident chain: the fused look-ahead needs a previous panel of four tiles and the rows in the same launch");
  const int64_t ld = mat->cap;
  if (!ctx->d_chain_flags) {
    LPGP_HIP(hipMalloc(&ctx->d_chain_flags, (size_t)CH_SLOTS * CH_SLOT_INTS * sizeof(int)));
    LPGP_HIP(hipMemsetAsync(ctx->d_chain_flags, 0, (size_t)CH_SLOTS * CH_SLOT_INTS * sizeof(int), stream));
  }
  ChainArgs g;
  g.a = mat->a + (int64_t)p0 * TILE * (ld + 1);
  g.ld = ld;
  g.linv = mat->linv + (int64_t)p0 * TILE * TILE;
  const int64_t n = ctx->chain_launches++;
  g.slot = ctx->d_chain_flags + (n % CH_SLOTS) * CH_SLOT_INTS;
  ctx->chain_last_slot = g.slot;
  ctx->chain_last_p0 = p0;
  g.slot_clear = ctx->d_chain_flags + ((n + CH_SLOTS / 2) % CH_SLOTS) * CH_SLOT_INTS;
  g.info = d_info;
  g.info_base = p0 * TILE;
  g.n_below = rows_here ? (T - p0 - 4) * (TILE / 32) : 0;
  const size_t shmem = (size_t)TILE_LDS_DOUBLES * sizeof(double);
  static_assert((size_t)TILE_LDS_DOUBLES >= (size_t)(2 * 32 * 64 + TSV_RING), "resident chain: the tile image must cover the row role's LDS");
  void (*kfn)(ChainArgs) = ahead ? panel_chain_kernel<4> : panel_chain_kernel<0>;
  LPGP_TRY_RC(ensure_lds_attr(ctx, reinterpret_cast<const void*>(kfn), shmem));
  // algorithmic flops of the panel's chain: four tile Choleskys + the triangular solve of the rows below against the block
  // (+ the look-ahead update of the panel's columns by the previous panel, K = 512, where it rides along)
  prof_begin(ctx, stream, LPGP_K_PANEL, 4.0 * TILE * TILE * TILE / 3.0 + (double)(T - p0 - 1) * TILE * 512.0 * 512.0 / 2.0 +
                                            (ahead ? 2.0 * 512.0 * ((double)(T - p0) * TILE * 512.0 - 0.5 * 512.0 * 511.0) : 0.0), 0.0);
  hipLaunchKernelGGL(kfn, dim3((unsigned)((ahead ? 17 : 13) + g.n_below)), dim3(512), shmem, stream, g);
  prof_end(ctx, stream);
  LPGP_HIP(hipGetLastError());
  return 0;
}

// rows below the diagonal block of panel p0 (down to tile T), following the chain launched LAST (launch_panel_chain(..., rows_here = false))
int launch_panel_chain_rows(lpgp_ctx* ctx, hipStream_t stream, lpgp_mat* mat, int p0, int T, int* d_info) {
  LPGP_CHECK(ctx->chain_last_slot && ctx->chain_last_p0 == p0, "resident chain: no chain launch of panel %d to follow", p0);
  const int64_t ld = mat->cap;
  const int below = T - p0 - 4;
  if (below <= 0) return 0;
  ChainArgs g;
  g.a = mat->a + (int64_t)p0 * TILE * (ld + 1);
  g.ld = ld;
  g.linv = mat->linv + (int64_t)p0 * TILE * TILE;
  g.slot = ctx->chain_last_slot;
  g.slot_clear = nullptr;
  g.info = d_info;
  g.info_base = p0 * TILE;
  g.n_below = 0;
  const size_t shmem = (size_t)(32 * 64 + TSV_RING) * sizeof(double);          // one row group's fragment image + the stage ring: 69 632 B
  LPGP_TRY_RC(ensure_lds_attr(ctx, reinterpret_cast<const void*>(&panel_chain_rows_kernel), shmem));
  prof_begin(ctx, stream, LPGP_K_PANEL, (double)below * TILE * 512.0 * 512.0, 0.0);
  hipLaunchKernelGGL(panel_chain_rows_kernel, dim3((unsigned)(below * (TILE / 16))), dim3(256), shmem, stream, g);
  prof_end(ctx, stream);
  LPGP_HIP(hipGetLastError());
  return 0;
}

// V: first row of the panel's rows of the right-hand side (rows p0 * 128 ...), `cols` columns (a multiple of 128); the chain of
// panel p0 must be the LAST one launched (launch_panel_chain), its flag slot is the one this launch follows.
int launch_panel_chain_v(lpgp_ctx* ctx, hipStream_t stream, lpgp_mat* mat, int p0, double* V, int64_t ldv, int64_t cols, int* d_info) {
  LPGP_CHECK(ctx->chain_last_slot && ctx->chain_last_p0 == p0, "resident chain: no chain launch of panel %d to follow", p0);
  const int64_t ld = mat->cap;
  ChainArgs g;
  g.a = mat->a + (int64_t)p0 * TILE * (ld + 1);
  g.ld = ld;
  g.linv = mat->linv + (int64_t)p0 * TILE * TILE;
  g.slot = ctx->chain_last_slot;
  g.slot_clear = nullptr;
  g.info = d_info;
  g.info_base = p0 * TILE;
  g.n_below = 0;
  const size_t shmem = (size_t)(2 * 32 * 64 + TSV_RING) * sizeof(double);          // the row role's fragment image + stage ring
  LPGP_TRY_RC(ensure_lds_attr(ctx, reinterpret_cast<const void*>(&panel_chain_v_kernel), shmem));
  prof_begin(ctx, stream, LPGP_K_PANEL, (double)cols * 512.0 * 512.0, 0.0);
  hipLaunchKernelGGL(panel_chain_v_kernel, dim3((unsigned)(cols / 32)), dim3(512), shmem, stream, g, V, ldv);
  prof_end(ctx, stream);
  LPGP_HIP(hipGetLastError());
  return 0;
}

}  // namespace lpgp
