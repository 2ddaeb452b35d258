// Blocked right-looking Cholesky, block append (Schur complement), multi-RHS triangular
// solves.  Replaces probnum `LinearOperator.cholesky/solve` -> LAPACK dpotrf/dpotrs/dtrtrs
// (_conditional.py:44,108,228) and `BlockMatrix2x2.schur/_cholesky/schur_update`
// (linops/_block.py:192-242): appending an observation block continues the factorisation
// of the padded in-place matrix instead of building a nested block operator.
//
// Structure per panel of `nb` columns (nb = 4 tiles of 128 by default):
//   for each 128-wide tile column:  potrf_tile (one workgroup, LDS resident; also emits
//   the explicit inverse of the diagonal tile)  ->  panel rows below: X = A * L^-T in place, a product with that
//   inverse refined once against the tile itself (tile_solve_kernel, solve.hip)  ->  rank-128 update of the rest of the panel;
//   then the rank-nb SYRK trailing update (gemm.hip), split into the next panel's columns
//   (high-priority stream, followed immediately by the next panel factorisation) and the
//   remainder (second stream): look-ahead of one panel.

#include <climits>
#include <cstdlib>

#include <algorithm>

#include "lpgp_internal.h"
#include "kernel_util.h"
#include "potrf_tile.h"

namespace lpgp {

__global__ __launch_bounds__(TILE_WAVES * 64) void potrf_tile_kernel(double* __restrict__ a, int64_t lda,
                                                          double* __restrict__ linv, int* __restrict__ info,
                                                          int info_base) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  potrf_tile_body(a, lda, linv, info, info_base, sm);
}

int debug_tile_xcc(int32_t* out8, int reset) {
  int h[8];
  LPGP_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tile_xcc_hist), sizeof(h)));
  for (int i = 0; i < 8; ++i) out8[i] = h[i];
  if (reset) {
    int z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    LPGP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_tile_xcc_hist), z, sizeof(z)));
  }
  return 0;
}

int launch_potrf_tile(lpgp_ctx* ctx, hipStream_t stream, double* a, int64_t lda, double* linv,
                      int* d_info, int info_base) {
  const size_t shmem = (size_t)TILE_LDS_DOUBLES * sizeof(double);
  LPGP_TRY_RC(ensure_lds_attr(ctx, reinterpret_cast<const void*>(&potrf_tile_kernel), shmem));
  prof_begin(ctx, stream, LPGP_K_POTRF_TILE, (double)TILE * TILE * TILE / 3.0, 0.0);
  hipLaunchKernelGGL(potrf_tile_kernel, dim3(1), dim3(TILE_WAVES * 64), shmem, stream, a, lda, linv, d_info, info_base);
  prof_end(ctx, stream);
  LPGP_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------
static inline GemmArgs mk(const double* A, int64_t lda, const double* B, int64_t ldb, double* C,
                          int64_t ldc, int mt, int nt, int k, double alpha, double beta, int tri) {
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.mt = mt; g.nt = nt; g.k = k; g.alpha = alpha; g.beta = beta;
  g.tri = tri;
  return g;
}

// X (mt tiles of rows x one tile column) <- X * Ljj^{-T}, in place; Ljj = diagonal tile jt of the factor
static inline int panel_trsm(lpgp_ctx* ctx, hipStream_t st, lpgp_mat* mat, int jt, double* X, int mt) {
  const int64_t ld = mat->cap;
  return launch_trsm_tile(ctx, st, X, ld, mat->linv + (int64_t)jt * TILE * TILE, mat->a + (int64_t)jt * TILE * (ld + 1), ld, mt,
                          LPGP_K_TRSM);
}

// ---- ride-along forward substitution (round 5) -------------------------------------------------------------------
// `lpgp_potrf_predict`: V = L^{-1} K_Xx is computed INSIDE the factorisation instead of after it.  A panel step of the
// forward substitution -- V[panel rows] <- L_KK^{-1} V[panel rows] (fused panel chain), then V[rows below] -=
// L[rows below, panel] V[panel rows] -- needs nothing but the panel's own columns of L, which are final as soon as
// the panel's chain is (its tile solves run down to the last row).  So every panel of the factorisation -- the old
// panels a block append pushes its new rows through included -- hands its columns over by ONE event, and the
// substitution's steps follow on a stream of their own (`ride->stream`), as far behind the factorisation as the chip
// leaves them: they are never on the factorisation's critical path, they fill the phase where the panel chain bounds
// it (the last third of c3, all of c2), and the second pipeline with its start-up and its own chain-bound tail is gone.
// The kernels are the forward substitution's (trsm_lower_blocked), launch for launch; only the schedule differs.
// (Reference: the same algebra as `BlockMatrix2x2.L_A_inv_B`, linops/_block.py:203-207, applied to K_Xx panel by panel.)
struct Ride {
  double* v = nullptr;       // K_Xx -> V, padded rows x m_pad, column-major
  int64_t ldv = 0;
  int mtl = 0;               // tile columns (m_pad / 128)
  hipStream_t stream = nullptr;     // the substitution's stream ...
  hipStream_t stream2 = nullptr;    // ... and, optionally, a second one: the right-hand side's columns are independent, so its two halves
                                    // run as two sequences of [panel chain, update]; while one half is in its latency-bound
                                    // panel chain the other one's update has the chip (the look-ahead of trsm_lower_blocked
                                    // without splitting any launch)
  hipEvent_t ev[2] = {nullptr, nullptr};
  int it = 0;
  // The gate: while the trailing update bounds the factorisation the chip is full anyway, and substitution steps released
  // then only take from the update what they gain (kernel trace, profiles/r05_fused_*: with every step released at once the
  // factorisation stretches over the whole step and STILL ends with its chain-bound panels, because the substitution has
  // kept pace and has no work left to fill them with).  So the steps are held back -- queued here -- until at most
  // `gate_pct` per cent of the tile rows are left to factor; from then on the substitution's big early updates run beside
  // the factorisation's chain-bound last part.  gate_pct >= 100: no gate.  Measured (ms per step; none / 58 / 48 / 36 %
  // at c3: 53.4 / 52.7 / 52.1 / 52.6 against 55.4 for the two pipelines; c2 8.49 / - / 8.11 / - against 9.37).
  int gate_pct = 100;
  // Two-level form for large factors (the scheme of trsm_lower_two_level): inside an OUTER block of `outer_t` tile rows the
  // rank-nb updates touch the block's own rows only; everything below is updated once per block with K = outer_t * 128 (every
  // C tile of the right-hand side is then read and written once per 4 096 rows of contraction instead of once per 512: at c4,
  // 66 560 x 16 512 doubles = 8.8 GB per pass).  0: every panel updates all rows below it.
  int outer_t = 0;
  int blk_q0 = 0;            // first tile row of the outer block the steps are in
  bool open = true;
  std::vector<std::pair<int, int>> held;
  int64_t chain_launches0 = 0;   // resident chain launches before this factorisation (flag slots are recycled half a ring later)
};
static int ride_panel_now(lpgp_ctx* ctx, lpgp_mat* mat, int T, int p0, int p1, Ride* rd, bool sync_first, bool resident = false);

// panel [p0, p1) of the factor is final on the panel stream from here on: enqueue its substitution step(s)
// (resident: the panel's chain was the resident kernel, launched last on the panel stream -- the step may follow it flag by flag)
static int ride_panel(lpgp_ctx* ctx, lpgp_mat* mat, int T, int p0, int p1, Ride* rd, bool old_panel = false, bool resident = false) {
  // (steps of OLD panels -- a block append pushes its rows through them before it factors anything -- may pass the gate: they fill
  //  the start-up of the append, the first panel chain on an otherwise idle chip; ride_old_ungated)
  if (!rd->open && old_panel && ctx->ride_old_ungated && rd->held.empty()) return ride_panel_now(ctx, mat, T, p0, p1, rd, true);
  if (!rd->open) {
    if ((int64_t)(T - p1) * 100 > (int64_t)rd->gate_pct * T && p1 < T) {
      rd->held.emplace_back(p0, p1);
      return 0;
    }
    rd->open = true;
    bool first = true;
    for (auto& h : rd->held) {
      LPGP_TRY(ride_panel_now(ctx, mat, T, h.first, h.second, rd, first));
      first = false;
    }
    const bool none_held = rd->held.empty();
    rd->held.clear();
    return ride_panel_now(ctx, mat, T, p0, p1, rd, true, resident && none_held);
  }
  return ride_panel_now(ctx, mat, T, p0, p1, rd, true, resident);
}

static int ride_panel_now(lpgp_ctx* ctx, lpgp_mat* mat, int T, int p0, int p1, Ride* rd, bool sync_first, bool resident) {
  const int64_t ld = mat->cap, tb = TILE;
  const double* a = mat->a;
  const bool two = rd->stream2 != nullptr && rd->stream2 != rd->stream && rd->mtl >= 8;
  // A panel of the RESIDENT chain, a right-hand side of few columns: the panel step does not wait for the chain kernel to end
  // -- it follows the factor workgroup through the chain's own flags (chain.hip: panel_chain_v_kernel) and ends one tile solve
  // after it.  Its workgroups wait on the chip, one per CU, hence the bound on their number; the flag slot is recycled 32 chain
  // launches later, hence the bound on the resident launches of one factorisation (the substitution's stream may lag them).
  // (Wide right-hand sides stay behind the event: c3's 132 workgroups following the LAST panel -- no update beside it they could be
  //  in the way of -- made the step 0.5 ms slower, 50.7-50.9 against 50.2-50.3: by then the substitution lags the chain anyway,
  //  and 32 columns per workgroup is the slower panel step of the two when nothing is left to overlap with.)
  const bool vchain = resident && sync_first && !two && ctx->fused_solve && p1 - p0 == 4 && ctx->ride_vchain_max_wgs > 0 &&
                      rd->mtl * 4 <= ctx->ride_vchain_max_wgs && ctx->chain_last_p0 == p0 && ctx->chain_launches - rd->chain_launches0 <= 24;
  // ... and are not dispatched before the chain kernel itself can be (everything in front of it on the panel stream is done):
  // they would only hold their CUs waiting.  (Under a profiler that SERIALISES kernels -- rocprofv3 --pmc -- a follower that is
  // picked before its chain kernel waits out its poll limit and the step fails with a negative status: LPGP_RIDE_VCHAIN=0 there.)
  if (vchain && rd->stream != ctx->s_main && ctx->ride_vchain_pre) LPGP_HIP(hipStreamWaitEvent(rd->stream, ctx->ev_chain_pre, 0));
  if (vchain)
    LPGP_TRY(launch_panel_chain_v(ctx, rd->stream, mat, p0, rd->v + (int64_t)p0 * tb, rd->ldv, (int64_t)rd->mtl * tb, ctx->d_info_cur));
  if (sync_first && (rd->stream != ctx->s_main || two)) {
    hipEvent_t ev = rd->ev[rd->it++ & 1];
    LPGP_HIP(hipEventRecord(ev, ctx->s_main));
    if (rd->stream != ctx->s_main) LPGP_HIP(hipStreamWaitEvent(rd->stream, ev, 0));
    if (two && rd->stream2 != ctx->s_main) LPGP_HIP(hipStreamWaitEvent(rd->stream2, ev, 0));
  }
  const int halves = two ? 2 : 1;
  for (int q0 = p0; q0 < p1; q0 += 4) {
    const int q1 = (q0 + 4 < p1) ? q0 + 4 : p1;
    // Rows this step updates.  Two-level: down to the end of the current outer block -- plus a margin of three tile rows,
    // because the steps (at most four tile rows each, on the panel grid of the factorisation, which an appended block
    // shifts) need not end on the block grid: a step that starts before the block's end E ends before E + 4, so with the
    // margin every row below E + 3 has all of the block's updates when the block closes, and the ONE outer update of the
    // block (K = its height) takes the rows from E + 3 on.
    int lim = T, outer_from = -1, outer_q0 = 0;
    if (rd->outer_t > 0) {
      const int E = rd->blk_q0 + rd->outer_t;
      lim = std::min(T, E + 3);
      if (q1 >= E || q1 >= T) {
        outer_from = lim; outer_q0 = rd->blk_q0;
        rd->blk_q0 = q1;
      }
    }
    for (int h = 0; h < halves; ++h) {
      hipStream_t sV = h == 0 ? rd->stream : rd->stream2;
      const int c0 = (h == 0) ? 0 : rd->mtl / 2, c1 = (two && h == 0) ? rd->mtl / 2 : rd->mtl;       // tile columns of this half
      double* vh = rd->v + (int64_t)c0 * tb * rd->ldv;
      const int mtl = c1 - c0;
      double* Vq = vh + (int64_t)q0 * tb;
      if (vchain) {
        // (done by panel_chain_v_kernel above)
      } else if (ctx->fused_solve) {
        LPGP_TRY(launch_trsv_panel(ctx, sV, Vq, rd->ldv, mat->linv + (int64_t)q0 * tb * tb, a + (int64_t)q0 * tb * (ld + 1), ld, q1 - q0, mtl,
                                   LPGP_K_PANEL));
      } else {
        for (int jt = q0; jt < q1; ++jt) {
          double* Vj = vh + (int64_t)jt * tb;
          LPGP_TRY(launch_trsv_tile(ctx, sV, Vj, rd->ldv, mat->linv + (int64_t)jt * tb * tb, a + (int64_t)jt * tb * (ld + 1), ld, mtl, LPGP_K_TRSM));
          if (jt + 1 < q1)
            LPGP_TRY(launch_gemm(ctx, sV, 0, 1,
                                 mk(a + (int64_t)(jt + 1) * tb + (int64_t)jt * tb * ld, ld, Vj, rd->ldv, vh + (int64_t)(jt + 1) * tb, rd->ldv,
                                    q1 - jt - 1, mtl, TILE, -1.0, 1.0, 0),
                                 LPGP_K_GEMM));
        }
      }
      if (q1 < lim) {
        GemmArgs g = mk(a + (int64_t)q1 * tb + (int64_t)q0 * tb * ld, ld, Vq, rd->ldv, vh + (int64_t)q1 * tb, rd->ldv, lim - q1, mtl,
                        (q1 - q0) * TILE, -1.0, 1.0, 0);
        g.occ3 = ctx->ride_occ3;
        LPGP_TRY(launch_gemm(ctx, sV, 0, 1, g, LPGP_K_GEMM));
      }
      if (outer_from >= 0 && outer_from < T) {
        // the outer block [outer_q0, q1) is solved: everything from its margin on in ONE update with K = its height
        GemmArgs g = mk(a + (int64_t)outer_from * tb + (int64_t)outer_q0 * tb * ld, ld, vh + (int64_t)outer_q0 * tb, rd->ldv,
                        vh + (int64_t)outer_from * tb, rd->ldv, T - outer_from, mtl, (q1 - outer_q0) * TILE, -1.0, 1.0, 0);
        g.occ3 = ctx->ride_occ3;
        LPGP_TRY(launch_gemm(ctx, sV, 0, 1, g, LPGP_K_GEMM));
      }
    }
  }
  return 0;
}

// Factor tile columns [c0, cl) (all rows down to T) right-looking by panels of nb columns with a
// look-ahead of one panel; the rank-nb updates touch only columns < cl (cl == T: the whole trailing
// matrix).  `dep`: event behind the last write to columns [c0 + nb, cl) by an earlier launch on
// another stream (null: none); the first panel chain does not wait for it, the first update does.
// On return the panel stream is behind every update of the call.
// TR >= T: the matrix has TR tile ROWS (rows beyond tile T belong to a block that is solved and updated but never factored: the
// augmented form of potrf_predict_blocked); T stays the limit of the COLUMNS.
static int factor_columns(lpgp_ctx* ctx, lpgp_mat* mat, int T, int c0, int cl, hipStream_t sU, hipEvent_t dep, Ride* ride, int TR) {
  const int64_t ld = mat->cap;
  double* a = mat->a;
  const int nbt = (int)(ctx->nb / TILE);
  const int64_t tb = TILE;
  hipStream_t sP = ctx->s_main;
  bool dep_pending_p = dep != nullptr, dep_pending_u = dep != nullptr;
  // Panel width schedule: while the trailing matrix is large the pipeline is bound by the
  // SYRK update, whose efficiency grows with K (46 TFLOP/s at K = 512, ~52 at K = 1024, in
  // situ), and the longer panel chain hides behind it; once the update gets short the
  // pipeline is bound by the panel chain and the narrow panel wins.
  const bool la = ctx->lookahead != 0;
  auto width_at = [&](int p0) {
    const int rem = T - p0;
    return (ctx->nb_big > ctx->nb && rem > ctx->nb_big_min_tiles) ? (int)(ctx->nb_big / TILE) : nbt;
  };
  int have_upd_event = 0;          // ev_upd[...] recorded for the previous remainder update
  hipEvent_t last_upd = nullptr;
  int it = 0;
  bool ahead_next = false;
  for (int p0 = c0; p0 < cl; ++it) {
    const int w0 = width_at(p0);
    const int p1 = (p0 + w0 < cl) ? p0 + w0 : cl;
    // `dep` covers the columns from c0 + nb on: a first panel wider than nb needs it before its chain
    if (dep_pending_p && p1 - p0 > nbt) {
      LPGP_HIP(hipStreamWaitEvent(sP, dep, 0));
      dep_pending_p = false;
    }
    // panel factorisation on sP: the resident chain (one launch, chain.hip) where the chain is what bounds the pipeline ...
    const bool resident = ctx->chain_resident_max_rows >= 0 && p1 - p0 == 4 && TR - p1 <= ctx->chain_resident_max_rows && !ctx->distributed();
    const bool ahead = ahead_next;          // the previous iteration left this panel's look-ahead update to its chain kernel
    ahead_next = false;
    LPGP_CHECK(!ahead || resident, "factor_columns: a look-ahead update was left to a chain kernel that is not launched");
    if (resident) {
      if (ride && ride->stream != sP && ctx->ride_vchain_pre) LPGP_HIP(hipEventRecord(ctx->ev_chain_pre, sP));      // (see ride_panel_now: the step that follows the chain's flags)
      LPGP_TRY(launch_panel_chain(ctx, sP, mat, p0, TR, ctx->d_info_cur, true, ahead));
    }
    // ... a WIDER panel as two launches (round 6): the factor workgroup and the in-block rows (13 workgroups of 152 KB: thirteen CUs)
    // here, the rows below -- 16 per workgroup, 68 KB, two per CU -- on the outer-update stream, which is idle at these sizes and
    // CU-masked: its workgroups wait ON the chip for the factor workgroup's flags and must not sit on the CUs that workgroup needs
    // (the mask leaves it `reserve` of them); the launch is ordered behind everything in front of the chain kernel (an event), so
    // that it is not dispatched before the chain kernel can be.  The substitution's follower is not used with this form.
    hipStream_t sR = ctx->s_outer;
    const bool resident2 = !resident && ctx->chain_resident2_max_rows > 0 && ctx->chain_resident_max_rows >= 0 && p1 - p0 == 4 && TR - p1 > 0 &&
                           TR - p1 <= ctx->chain_resident2_max_rows && !ctx->distributed() && !ctx->single_stream && sR && sR != sP && ctx->ev_chain_rows;
    if (resident2) {
      LPGP_HIP(hipEventRecord(ctx->ev_chain_pre, sP));
      LPGP_TRY(launch_panel_chain(ctx, sP, mat, p0, TR, ctx->d_info_cur, false));
      LPGP_HIP(hipStreamWaitEvent(sR, ctx->ev_chain_pre, 0));
      LPGP_TRY(launch_panel_chain_rows(ctx, sR, mat, p0, TR, ctx->d_info_cur));
      LPGP_HIP(hipEventRecord(ctx->ev_chain_rows, sR));
      LPGP_HIP(hipStreamWaitEvent(sP, ctx->ev_chain_rows, 0));
    }
    // ... else tile by tile
    for (int jt = p0; jt < p1 && !resident && !resident2; ++jt) {
      double* dj = a + (int64_t)jt * tb * (ld + 1);
      double* linv = mat->linv + (int64_t)jt * tb * tb;
      LPGP_TRY(launch_potrf_tile(ctx, sP, dj, ld, linv, ctx->d_info_cur, jt * TILE));
      if (jt + 1 < TR) {
        double* X = dj + tb;     // rows below, same tile column
        LPGP_TRY(panel_trsm(ctx, sP, mat, jt, X, TR - jt - 1));
        if (jt + 1 < p1)
          LPGP_TRY(launch_gemm(ctx, sP, 0, 0,
                               mk(X, ld, X, ld, a + (int64_t)(jt + 1) * tb * (ld + 1), ld, TR - jt - 1,
                                  p1 - jt - 1, TILE, -1.0, 1.0, 2),
                               LPGP_K_SYRK_PANEL));
      }
    }
    if (ride) LPGP_TRY(ride_panel(ctx, mat, T, p0, p1, ride, false, resident));          // columns [p0, p1) are final: their substitution step follows on the ride stream
    if (p1 >= cl) break;
    const int K = (p1 - p0) * TILE;
    const double* P = a + (int64_t)p1 * tb + (int64_t)p0 * tb * ld;      // panel rows below
    if (dep_pending_p) {
      LPGP_HIP(hipStreamWaitEvent(sP, dep, 0));
      dep_pending_p = false;
    }
    if (!la) {
      LPGP_TRY(launch_gemm(ctx, sP, 0, 0,
                           mk(P, ld, P, ld, a + (int64_t)p1 * tb * (ld + 1), ld, TR - p1, cl - p1, K, -1.0, 1.0, 1),
                           LPGP_K_SYRK));
      p0 = p1;
      continue;
    }
    const int w1 = width_at(p1);
    const int p2 = (p1 + w1 < cl) ? p1 + w1 : cl;
    // Estimated duration of the remainder update (b) at 50 TFLOP/s against that of the next panel
    // chain: decides who bounds the pipeline from here on.
    const double remc = (double)(cl - p2), remr = (double)(TR - p2);         // (b): remr x remc lower trapezoid
    const double t_b_us = (remr * remc - 0.5 * remc * (remc - 1.0)) * (2.0 * TILE * TILE * (double)K / 50e6);
    const double t_chain_us = ctx->chain_us_tile * (double)(p2 - p1) + ctx->chain_us_fixed;
    const bool chain_bound = t_b_us < t_chain_us;
    // While the CHAIN bounds it, (b) starts only when (a) is COMPLETE: (a) is on the critical
    // path (the next panel waits for it), (b) is not, and launched together they share the chip by
    // workgroup count -- measured at panel 20 of c3: (a) took 229 us next to (b) instead of
    // ~70 us alone.  While the UPDATE bounds it, (b) is released with the panel and (a) runs
    // underneath it (alone it would leave part of the chip idle).
    hipEvent_t evp = ctx->ev_panel[it & 1];
    // Round 6: where the NEXT panel's chain is the resident kernel (and both panels are four tiles wide), the look-ahead update (a)
    // is not a launch: that kernel's row workgroups apply it to their own rows in front of their chain (chain.hip, AHEAD = 4) --
    // one launch per panel where it used to be chain, (a), chain, and no (a) that waits for slots beside the substitution's
    // long updates (two of the nine in a c3 step took 0.98 and 0.48 ms: profiles/r05_fused_final_trace_compact.txt.gz).  Only
    // where at least `chain_ahead_min_rows` tile rows lie below the next panel: the fused form costs the chain ~35 us (the
    // factor workgroup waits for four products of tile 0's rows), a launch of (a) on a few tiles less than that.
    const bool fuse_next = ctx->chain_ahead && p1 - p0 == 4 && p2 - p1 == 4 && ctx->chain_resident_max_rows >= 0 &&
                           TR - p2 <= ctx->chain_resident_max_rows && TR - p2 >= ctx->chain_ahead_min_rows && !ctx->distributed();
    if (!chain_bound || fuse_next) LPGP_HIP(hipEventRecord(evp, sP));
    // (a) next panel's columns on sP; they were last written by the previous remainder update
    if (have_upd_event) LPGP_HIP(hipStreamWaitEvent(sP, ctx->ev_upd[(it + 1) & 1], 0));
    if (fuse_next) {
      ahead_next = true;
    } else {
      LPGP_TRY(launch_gemm(ctx, sP, 0, 0,
                           mk(P, ld, P, ld, a + (int64_t)p1 * tb * (ld + 1), ld, TR - p1, p2 - p1, K, -1.0, 1.0, 3),
                           LPGP_K_SYRK_AHEAD));
      if (chain_bound) LPGP_HIP(hipEventRecord(evp, sP));
    }
    // (b) remainder on an update stream.  Once its estimated duration even on the narrow stream
    // (which leaves a quarter of the CUs to the panel chain) is below that of the chain, it moves
    // there.
    if (p2 < cl) {
      const double narrow_frac = ctx->cus > 0 ? (double)ctx->cus / (double)(ctx->cus - ctx->reserve_narrow) : 1.0;
      // (a riding substitution that runs its second half on the update stream has it to itself once its gate is open: the
      //  remainder updates move to the narrow stream from then on)
      const bool ride_has_upd = ride && ride->open && ride->stream2 == sU;
      hipStream_t sB = (ctx->s_upd_narrow && (t_b_us * narrow_frac < t_chain_us || ride_has_upd)) ? ctx->s_upd_narrow : sU;
      // (round 6, option ride_b_on_ride: once the gate is open the remainder update queues on the SUBSTITUTION's stream, behind the
      //  step of the panel it belongs to: one update grid at a time -- the factorisation's and the substitution's launches alternate
      //  instead of sharing the chip -- with the panel chain beside it)
      if (ctx->ride_b_on_ride && ride && ride->open && ride->stream != sP && ride->held.empty()) sB = ride->stream;
      if (last_upd) LPGP_HIP(hipStreamWaitEvent(sB, last_upd, 0));     // behind the previous remainder update
      if (dep_pending_u) {
        LPGP_HIP(hipStreamWaitEvent(sB, dep, 0));
        dep_pending_u = false;
      }
      const double* P2 = a + (int64_t)p2 * tb + (int64_t)p0 * tb * ld;
      LPGP_HIP(hipStreamWaitEvent(sB, evp, 0));
      GemmArgs gb = mk(P2, ld, P2, ld, a + (int64_t)p2 * tb * (ld + 1), ld, TR - p2, cl - p2, K, -1.0, 1.0, 1);
      gb.occ3 = t_b_us > ctx->gemm3_margin * t_chain_us;       // (with gemm3_fact: three workgroups per CU only while the chain beside it has slack)
      LPGP_TRY(launch_gemm(ctx, sB, 0, 0, gb, LPGP_K_SYRK));
      LPGP_HIP(hipEventRecord(ctx->ev_upd[it & 1], sB));
      have_upd_event = 1;
      last_upd = ctx->ev_upd[it & 1];
    } else {
      have_upd_event = 0;
    }
    p0 = p1;
  }
  // join: sP must not run ahead of the last remainder update
  if (la && last_upd) LPGP_HIP(hipStreamWaitEvent(sP, last_upd, 0));
  return 0;
}

// Factor tile columns [t_done, T) of the padded matrix; columns [0, t_done) already hold L.
static int potrf_blocked_impl(lpgp_ctx* ctx, lpgp_mat* mat, int64_t t_done64, int64_t T64, int32_t* info, Ride* ride, int TR = 0);
int potrf_blocked(lpgp_ctx* ctx, lpgp_mat* mat, int64_t t_done64, int64_t T64, int32_t* info) {
  return potrf_blocked_impl(ctx, mat, t_done64, T64, info, nullptr);
}

// dst[c + r * ldd] = src[r + c * lds] for r < rows, c < cols (both multiples of 32): 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_kernel(double* __restrict__ dst, int64_t ldd, const double* __restrict__ src, int64_t lds) {
  __shared__ double t[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t r0 = (int64_t)blockIdx.x * 32, c0 = (int64_t)blockIdx.y * 32;
#pragma unroll
  for (int k = 0; k < 4; ++k) t[ty + 8 * k][tx] = src[r0 + tx + (c0 + ty + 8 * k) * lds];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) dst[c0 + tx + (r0 + ty + 8 * k) * ldd] = t[tx][ty + 8 * k];
}
static int transpose(hipStream_t st, double* dst, int64_t ldd, const double* src, int64_t lds, int64_t rows, int64_t cols) {
  for (int64_t c0 = 0; c0 < cols; c0 += 32 * 65535) {
    const int64_t nc = std::min<int64_t>(cols - c0, 32 * 65535);
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)(rows / 32), (unsigned)(nc / 32)), dim3(256), 0, st, dst + c0, ldd, src + c0 * lds, lds);
  }
  LPGP_HIP(hipGetLastError());
  return 0;
}

// The factorisation of tile columns [t_done, T) with the forward substitution of `v` (T * 128 rows x m_pad columns, leading
// dimension ldv) riding inside it: on return (everything enqueued, the panel stream behind all of it) v holds L^{-1} v for
// the WHOLE factor, old panels included.  Status as with info == nullptr: the matrix's sticky word.
int potrf_predict_blocked(lpgp_ctx* ctx, lpgp_mat* mat, int64_t t_done, int64_t T, double* v, int64_t ldv, int64_t m_pad) {
  Ride rd;
  rd.v = v; rd.ldv = ldv; rd.mtl = (int)(m_pad / TILE);
  rd.chain_launches0 = ctx->chain_launches;
  // the ride streams: update streams that are idle during a factorisation of this size (HIP multiplexes a process's
  // streams over four hardware queues: no new stream).  ride_stream = first + 8 * second (second 7: none):
  // 0 s_outer (masked like s_upd), 1 s_upd_all (unmasked), 2 s_upd_narrow, 3 the panel stream itself, 4 s_upd
  hipStream_t cand[5] = {ctx->s_outer, ctx->s_upd_all, ctx->s_upd_narrow, ctx->s_main, ctx->s_upd};
  const int i1 = ctx->ride_stream & 7, i2 = (ctx->ride_stream >> 3) & 7;
  rd.stream = (i1 < 5 && cand[i1]) ? cand[i1] : ctx->s_main;
  rd.stream2 = (i2 < 5 && cand[i2]) ? cand[i2] : nullptr;
  if (ctx->single_stream) { rd.stream = ctx->s_main; rd.stream2 = nullptr; }
  rd.ev[0] = ctx->ev_ride[0]; rd.ev[1] = ctx->ev_ride[1];
  // Policy by size (measured, MEASUREMENTS.md round 5).  Up to 192 tile rows the steps are held back until 65 % of the tile rows are
  // left to factor (c3: 51.9 / 50.5 / 50.8 / 50.7 ms at 50 / 60 / 65 / 70 %, 53.4 released at once; c2 7.89 / 7.84 / 8.29 at 50 / 65 /
  // 80 %); beyond -- the factorisation's own two-level regime -- they are released at once (c5: 297.5 ms against 301.6 at 65 %,
  // 308.4 for two pipelines).  (With the tile-by-tile chain of rounds 1-4 the steps of factors of <= 12 tile rows went on
  // the panel stream itself; with the resident chain the second stream wins there too: N_tot = 1 152: 1.36 against 1.50 ms.)
  if (T <= ctx->ride_same_stream_max_tiles) { rd.stream = ctx->s_main; rd.stream2 = nullptr; }
  rd.gate_pct = ctx->ride_gate_pct >= 0 ? ctx->ride_gate_pct : (T <= 192 ? 65 : 100);
  rd.open = rd.gate_pct >= 100 || rd.stream == ctx->s_main;
  // two-level form: outer blocks of 2 048 rows from 64 tile rows on (c3 51.9 -> 51.4 ms, c5 301.8 -> 298.1; blocks of 4 096: c5 301.0)
  rd.outer_t = (ctx->ride_outer_rows >= 8 * TILE && T >= ctx->ride_outer_min_tiles && T >= 2 * (ctx->ride_outer_rows / TILE)) ? (int)(ctx->ride_outer_rows / TILE) : 0;
  // everything enqueued on the panel stream so far (cross-covariance assembly, residual column) precedes the first step
  LPGP_HIP(hipEventRecord(ctx->ev_ride[2], ctx->s_main));
  if (rd.stream != ctx->s_main) LPGP_HIP(hipStreamWaitEvent(rd.stream, ctx->ev_ride[2], 0));
  if (rd.stream2 && rd.stream2 != ctx->s_main) LPGP_HIP(hipStreamWaitEvent(rd.stream2, ctx->ev_ride[2], 0));
  if (t_done >= T) {
    // nothing left to factor: the plain substitution
    return trsm_lower_blocked(ctx, mat, T, v, ldv, m_pad);
  }
  if (ctx->ride_aug && (T + rd.mtl) * (int64_t)TILE <= mat->cap && !ctx->single_stream) {
    // AUGMENTED form (round 6, option ride_aug): V^T = K_xX L^{-T} are ROWS of the matrix being factored -- the Cholesky of
    // [G; K_xX] without its last diagonal block (BlockMatrix2x2.L_A_inv_B, linops/_block.py:203-207, read row-wise).  The
    // substitution's updates then are tiles of the factorisation's own trailing-update launches (ONE grid, no second claimant
    // of the chip) and its panel steps rows of the panel chain's tile solves.  K_Xx is transposed into the free rows below the
    // matrix's blocks and back: 2 x 8 N M bytes through HBM.
    double* vt = mat->a + T * (int64_t)TILE;
    LPGP_TRY(transpose(ctx->s_main, vt, mat->cap, v, ldv, T * (int64_t)TILE, m_pad));
    LPGP_TRY(potrf_blocked_impl(ctx, mat, t_done, T, nullptr, nullptr, (int)T + rd.mtl));
    return transpose(ctx->s_main, v, ldv, vt, mat->cap, m_pad, T * (int64_t)TILE);
  }
  if (T >= ctx->ride_max_tiles) {
    // very large factors: the two pipelines back to back (no host synchronisation in between).  At c4 (520 tile rows, 129 tile
    // columns of right-hand side) the factorisation's outer updates with K = 2 048 and the substitution's with K = 4 096 have
    // the chip to themselves for seconds and the chain-bound parts are a per-cent effect; riding inside costs 1.8 % there
    // (2 718 against 2 670 ms) where it gains 6 % at c3, 13 % at c2 and 3 % at c5.
    LPGP_TRY(potrf_blocked_impl(ctx, mat, t_done, T, nullptr, nullptr));
    return trsm_lower_blocked(ctx, mat, T, v, ldv, m_pad);
  }
  LPGP_TRY(potrf_blocked_impl(ctx, mat, t_done, T, nullptr, &rd));
  if (rd.stream != ctx->s_main) {
    LPGP_HIP(hipEventRecord(ctx->ev_ride[2], rd.stream));
    LPGP_HIP(hipStreamWaitEvent(ctx->s_main, ctx->ev_ride[2], 0));
  }
  if (rd.stream2 && rd.stream2 != ctx->s_main && rd.stream2 != rd.stream) {
    LPGP_HIP(hipEventRecord(ctx->ev_ride[3], rd.stream2));
    LPGP_HIP(hipStreamWaitEvent(ctx->s_main, ctx->ev_ride[3], 0));
  }
  return 0;
}

static int potrf_blocked_impl(lpgp_ctx* ctx, lpgp_mat* mat, int64_t t_done64, int64_t T64, int32_t* info, Ride* ride, int TR) {
  const int T = (int)T64, t_done = (int)t_done64;
  if (TR < T) TR = T;                                 // (TR > T: rows beyond the columns being factored, see factor_columns)
  const int64_t ld = mat->cap;
  double* a = mat->a;
  const int nbt = (int)(ctx->nb / TILE);
  const int64_t tb = TILE;
  hipStream_t sP = ctx->s_main, sU = ctx->s_upd;
  // info != nullptr: the status is read back at the end (one host synchronisation); nullptr (lpgp_potrf_enqueue): it
  // accumulates in the matrix's own sticky word and is read by lpgp_mat_check
  ctx->d_info_cur = info ? ctx->d_info : mat->d_status;
  if (info) LPGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), sP));

  // ---- phase A (append): push the new rows through the already factored columns ----
  hipEvent_t append_dep = nullptr;
  if (t_done > 0 && T > t_done) {
    const int mnew = TR - t_done;
    double* rows = a + (int64_t)t_done * tb;          // row offset of the new rows
    for (int p0 = 0; p0 < t_done; p0 += nbt) {
      const int p1 = (p0 + nbt < t_done) ? p0 + nbt : t_done;
      if (ctx->fused_solve && p1 - p0 <= 4) {
        // the old panel's diagonal block is final: the chain of the new rows through it in one launch (panel_solve_kernel)
        LPGP_TRY(launch_trsm_panel(ctx, sP, rows + (int64_t)p0 * tb * ld, ld, mat->linv + (int64_t)p0 * tb * tb,
                                   a + (int64_t)p0 * tb * (ld + 1), ld, p1 - p0, mnew, LPGP_K_PANEL));
      } else
      for (int jt = p0; jt < p1; ++jt) {
        double* X = rows + (int64_t)jt * tb * ld;
        LPGP_TRY(panel_trsm(ctx, sP, mat, jt, X, mnew));
        if (jt + 1 < p1)
          LPGP_TRY(launch_gemm(ctx, sP, 0, 0,
                               mk(X, ld, a + (int64_t)(jt + 1) * tb + (int64_t)jt * tb * ld, ld,
                                  rows + (int64_t)(jt + 1) * tb * ld, ld, mnew, p1 - jt - 1, TILE, -1.0, 1.0, 0),
                               LPGP_K_GEMM));
      }
      if (ride) LPGP_TRY(ride_panel(ctx, mat, T, p0, p1, ride, true));  // the old panel's columns, new rows included, are final
      const int K = (p1 - p0) * TILE;
      double* Xp = rows + (int64_t)p0 * tb * ld;
      if (p1 < t_done)
        LPGP_TRY(launch_gemm(ctx, sP, 0, 0,
                             mk(Xp, ld, a + (int64_t)p1 * tb + (int64_t)p0 * tb * ld, ld,
                                rows + (int64_t)p1 * tb * ld, ld, mnew, t_done - p1, K, -1.0, 1.0, 0),
                             LPGP_K_GEMM));
      // Round 6: the update of the new block by the LAST old panel with the look-ahead split every other trailing update has -- the first
      // new panel's columns on the panel stream, its chain right behind them, the remainder on the update stream underneath that
      // chain (until then ONE launch on the panel stream: at c3 the first chain of the PDE block, 0.5 ms on an otherwise idle
      // chip, waited for all 2.4 ms of the 16 384^2 x 512 update).  The remainder's event is the `dep` of factor_columns.
      const bool split = ctx->lookahead != 0 && ctx->append_split && p1 == t_done && (T - t_done) > nbt && mnew >= ctx->append_split_min_tiles &&
                         sU != sP && ctx->ev_append[0] && ctx->ev_append[1];
      if (!split) {
        LPGP_TRY(launch_gemm(ctx, sP, 0, 0,
                             mk(Xp, ld, Xp, ld, rows + (int64_t)t_done * tb * ld, ld, mnew, T - t_done, K, -1.0, 1.0, 1),
                             LPGP_K_SYRK));
      } else {
        const int c1 = t_done + nbt;
        LPGP_HIP(hipEventRecord(ctx->ev_append[0], sP));                   // the new rows are solved against the old panel
        LPGP_TRY(launch_gemm(ctx, sP, 0, 0,
                             mk(Xp, ld, Xp, ld, rows + (int64_t)t_done * tb * ld, ld, mnew, nbt, K, -1.0, 1.0, 3),
                             LPGP_K_SYRK_AHEAD));
        LPGP_HIP(hipStreamWaitEvent(sU, ctx->ev_append[0], 0));
        const double* Xq = Xp + (int64_t)nbt * tb;                           // rows from the second new panel on
        LPGP_TRY(launch_gemm(ctx, sU, 0, 0,
                             mk(Xq, ld, Xq, ld, a + (int64_t)c1 * tb * (ld + 1), ld, TR - c1, T - c1, K, -1.0, 1.0, 1),
                             LPGP_K_SYRK));
        LPGP_HIP(hipEventRecord(ctx->ev_append[1], sU));
        append_dep = ctx->ev_append[1];
      }
    }
  }

  // ---- phase B: right-looking with look-ahead; on LARGE matrices the far columns are updated once
  //      per nb_outer columns ----
  // A trailing update costs more than its flops when K is short: every C tile is read and written
  // once per launch and while a workgroup loads C its CU has a single workgroup in the k-loop (27 % of
  // the time at K = 512, 11 % at 2048; scratch/timeline.hip): in steady state the same output runs at
  // 65.1 TFLOP/s with K = 512, 67.1 with K = 1024, 68.5 with K = 2048 (profiles/r03_clock_power.txt; the
  // 51-53 / 58 / 62 of rounds 1-2 were five-launch bursts inside the clock dip that follows the onset
  // of load, which exaggerates the difference).  Longer-K workgroups hold their CU slots longer, which slows every kernel of the
  // panel chain next to them; at c3 sizes that costs more than the update gains,
  // so the schedule below is used only while more than nb_outer_min_tiles tile columns remain.
  // Then nb_outer / nb consecutive panels form an OUTER panel: inside it the rank-nb updates touch
  // only the outer panel's own columns, and everything to the right is updated once per outer panel
  // with K = nb_outer, in three pieces:
  //   (a0) the first inner panel of the next outer panel, on the panel stream (critical path);
  //   (a1) the rest of the next outer panel, on the outer update stream: it runs underneath the
  //        first chain of the next outer panel and is awaited by that panel's first inner update;
  //   (b)  all columns beyond, behind (a1) on the same stream.
  const int NBt = (int)(ctx->nb_outer / TILE);
  const bool la = ctx->lookahead != 0;
  hipStream_t sO = ctx->s_outer ? ctx->s_outer : sU;
  hipEvent_t ev_a1 = append_dep, ev_b = nullptr;
  int oit = 0;
  for (int q0 = t_done; q0 < T; ++oit) {
    const bool outer = la && NBt > nbt && (T - q0) > ctx->nb_outer_min_tiles && (T - q0) > NBt;
    const int q1 = outer ? q0 + NBt : T;
    LPGP_TRY(factor_columns(ctx, mat, T, q0, q1, sU, ev_a1, ride, TR));
    if (q1 >= T) break;
    hipEvent_t ev_fact = ctx->ev_outer_fact[oit & 1];
    LPGP_HIP(hipEventRecord(ev_fact, sP));             // outer panel [q0, q1) is final (factor_columns joins its updates into sP)
    const int K = (q1 - q0) * TILE;
    const bool next_outer = (T - q1) > ctx->nb_outer_min_tiles && (T - q1) > NBt;
    const int q2 = next_outer ? q1 + NBt : T;
    const int qa = (q1 + nbt < q2) ? q1 + nbt : q2;    // end of the next outer panel's first inner panel
    // (a0) on the panel stream; its columns were last written by the previous (b)
    if (ev_b) LPGP_HIP(hipStreamWaitEvent(sP, ev_b, 0));
    const double* P = a + (int64_t)q1 * tb + (int64_t)q0 * tb * ld;
    LPGP_TRY(launch_gemm(ctx, sP, 0, 0,
                         mk(P, ld, P, ld, a + (int64_t)q1 * tb * (ld + 1), ld, TR - q1, qa - q1, K, -1.0, 1.0, 3),
                         LPGP_K_SYRK_AHEAD));
    LPGP_HIP(hipStreamWaitEvent(sO, ev_fact, 0));
    ev_a1 = nullptr;
    if (qa < q2) {                                     // (a1): columns [qa, q2), rows [qa, T)
      const double* Pa = a + (int64_t)qa * tb + (int64_t)q0 * tb * ld;
      LPGP_TRY(launch_gemm(ctx, sO, 0, 0,
                           mk(Pa, ld, Pa, ld, a + (int64_t)qa * tb * (ld + 1), ld, TR - qa, q2 - qa, K, -1.0, 1.0, 1),
                           LPGP_K_SYRK));
      ev_a1 = ctx->ev_outer_a1[oit & 1];
      LPGP_HIP(hipEventRecord(ev_a1, sO));
    }
    ev_b = nullptr;
    if (q2 < T) {                                      // (b): columns [q2, T)
      const double* Pb = a + (int64_t)q2 * tb + (int64_t)q0 * tb * ld;
      LPGP_TRY(launch_gemm(ctx, sO, 0, 0,
                           mk(Pb, ld, Pb, ld, a + (int64_t)q2 * tb * (ld + 1), ld, TR - q2, T - q2, K, -1.0, 1.0, 1),
                           LPGP_K_SYRK));
      ev_b = ctx->ev_outer[oit & 1];
      LPGP_HIP(hipEventRecord(ev_b, sO));
    }
    q0 = q1;
  }
  if (ev_a1) LPGP_HIP(hipStreamWaitEvent(sP, ev_a1, 0));
  if (ev_b) LPGP_HIP(hipStreamWaitEvent(sP, ev_b, 0));
  if (!info) return 0;
  // (into pinned memory: a copy into pageable memory is staged and blocks twice; this wait is once per conditioning of the default mode)
  int* const hp = ctx->h_info_pinned + 5;
  LPGP_HIP(hipMemcpyAsync(hp, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, sP));
  LPGP_HIP(hipStreamSynchronize(sP));
  const int h_info = *hp;
  if (h_info < 0) mat->poisoned = 1;
  LPGP_CHECK(h_info >= 0, "resident panel chain: a hand-over between workgroups timed out (device status %d); set LPGP_CHAIN_RESIDENT=-1", h_info);
  *info = h_info;
  return 0;
}

// column-major rows x cols block copy (rows even): pack / unpack of panels, small 2-D copies.
// (hipMemcpy2DAsync device-to-device ran at ~0.1 TB/s here: 1.1 ms per panel at N = 34k.)
__global__ __launch_bounds__(256) void copy2d_kernel(double* __restrict__ dst, int64_t ldd,
                                                      const double* __restrict__ src, int64_t lds, int64_t rows) {
  const int64_t c = blockIdx.y;
  const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (r < rows)
    *reinterpret_cast<double2*>(dst + c * ldd + r) = *reinterpret_cast<const double2*>(src + c * lds + r);
}
int copy2d(hipStream_t st, double* dst, int64_t ldd, const double* src, int64_t lds, int64_t rows, int64_t cols) {
  if (rows <= 0 || cols <= 0) return 0;
  for (int64_t c0 = 0; c0 < cols; c0 += 65535) {          // grid.y limit
    const int64_t nc = cols - c0 < 65535 ? cols - c0 : 65535;
    hipLaunchKernelGGL(copy2d_kernel, dim3((unsigned)((rows / 2 + 255) / 256), (unsigned)nc), dim3(256), 0, st, dst + c0 * ldd, ldd,
                       src + c0 * lds, lds, rows);
  }
  LPGP_HIP(hipGetLastError());
  return 0;
}

// Two-level forward substitution (round 3).  The right-looking solve below applies every panel of 512 columns to ALL rows
// beneath it: each update reads and writes the whole remaining right-hand side for 512 steps of contraction, and runs at
// the K = 512 rate of the GEMM (51-52 TFLOP/s in situ against 57 at K = 1024 and 61+ at K = 2048: every C tile is loaded
// and stored once per launch).  Here `NBt` tile rows (2048 rows) form an OUTER block: inside it the
// fused panel chains (panel_solve_kernel, 512 rows each) and their rank-512 updates touch the block's own rows only --
// short launches on the panel stream -- and everything below the block is updated ONCE per block with K = 2048, split
// like every update of this file into the next block's rows (a), on the panel stream, and the rest (b), on the update
// stream, underneath which the next block's inner work runs.  7/8 of the flops move to the long-K launches.  Unlike the
// factorisation -- where the same idea (`nb_outer`) loses at c3 sizes because workgroups of a long-K update hold their CU
// slots four times longer and starve the panel chain's many small kernels -- the substitution's chain is one fused launch
// per panel, and the inner work of a block has a whole outer update to hide under.
// Measured (bench.py, predict phase; LPGP_NB_OUTER_SOLVE = 0 / 1024 / 2048 / 4096): c4 (520 tile rows) 1113 / - / 1075 /
// 1063 ms; c5 (263) 80.2 / 78.8 / 80.9 / - ms; c3 (132) 22.9 / 22.7 / 23.6 / - ms; c2 (65) 2.80 / 2.69 / 2.87 / - ms: the
// outer updates do run at the long-K rate, but with few outer blocks the inner work of the first block and the tail are not
// hidden, and (a), (b) and the inner launches share one chip.  Used from 384 tile rows on, with blocks of 4096 rows.
static int trsm_lower_two_level(lpgp_ctx* ctx, lpgp_mat* mat, int T, double* v, int64_t ldv, int mtl, int nbt, int NBt) {
  const int64_t ld = mat->cap, tb = TILE;
  const double* a = mat->a;
  hipStream_t sP = ctx->s_main, sU = ctx->s_upd_all;
  auto upd = [&](hipStream_t st, int c0, int c1, int r0, int r1) -> int {        // rows [r0, r1) -= L[rows, c0:c1] V[c0:c1]
    if (r1 <= r0) return 0;
    GemmArgs g = mk(a + (int64_t)r0 * tb + (int64_t)c0 * tb * ld, ld, v + (int64_t)c0 * tb, ldv, v + (int64_t)r0 * tb, ldv, r1 - r0, mtl,
                    (c1 - c0) * TILE, -1.0, 1.0, 0);
    return launch_gemm(ctx, st, 0, 1, g, LPGP_K_GEMM);
  };
  bool have_upd_event = false;
  int it = 0;
  struct PadGuard { lpgp_ctx* c; ~PadGuard() { c->panel_lds_extra = 0; } } pad_guard{ctx};
  ctx->panel_lds_extra = ctx->panel_exclusive ? 20480 : 0;       // (solve_panel.h: the inner panel chains do not slip into the long outer updates)
  for (int q0 = 0; q0 < T; q0 += NBt, ++it) {
    const int q1 = (q0 + NBt < T) ? q0 + NBt : T;
    // inner: the block's own rows, right-looking by fused panels
    for (int p0 = q0; p0 < q1; p0 += nbt) {
      const int p1 = (p0 + nbt < q1) ? p0 + nbt : q1;
      LPGP_TRY(launch_trsv_panel(ctx, sP, v + (int64_t)p0 * tb, ldv, mat->linv + (int64_t)p0 * tb * tb, a + (int64_t)p0 * tb * (ld + 1), ld,
                                 p1 - p0, mtl, LPGP_K_PANEL));
      LPGP_TRY(upd(sP, p0, p1, p1, q1));
    }
    if (q1 >= T) break;
    const int q2 = (q1 + NBt < T) ? q1 + NBt : T;
    // outer update with K = (q1 - q0) * 128.  (b) is released with the block while it bounds the pipeline -- (a), too small
    // to fill the chip alone, then runs underneath it -- and after (a) once the next block's inner work would be the bound
    const double K = (double)(q1 - q0) * TILE;
    const double t_b_us = (double)(T - q2) * mtl * (2.0 * TILE * TILE * K / 55e6);
    const double t_inner_us = ctx->solve_chain_us_tile * (double)(q2 - q1) + ctx->chain_us_fixed * (double)((q2 - q1 + nbt - 1) / nbt) +
                              0.5 * (double)(q2 - q1) * mtl * (2.0 * TILE * TILE * (double)(nbt * TILE) / 45e6);
    const bool chain_bound = t_b_us < t_inner_us;
    hipEvent_t evp = ctx->ev_panel[it & 1];
    if (!chain_bound) LPGP_HIP(hipEventRecord(evp, sP));
    if (have_upd_event) LPGP_HIP(hipStreamWaitEvent(sP, ctx->ev_upd[(it + 1) & 1], 0));      // (a)'s rows: last written by the previous (b)
    LPGP_TRY(upd(sP, q0, q1, q1, q2));
    if (chain_bound) LPGP_HIP(hipEventRecord(evp, sP));
    if (q2 < T) {
      LPGP_HIP(hipStreamWaitEvent(sU, evp, 0));
      LPGP_TRY(upd(sU, q0, q1, q2, T));
      LPGP_HIP(hipEventRecord(ctx->ev_upd[it & 1], sU));
      have_upd_event = true;
    } else {
      have_upd_event = false;
    }
  }
  LPGP_HIP(hipEventRecord(ctx->ev_upd[0], sU));
  LPGP_HIP(hipStreamWaitEvent(sP, ctx->ev_upd[0], 0));
  return 0;
}

// V <- L^{-1} V for a padded (T*128) x m_pad block, column-major with leading dim ldv.
// Right-looking by panels with the same look-ahead as the factorisation: the latency-bound
// tile steps of panel p+1 (products with tile inverses, K = 128) run on the panel stream while
// the update stream still applies panel p to the rows below.
int trsm_lower_blocked(lpgp_ctx* ctx, lpgp_mat* mat, int64_t T64, double* v, int64_t ldv, int64_t m_pad) {
  const int T = (int)T64;
  const int64_t ld = mat->cap, tb = TILE;
  const int mtl = (int)(m_pad / TILE);
  // panel width of the substitution: its chain has no tile Cholesky, so wider panels (longer K in the
  // updates) pay earlier than in the factorisation.  Measured (nb_solve = 512 / 768 / 1024 / 2048):
  // c2 (65 tile rows) 9.57 / - / 9.74 / - ms, c3 (132) 55.5 / 55.1 / 56.1 / - ms, c4 (520) 2702 / - / 2671 /
  // 2655 ms.
  // Panels of at most 512 rows run their whole chain -- the tile solves and the in-panel updates between them -- in ONE
  // launch (panel_solve_kernel, gemm.hip: a workgroup owns 16 or 32 columns through the panel) up to 192 tile rows; beyond,
  // the longer K of wider panels in the updates is worth more than the shorter chain (c5, 263 tile rows: 311.9 ms with
  // panels of 768 and per-tile launches, 313.6 with fused panels of 512; scratch/fused_ab.sh)
  const int64_t nb_auto = T >= 384 ? 2048 : ((ctx->fused_solve && T <= 192) ? 512 : (T >= 96 ? 768 : ctx->nb));
  const int nbt = (int)((ctx->nb_solve > 0 ? ctx->nb_solve : nb_auto) / TILE);
  const bool fused = ctx->fused_solve && nbt <= 4;
  const double* a = mat->a;
  const bool la = ctx->lookahead != 0 && mtl >= 8;       // worth it only for wide right-hand sides
  {
    const int NBt = (int)(ctx->nb_outer_solve / TILE), nbi = (int)(ctx->nb / TILE);
    if (la && ctx->fused_solve && ctx->nb_solve == 0 && nbi <= 4 && NBt > nbi && NBt % nbi == 0 && T >= ctx->nb_outer_solve_min_tiles && T >= 2 * NBt)
      return trsm_lower_two_level(ctx, mat, T, v, ldv, mtl, nbi, NBt);
  }
  hipStream_t sP = ctx->s_main, sU = la ? ctx->s_upd_all : ctx->s_main;
  bool have_upd_event = false;
  int it = 0;
  // Round 4: while the remainder update is long, the look-ahead update rides IN FRONT of the next panel's fused chain
  // (panel_solve_kernel<NT, 1, true, 4>): ONE launch on the panel stream, which becomes runnable at the moment the previous
  // remainder update ends -- together with the next one -- and is resident before that one has filled the chip.  (As a launch
  // of its own the look-ahead update took those slots and the panel chain behind it starved until the remainder update
  // drained: the update stream idled 57-95 us per panel.  Kernel traces: profiles/r04_predict_fused_ahead.txt.)  The fused
  // kernel shares its CUs with the update for most of the update's duration, so it pays only while the update is long.
  static const int pad_blocked = [] { const char* e = std::getenv("LPGP_PANEL_LDS_EXTRA_BLOCKED"); return e ? std::atoi(e) : 0; }();   // (measurement aid)
  struct PadGuard { lpgp_ctx* c; ~PadGuard() { c->panel_lds_extra = 0; } } pad_guard{ctx};
  ctx->panel_lds_extra = pad_blocked;
  const bool ahead_ok = fused && la && nbt == 4 && ctx->fused_ahead != 0;
  bool solved = false;                      // the panel at the top of the loop has been solved by the previous iteration's fused launch
  for (int p0 = 0; p0 < T; p0 += nbt, ++it) {
    const int p1 = (p0 + nbt < T) ? p0 + nbt : T;
    if (solved)
      solved = false;
    else if (fused)
      LPGP_TRY(launch_trsv_panel(ctx, sP, v + (int64_t)p0 * tb, ldv, mat->linv + (int64_t)p0 * tb * tb, a + (int64_t)p0 * tb * (ld + 1), ld,
                                 p1 - p0, mtl, LPGP_K_PANEL));
    else
    for (int jt = p0; jt < p1; ++jt) {
      double* Vj = v + (int64_t)jt * tb;
      LPGP_TRY(launch_trsv_tile(ctx, sP, Vj, ldv, mat->linv + (int64_t)jt * tb * tb, a + (int64_t)jt * tb * (ld + 1), ld, mtl,
                                LPGP_K_TRSM));
      if (jt + 1 < p1)
        LPGP_TRY(launch_gemm(ctx, sP, 0, 1,
                             mk(a + (int64_t)(jt + 1) * tb + (int64_t)jt * tb * ld, ld, Vj, ldv,
                                v + (int64_t)(jt + 1) * tb, ldv, p1 - jt - 1, mtl, TILE, -1.0, 1.0, 0),
                             LPGP_K_GEMM));
    }
    if (p1 >= T) break;
    const int K = (p1 - p0) * TILE;
    const double* Vp = v + (int64_t)p0 * tb;
    if (!la) {
      LPGP_TRY(launch_gemm(ctx, sP, 0, 1,
                           mk(a + (int64_t)p1 * tb + (int64_t)p0 * tb * ld, ld, Vp, ldv, v + (int64_t)p1 * tb, ldv,
                              T - p1, mtl, K, -1.0, 1.0, 0),
                           LPGP_K_GEMM));
      continue;
    }
    const int p2 = (p1 + nbt < T) ? p1 + nbt : T;
    // (b) is released after (a) only while the panel chain bounds the pipeline (see potrf_blocked);
    // while the update does, (a) -- here only (p2-p1) x mtl tiles, too few to fill the chip alone
    // (measured: 62 us at 35 TFLOP/s per panel) -- runs underneath (b)
    const double t_b_us = (double)(T - p2) * mtl * (2.0 * TILE * TILE * (double)K / 50e6);
    const bool chain_bound = t_b_us < ctx->solve_chain_us_tile * (double)(p2 - p1) + ctx->chain_us_fixed;
    hipEvent_t evp = ctx->ev_panel[it & 1];
    if (ahead_ok && p1 - p0 == 4 && t_b_us >= (double)ctx->fused_ahead_min_us) {
      LPGP_HIP(hipEventRecord(evp, sP));                                                   // panel [p0, p1) is solved
      if (have_upd_event) LPGP_HIP(hipStreamWaitEvent(sP, ctx->ev_upd[(it + 1) & 1], 0));     // rows [p1, p2): last written by the previous remainder update
      LPGP_TRY(launch_trsv_panel_ahead(ctx, sP, v + (int64_t)p1 * tb, ldv, mat->linv + (int64_t)p1 * tb * tb, a + (int64_t)p1 * tb * (ld + 1),
                                       a + (int64_t)p1 * tb + (int64_t)p0 * tb * ld, ld, p2 - p1, mtl, LPGP_K_PANEL));
      solved = true;
      if (p2 < T) {
        LPGP_HIP(hipStreamWaitEvent(sU, evp, 0));
        GemmArgs gb = mk(a + (int64_t)p2 * tb + (int64_t)p0 * tb * ld, ld, Vp, ldv, v + (int64_t)p2 * tb, ldv, T - p2, mtl, K, -1.0, 1.0, 0);
        gb.occ3 = t_b_us > ctx->gemm3_margin * (ctx->solve_chain_us_tile * (double)(p2 - p1) + ctx->chain_us_fixed);
        LPGP_TRY(launch_gemm(ctx, sU, 0, 1, gb, LPGP_K_GEMM));
        LPGP_HIP(hipEventRecord(ctx->ev_upd[it & 1], sU));
        have_upd_event = true;
      } else {
        have_upd_event = false;
      }
      continue;
    }
    if (!chain_bound) LPGP_HIP(hipEventRecord(evp, sP));
    if (have_upd_event) LPGP_HIP(hipStreamWaitEvent(sP, ctx->ev_upd[(it + 1) & 1], 0));
    LPGP_TRY(launch_gemm(ctx, sP, 0, 1,
                         mk(a + (int64_t)p1 * tb + (int64_t)p0 * tb * ld, ld, Vp, ldv, v + (int64_t)p1 * tb, ldv,
                            p2 - p1, mtl, K, -1.0, 1.0, 0),
                         LPGP_K_GEMM));
    if (chain_bound) LPGP_HIP(hipEventRecord(evp, sP));
    if (p2 < T) {
      LPGP_HIP(hipStreamWaitEvent(sU, evp, 0));
      GemmArgs gb = mk(a + (int64_t)p2 * tb + (int64_t)p0 * tb * ld, ld, Vp, ldv, v + (int64_t)p2 * tb, ldv, T - p2, mtl, K, -1.0, 1.0, 0);
      gb.occ3 = t_b_us > ctx->gemm3_margin * (ctx->solve_chain_us_tile * (double)(p2 - p1) + ctx->chain_us_fixed);
      LPGP_TRY(launch_gemm(ctx, sU, 0, 1, gb, LPGP_K_GEMM));
      LPGP_HIP(hipEventRecord(ctx->ev_upd[it & 1], sU));
      have_upd_event = true;
    } else {
      have_upd_event = false;
    }
  }
  if (la) {
    LPGP_HIP(hipEventRecord(ctx->ev_upd[0], sU));
    LPGP_HIP(hipStreamWaitEvent(sP, ctx->ev_upd[0], 0));
  }
  return 0;
}

// V <- L^{-T} V (backward substitution), same layout.
int trsm_lower_t_blocked(lpgp_ctx* ctx, lpgp_mat* mat, int64_t T64, double* v, int64_t ldv, int64_t m_pad) {
  const int T = (int)T64;
  const int64_t ld = mat->cap, tb = TILE;
  const int mtl = (int)(m_pad / TILE);
  const int nbt = (int)(ctx->nb / TILE);
  const double* a = mat->a;
  hipStream_t st = ctx->s_main;
  void* sp = nullptr;
  const size_t sbytes = (size_t)TILE * (size_t)m_pad * sizeof(double);
  if (pool_alloc(ctx, &sp, sbytes, nullptr) != 0) return -1;
  double* S = (double*)sp;
  struct Release { lpgp_ctx* c; void* p; size_t b; ~Release() { pool_free(c, p, b); } } release{ctx, sp, sbytes};   // reuse is stream-ordered
  for (int p1 = T; p1 > 0;) {
    const int p0 = (p1 - nbt > 0) ? p1 - nbt : 0;
    for (int jt = p1 - 1; jt >= p0; --jt) {
      double* Vj = v + (int64_t)jt * tb;
      // x_jt = L_jj^{-T} y_jt with one refinement step (see tile_solve_kernel): S = Linv^T y;  y <- y - L^T S;
      // S <- S + Linv^T y;  y <- S.  (Not a hot path: `gram.solve(B)` / lpgp_potrs only.)
      const double* linv = mat->linv + (int64_t)jt * tb * tb;
      const double* Ljj = a + (int64_t)jt * tb * (ld + 1);
      LPGP_TRY(launch_gemm(ctx, st, 1, 1, mk(linv, tb, Vj, ldv, S, tb, 1, mtl, TILE, 1.0, 0.0, 0), LPGP_K_TRSM));
      LPGP_TRY(launch_gemm(ctx, st, 1, 1, mk(Ljj, ld, S, tb, Vj, ldv, 1, mtl, TILE, -1.0, 1.0, 0), LPGP_K_TRSM));
      LPGP_TRY(launch_gemm(ctx, st, 1, 1, mk(linv, tb, Vj, ldv, S, tb, 1, mtl, TILE, 1.0, 1.0, 0), LPGP_K_TRSM));
      LPGP_TRY(copy2d(st, Vj, ldv, S, tb, tb, m_pad));
      if (jt > p0)
        LPGP_TRY(launch_gemm(ctx, st, 1, 1,
                             mk(a + (int64_t)jt * tb + (int64_t)p0 * tb * ld, ld, Vj, ldv,
                                v + (int64_t)p0 * tb, ldv, jt - p0, mtl, TILE, -1.0, 1.0, 0),
                             LPGP_K_GEMM));
    }
    if (p0 > 0)
      LPGP_TRY(launch_gemm(ctx, st, 1, 1,
                           mk(a + (int64_t)p0 * tb, ld, v + (int64_t)p0 * tb, ldv, v, ldv, p0, mtl,
                              (p1 - p0) * TILE, -1.0, 1.0, 0),
                           LPGP_K_GEMM));
    p1 = p0;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------
// single right-hand side: representer weights  w = L^{-T} L^{-1} r   (_conditional.py:44,108)
// One fused launch per 128-row tile: every workgroup first forms the solution of the tile
// from the explicit tile inverse (128x128 matvec, L2 resident) with ONE step of iterative refinement
// against the diagonal tile of the factor (x0 = Linv b, x = x0 + Linv (b - L x0): a product with the
// explicit inverse alone has a backward error of cond(L_tile) eps, see tile_solve_kernel in gemm.hip), then
// applies it to its own slice of the remaining right-hand side.  x and b are distinct vectors, so no
// workgroup reads what another one writes inside a launch.
// ---------------------------------------------------------------------------------------
// 128x128 matvec, 512 threads: returns (M sin)[t] for t < 128 (M[r + c ldm]); part: 512 doubles of LDS
__device__ __forceinline__ double tile_mv(const double* __restrict__ M, int64_t ldm, const double* sin, double* part, int t) {
  const int r = t & 127, qd = t >> 7;
  double acc = 0.0;
#pragma unroll
  for (int c = 0; c < 32; ++c) acc = fma(M[(int64_t)(32 * qd + c) * ldm + r], sin[32 * qd + c], acc);
  part[t] = acc;
  __syncthreads();
  double v = 0.0;
  if (t < TILE) v = (part[t] + part[t + 128]) + (part[t + 256] + part[t + 384]);
  __syncthreads();
  return v;
}
// transposed 128x128 matvec, 8 waves: sout[c] = sum_r M[r + c ldm] sin[r]  (valid after the trailing barrier)
__device__ __forceinline__ void tile_mv_t(const double* __restrict__ M, int64_t ldm, const double* sin, double* sout, int t) {
  const int lane = t & 63, w = t >> 6;
  double acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const double* Mc = M + (int64_t)(w * 16 + j) * ldm;
    acc[j] = Mc[lane] * sin[lane] + Mc[lane + 64] * sin[lane + 64];
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    double a = acc[j];
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
    if (lane == 0) sout[w * 16 + j] = a;
  }
  __syncthreads();
}
// x = L^{-1} b (TRANS: L^{-T} b) for one tile, refined once; sb: right-hand side in LDS (128), s1 / s2: 128 doubles of
// LDS scratch each, part: 512.  On return s1 holds x (all threads may read it after the call).
template <bool TRANS>
__device__ __forceinline__ void tile_solve_vec(const double* __restrict__ linv, const double* __restrict__ Lt, int64_t ldl,
                                               const double* sb, double* s1, double* s2, double* part, int t) {
  if (!TRANS) {
    const double x0 = tile_mv(linv, TILE, sb, part, t);
    if (t < TILE) s1[t] = x0;
    __syncthreads();
    const double lx = tile_mv(Lt, ldl, s1, part, t);
    if (t < TILE) s2[t] = sb[t] - lx;
    __syncthreads();
    const double dx = tile_mv(linv, TILE, s2, part, t);
    if (t < TILE) s1[t] = x0 + dx;
    __syncthreads();
  } else {
    tile_mv_t(linv, TILE, sb, s1, t);                 // x0
    tile_mv_t(Lt, ldl, s1, s2, t);                    // L^T x0
    if (t < TILE) s2[t] = sb[t] - s2[t];
    __syncthreads();
    tile_mv_t(linv, TILE, s2, part, t);               // correction
    if (t < TILE) s1[t] += part[t];
    __syncthreads();
  }
}

// x_0 = L_00^{-1} b_0
__global__ __launch_bounds__(512) void trsv_fwd_head_kernel(const double* __restrict__ linv, const double* __restrict__ Lt, int64_t ld,
                                                             const double* __restrict__ b, double* __restrict__ x) {
  __shared__ double sb[TILE], s1[TILE], s2[TILE], part[512];
  const int t = threadIdx.x;
  if (t < TILE) sb[t] = b[t];
  __syncthreads();
  tile_solve_vec<false>(linv, Lt, ld, sb, s1, s2, part, t);
  if (t < TILE) x[t] = s1[t];
}

// step k: b[rows > tile k] -= L[rows, tile k] * x_k; the workgroup that owns tile k+1 then
// forms x_{k+1} = L_{k+1,k+1}^{-1} b_{k+1} (its right-hand side is final after this update).
__global__ __launch_bounds__(512) void trsv_fwd_step_kernel(const double* __restrict__ L, int64_t ld,
                                                             const double* __restrict__ linv_next,
                                                             double* __restrict__ b, double* __restrict__ x,
                                                             int64_t k0) {
  __shared__ double sy[TILE], sb[TILE], s1[TILE], s2[TILE], part[512];
  const int t = threadIdx.x, r = t & 127, qd = t >> 7;
  if (t < TILE) sy[t] = x[k0 + t];
  __syncthreads();
  const int64_t row = k0 + TILE + (int64_t)blockIdx.x * TILE + r;
  const double* Lp = L + row + (k0 + 32 * qd) * ld;
  double acc = 0.0;
#pragma unroll
  for (int c = 0; c < 32; ++c) acc = fma(Lp[(int64_t)c * ld], sy[32 * qd + c], acc);
  part[t] = acc;
  __syncthreads();
  if (t < TILE) {
    const double nb = b[row] - ((part[t] + part[t + 128]) + (part[t + 256] + part[t + 384]));
    b[row] = nb;
    sb[t] = nb;
  }
  if (blockIdx.x != 0) return;
  __syncthreads();
  tile_solve_vec<false>(linv_next, L + (k0 + TILE) * (ld + 1), ld, sb, s1, s2, part, t);
  if (t < TILE) x[k0 + TILE + t] = s1[t];
}

// x_last = L_last^{-T} y_last
__global__ __launch_bounds__(512) void trsv_bwd_head_kernel(const double* __restrict__ linv, const double* __restrict__ Lt, int64_t ld,
                                                             const double* __restrict__ y, double* __restrict__ x) {
  __shared__ double sy[TILE], s1[TILE], s2[TILE], part[TILE];
  const int t = threadIdx.x;
  if (t < TILE) sy[t] = y[t];
  __syncthreads();
  tile_solve_vec<true>(linv, Lt, ld, sy, s1, s2, part, t);
  if (t < TILE) x[t] = s1[t];
}

// step k (descending): y[c] -= L[tile k rows, c]^T x_k for 128 columns c per workgroup; the
// workgroup that owns tile k-1 (the last one) then forms x_{k-1} = L_{k-1,k-1}^{-T} y_{k-1}.
__global__ __launch_bounds__(512) void trsv_bwd_step_kernel(const double* __restrict__ L, int64_t ld,
                                                             const double* __restrict__ linv_prev,
                                                             double* __restrict__ y, double* __restrict__ x, int64_t k0) {
  __shared__ double sx[TILE], sy[TILE], s1[TILE], s2[TILE], part[TILE];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t < TILE) sx[t] = x[k0 + t];
  __syncthreads();
  const int64_t c0 = (int64_t)blockIdx.x * TILE;      // this workgroup's 128 columns (all < k0)
  double acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const double* Lp = L + k0 + (c0 + w * 16 + j) * ld;
    acc[j] = Lp[lane] * sx[lane] + Lp[lane + 64] * sx[lane + 64];
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    double a = acc[j];
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
    if (lane == 0) {
      const double ny = y[c0 + w * 16 + j] - a;
      y[c0 + w * 16 + j] = ny;
      sy[w * 16 + j] = ny;
    }
  }
  if (c0 + TILE != k0) return;                         // not the owner of tile k-1
  __syncthreads();
  tile_solve_vec<true>(linv_prev, L + c0 * (ld + 1), ld, sy, s1, s2, part, t);
  if (t < TILE) x[c0 + t] = s1[t];
}

// x <- L^{-1} b on `st` (padded length T*128, device); b is consumed as the running right-hand side
int solve_vec_fwd(lpgp_ctx* ctx, hipStream_t st, lpgp_mat* mat, int64_t T64, double* b, double* x) {
  const int T = (int)T64;
  const int64_t ld = mat->cap;
  const double* a = mat->a;
  auto linv = [&](int k) { return (const double*)(mat->linv + (int64_t)k * TILE * TILE); };
  hipLaunchKernelGGL(trsv_fwd_head_kernel, dim3(1), dim3(512), 0, st, linv(0), a, ld, (const double*)b, x);
  for (int k = 0; k + 1 < T; ++k)
    hipLaunchKernelGGL(trsv_fwd_step_kernel, dim3(T - k - 1), dim3(512), 0, st, a, ld, linv(k + 1), b, x,
                       (int64_t)k * TILE);
  LPGP_HIP(hipGetLastError());
  return 0;
}

// v (padded length T*128, device) <- G^{-1} v, using scratch vector `tmp` of the same length
int solve_vec(lpgp_ctx* ctx, lpgp_mat* mat, int64_t T64, double* v, double* tmp, int* info) {
  if (ctx->trsv_resident) return solve_vec_resident(ctx, mat, T64, v, tmp, info);
  const int T = (int)T64;
  const int64_t ld = mat->cap;
  hipStream_t st = ctx->s_main;
  const double* a = mat->a;
  auto linv = [&](int k) { return (const double*)(mat->linv + (int64_t)k * TILE * TILE); };
  // forward: L tmp = v   (v is consumed as the running right-hand side)
  int rc = solve_vec_fwd(ctx, st, mat, T64, v, tmp);
  if (rc != 0) return rc;
  // backward: L^T v = tmp  (tmp is consumed as the running right-hand side)
  hipLaunchKernelGGL(trsv_bwd_head_kernel, dim3(1), dim3(512), 0, st, linv(T - 1), a + (int64_t)(T - 1) * TILE * (ld + 1), ld,
                     (const double*)(tmp + (int64_t)(T - 1) * TILE), v + (int64_t)(T - 1) * TILE);
  for (int k = T - 1; k >= 1; --k)
    hipLaunchKernelGGL(trsv_bwd_step_kernel, dim3(k), dim3(512), 0, st, a, ld, linv(k - 1), tmp, v, (int64_t)k * TILE);
  LPGP_HIP(hipGetLastError());
  return 0;
}

}  // namespace lpgp
