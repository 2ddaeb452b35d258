// panel_solve_kernel<NT, 1, true, 4>: the fused panel chain of the forward substitution WITH the look-ahead update by the
// previous panel in front of it (solve_panel.h, round 4); a translation unit of its own (compile time)
#include "solve_panel.h"

namespace lpgp {

// V (nt_rows <= 4 tiles of rows x nt_cols * 128 columns) <- L_KK^{-1} (V - Lprev Vprev): Vprev = the 512 solved rows right above V,
// Lprev = the nt_rows x 4 tile block of the factor left of the panel's diagonal block L
int launch_trsv_panel_ahead(lpgp_ctx* ctx, hipStream_t stream, double* V, int64_t ldv, const double* linv, const double* L, const double* Lprev,
                            int64_t ldl, int nt_rows, int nt_cols, int prof_kernel) {
  if (nt_rows <= 0 || nt_cols <= 0) return 0;
  LPGP_CHECK(nt_rows <= 4, "fused panel solve: at most 4 tiles per panel (got %d)", nt_rows);
  PanelSolveArgs a;
  a.V = V; a.ldv = ldv; a.linv = linv; a.L = L; a.ldl = ldl; a.Lprev = Lprev;
  const int64_t cols = (int64_t)nt_cols * TILE;
  // algorithmic flops: the look-ahead update (2 x 512 x rows x cols) and the triangular solve of the panel (cols x rows^2)
  const double rows = (double)nt_rows * TILE;
  if (prof_kernel >= 0) prof_begin(ctx, stream, prof_kernel, (double)cols * rows * (rows + 2.0 * 4 * TILE), 0.0);
  int rc;
  switch (nt_rows) {
    case 1: rc = launch_panel_solve_nt<1, true, 4>(ctx, stream, a, cols); break;
    case 2: rc = launch_panel_solve_nt<2, true, 4>(ctx, stream, a, cols); break;
    case 3: rc = launch_panel_solve_nt<3, true, 4>(ctx, stream, a, cols); break;
    default: rc = launch_panel_solve_nt<4, true, 4>(ctx, stream, a, cols); break;
  }
  if (prof_kernel >= 0) prof_end(ctx, stream);
  return rc;
}

}  // namespace lpgp
