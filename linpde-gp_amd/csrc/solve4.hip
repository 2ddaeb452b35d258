// panel_solve_kernel<4, 1, *>: a translation unit of its own (compile time), see solve_panel.h
#include "solve_panel.h"

namespace lpgp {

int launch_panel_solve_4(lpgp_ctx* ctx, hipStream_t stream, const PanelSolveArgs& a, int64_t cols, bool kfast) {
  return kfast ? launch_panel_solve_nt<4, true>(ctx, stream, a, cols) : launch_panel_solve_nt<4, false>(ctx, stream, a, cols);
}

}  // namespace lpgp
