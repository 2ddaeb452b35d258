// HIP-free part of the internal declarations: error reporting, the lowered kernel descriptor and its
// host-side lowering.  Included by lpgp_internal.h (device build) and by the host-only sources
// (lower.cpp, hosttest/) that are also compiled with the host compiler + AddressSanitizer.
#pragma once

#include <cstdint>

#include "lpgp.h"

namespace lpgp {

void set_error(const char* fmt, ...);
const char* last_error();

#define LPGP_CHECK(cond, ...)                                                       \
  do {                                                                              \
    if (!(cond)) {                                                                  \
      ::lpgp::set_error(__VA_ARGS__);                                               \
      return -2;                                                                    \
    }                                                                               \
  } while (0)

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// ---- lowered kernel descriptor (device form) ------------------------------------------
// entry = sum_g scale_g * exp(-sum_d E_d(r_d)) * sum_c sgn^{parity_c} Poly_c(r_1..r_d),
// r_d = |a_d (x_d - x'_d)|, E = r (Matern) or r^2/2 (ExpQuad); Poly_c dense nested-Horner
// coefficient tensor.  Built on the host by lower_kdesc (lower.cpp).
constexpr int MAXCLS = 16;         // parity classes (2^d, d <= 4)
constexpr int MAXCOEF = 8192;      // coefficient doubles over all groups (only the used part is staged to the device)

struct DevGroup {
  double scale;
  double a[LPGP_MAXD];
  int32_t expkind[LPGP_MAXD];      // 1: exp(-r), 2: exp(-r^2/2)
  int32_t deg[LPGP_MAXD];          // polynomial degree per dim
  int32_t ncls;
  int32_t parity[MAXCLS];          // bit d set => factor sign(x_d - x'_d)
  int32_t coef_off[MAXCLS];        // offset into coef[]
  // isotropic Matern group (LPGP_MATERN_ISO): with u = a .* (x - x'), s = |u|,
  //   entry = scale * exp(-s) * [ Q0(s) + (w . u) Q1(s) + (u^T B u) Q2(s) ],
  // Q0, Q1, Q2 of degree deg[0] at coef_off[0..2] (ncls = 3; parity[0] = 0 so that the constant
  // coefficient of Q0 is the diagonal value, as for the product form)
  int32_t iso, has_lin, has_quad;
  double w[LPGP_MAXD];
  double B[LPGP_MAXD * LPGP_MAXD];
};

struct DevDesc {
  int32_t d;
  int32_t ngroups;
  DevGroup g[LPGP_MAXG];
  double coef[MAXCOEF];
};

int lower_kdesc(const lpgp_kdesc* kd, int ngroups, DevDesc* out);
// value of sum_g (kd[g])(x, x): only the constant coefficient of the all-even parity classes survives
double desc_diag(const DevDesc& desc);

}  // namespace lpgp
