// Host-only TEST library (never loaded by the product): the descriptor lowering of lower.cpp and the
// evaluation core of eval_entries.h compiled with the host compiler + AddressSanitizer
// (`build.sh --host-asan`), so that the non-trivial host C++ of the library -- exact polynomial tables,
// parity-class folding, the isotropic-group folding, descriptor validation -- runs under a sanitizer on
// the CPU box (GPU AddressSanitizer is not available).  tests/test_host_asan.py drives it
// over the descriptor zoo and compares with the oracle.
#include <cstdint>
#include <vector>

#include "../eval_entries.h"

using namespace lpgp;

template <int D>
static void eval_block(const DevDesc& desc, const double* X0, int64_t n0, const double* X1, int64_t n1, double* out) {
  for (int64_t i = 0; i < n0; ++i)
    for (int64_t j0 = 0; j0 < n1; j0 += AE) {
      double dx[D][AE], res[AE];
      for (int e = 0; e < AE; ++e) {
        const int64_t j = (j0 + e < n1) ? j0 + e : n1 - 1;
        for (int dd = 0; dd < D; ++dd) dx[dd][e] = X0[i * D + dd] - X1[j * D + dd];
      }
      eval_entries<D>(&desc, dx, res, ExpTab{g_exp_table});
      for (int e = 0; e < AE && j0 + e < n1; ++e) out[i * n1 + j0 + e] = res[e];
    }
}

// The factored evaluation (eval_entries.h: `Fac`; on the device the per-point factors sit in LDS, assemble.hip:
// LdsFactors): per-point exponentials e^{-+ a (x - x0)} of every Matern dimension relative to the first column point
// of the 8-column strip, the per-entry exponential as the minimum of the two products -- the SAME eval_entries code path.
template <int D>
struct HostFactors {
  static constexpr bool enabled = true;
  const DevDesc* desc;
  double xr[D], xc[D][AE], x0[D];
  double pair(int g, int j, int e) const {
    const double a = desc->g[g].a[j];
    double rp, rm, cp, cm, t;
    lpgp_exp_factors(a, xr[j], x0[j], rp, rm, t);
    lpgp_exp_factors(a, xc[j][e], x0[j], cp, cm, t);
    const double p1 = rp * cm, p2 = rm * cp;
    return p1 < p2 ? p1 : p2;
  }
};

template <int D>
static void eval_block_fact(const DevDesc& desc, const double* X0, int64_t n0, const double* X1, int64_t n1, double* out) {
  for (int64_t i = 0; i < n0; ++i)
    for (int64_t j0 = 0; j0 < n1; j0 += AE) {
      double dx[D][AE], res[AE];
      HostFactors<D> fac;
      fac.desc = &desc;
      for (int dd = 0; dd < D; ++dd) { fac.xr[dd] = X0[i * D + dd]; fac.x0[dd] = X1[j0 * D + dd]; }
      for (int e = 0; e < AE; ++e) {
        const int64_t j = (j0 + e < n1) ? j0 + e : n1 - 1;
        for (int dd = 0; dd < D; ++dd) { dx[dd][e] = X0[i * D + dd] - X1[j * D + dd]; fac.xc[dd][e] = X1[j * D + dd]; }
      }
      eval_entries<D, AE, HostFactors<D>>(&desc, dx, res, ExpTab{g_exp_table}, fac);
      for (int e = 0; e < AE && j0 + e < n1; ++e) out[i * n1 + j0 + e] = res[e];
    }
}

extern "C" {

const char* lpgp_host_last_error(void) { return last_error(); }

// out (n0 x n1, C order) = sum_g (kd[g])(X0, X1), X0 / X1 (n x d, C order)
int lpgp_host_kernel_matrix(const lpgp_kdesc* kd, int32_t ngroups, const double* X0, int64_t n0, const double* X1,
                            int64_t n1, double* out) {
  // the descriptor lives on the heap so that AddressSanitizer sees reads past its end
  std::vector<DevDesc> store(1);
  int rc = lower_kdesc(kd, ngroups, &store[0]);
  if (rc != 0) return rc;
  if (n0 <= 0 || n1 <= 0) return 0;
  switch (store[0].d) {
    case 1: eval_block<1>(store[0], X0, n0, X1, n1, out); break;
    case 2: eval_block<2>(store[0], X0, n0, X1, n1, out); break;
    case 3: eval_block<3>(store[0], X0, n0, X1, n1, out); break;
    case 4: eval_block<4>(store[0], X0, n0, X1, n1, out); break;
    default: return -2;
  }
  return 0;
}

// the same block through the factored evaluation path
int lpgp_host_kernel_matrix_fact(const lpgp_kdesc* kd, int32_t ngroups, const double* X0, int64_t n0, const double* X1,
                                 int64_t n1, double* out) {
  std::vector<DevDesc> store(1);
  int rc = lower_kdesc(kd, ngroups, &store[0]);
  if (rc != 0) return rc;
  if (n0 <= 0 || n1 <= 0) return 0;
  switch (store[0].d) {
    case 1: eval_block_fact<1>(store[0], X0, n0, X1, n1, out); break;
    case 2: eval_block_fact<2>(store[0], X0, n0, X1, n1, out); break;
    case 3: eval_block_fact<3>(store[0], X0, n0, X1, n1, out); break;
    case 4: eval_block_fact<4>(store[0], X0, n0, X1, n1, out); break;
    default: return -2;
  }
  return 0;
}

int lpgp_host_kernel_diag(const lpgp_kdesc* kd, int32_t ngroups, double* out_value) {
  std::vector<DevDesc> store(1);
  int rc = lower_kdesc(kd, ngroups, &store[0]);
  if (rc != 0) return rc;
  *out_value = desc_diag(store[0]);
  return 0;
}

// out[i] = lpgp_exp_neg(s[i]): the per-entry exponential of the assembly kernels (eval_entries.h), host instantiation
void lpgp_host_exp_neg(const double* s, int64_t n, double* out) {
  const ExpTab tab{g_exp_table};
  for (int64_t i = 0; i < n; ++i) out[i] = lpgp_exp_neg(s[i], tab);
}

}  // extern "C"
