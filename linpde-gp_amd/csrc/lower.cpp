// Host-side lowering of a kernel descriptor (C ABI `lpgp_kdesc`, a term list
//   sum_t c_t prod_d d^{n0} d'^{n1} k_d )
// to the device form `DevDesc`: per parity class a dense polynomial in r_d = |a_d (x_d - x'_d)| with
// exact integer tables for the Matern derivative polynomials (_matern.py:613-639) and the Hermite
// polynomials (_expquad.py).  Pure C++ (no HIP): built into liblpgp.so by hipcc and, with
// -fsanitize=address, into the host-only test library of `build.sh --host-asan`.

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstddef>
#include <cstring>
#include <vector>

#include "lpgp_desc.h"

namespace lpgp {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

const char* last_error() { return g_err; }

// ---------------------------------------------------------------------------------------
// host: polynomial tables
// ---------------------------------------------------------------------------------------
static long double ifact(int n) {
  long double r = 1;
  for (int i = 2; i <= n; ++i) r *= i;
  return r;
}

// Integer numerators of P_n for Matern nu = p + 1/2 over the common denominator
// D_p = (2p)!/p!  (c_k D_p = (2p-k)!/((p-k)! k!) 2^k are integers; P_n = P'_{n-1} - P_{n-1}).
static void matern_poly(int p, int n, long double* out /* p+1 */) {
  long double cur[16], nxt[16];
  for (int k = 0; k <= p; ++k)
    cur[k] = ifact(2 * p - k) / (ifact(p - k) * ifact(k)) * std::pow(2.0L, k);
  for (int it = 0; it < n; ++it) {
    for (int k = 0; k <= p; ++k) {
      long double d = (k + 1 <= p) ? (k + 1) * cur[k + 1] : 0.0L;
      nxt[k] = d - cur[k];
    }
    for (int k = 0; k <= p; ++k) cur[k] = nxt[k];
  }
  long double D = ifact(2 * p) / ifact(p);
  // round to double exactly like float(Fraction(num, D)) and continue in long double
  for (int k = 0; k <= p; ++k) out[k] = (long double)(double)(cur[k] / D);
}

// Integer numerators of P_n over D_p = (2p)!/p! (exact in long double for p <= 6, n <= 12).
static long double matern_poly_num(int p, int n, long double* num /* p+1 */) {
  long double cur[16], nxt[16];
  for (int k = 0; k <= p; ++k)
    cur[k] = ifact(2 * p - k) / (ifact(p - k) * ifact(k)) * std::pow(2.0L, k);
  for (int it = 0; it < n; ++it) {
    for (int k = 0; k <= p; ++k) nxt[k] = ((k + 1 <= p) ? (k + 1) * cur[k + 1] : 0.0L) - cur[k];
    for (int k = 0; k <= p; ++k) cur[k] = nxt[k];
  }
  for (int k = 0; k <= p; ++k) num[k] = cur[k];
  return ifact(2 * p) / ifact(p);
}

// Isotropic Matern with at most one derivative per argument (diffops/_matern.py:17-86,138-203):
//   k = kappa(s), s = |u|, u = a .* (x - x');   d/dx_i k = (P_1/s) e^{-s} a_i u_i = -d/dx'_i k
//   d/dx_i d/dx'_j k = -[ a_i a_j u_i u_j (P_2 - P_1/s)/s^2 + a_i^2 delta_ij P_1/s ] e^{-s}
// summed over the term list into  e^{-s} [Q0(s) + (w.u) Q1(s) + (u^T B u) Q2(s)].
static int lower_iso_group(const lpgp_kdesc& K, int d, DevGroup& G, double* coef, int& coef_used) {
  const int p = K.p[0];
  LPGP_CHECK(p >= 0 && p <= 6, "lower_kdesc: Matern p=%d unsupported", p);
  long double a[LPGP_MAXD];
  for (int j = 0; j < d; ++j) {
    LPGP_CHECK(K.family[j] == LPGP_MATERN_ISO && K.p[j] == p,
               "lower_kdesc: an isotropic Matern spans all dimensions with one nu");
    LPGP_CHECK(K.lengthscale[j] > 0, "lower_kdesc: lengthscale must be positive");
    const double as = std::sqrt(2.0 * (p + 0.5)) / K.lengthscale[j];
    a[j] = as;
    G.a[j] = as;
    G.expkind[j] = 1;
    G.deg[j] = 0;
  }
  G.deg[0] = p;
  G.iso = 1;
  long double c00 = 0, tr = 0, w[LPGP_MAXD] = {0, 0, 0, 0}, B[LPGP_MAXD][LPGP_MAXD] = {};
  bool first = false, second = false;
  for (int t = 0; t < K.nterms; ++t) {
    const lpgp_term& T = K.terms[t];
    int i0 = -1, i1 = -1, o0 = 0, o1 = 0;
    for (int j = 0; j < d; ++j) {
      LPGP_CHECK(T.n0[j] >= 0 && T.n1[j] >= 0, "lower_kdesc: derivative order out of range");
      o0 += T.n0[j];
      o1 += T.n1[j];
      if (T.n0[j]) i0 = j;
      if (T.n1[j]) i1 = j;
    }
    LPGP_CHECK(o0 <= 1 && o1 <= 1,
               "lower_kdesc: the isotropic Matern has closed forms for identity and directional derivatives only");
    if (T.coef == 0.0) continue;
    if (!o0 && !o1) c00 += T.coef;
    else if (o0 && !o1) { w[i0] += T.coef * a[i0]; first = true; }
    else if (!o0 && o1) { w[i1] -= T.coef * a[i1]; first = true; }
    else {
      B[i0][i1] += T.coef * a[i0] * a[i1];
      if (i0 == i1) tr += T.coef * a[i0] * a[i0];
      second = true;
    }
  }
  LPGP_CHECK(!first || p >= 1, "lower_kdesc: Matern-1/2 is not differentiable");
  LPGP_CHECK(!second || p >= 2, "lower_kdesc: a multivariate Matern needs nu >= 5/2 for a derivative on both arguments");
  long double P0[16], P1[16], P2[16];
  const long double D = matern_poly_num(p, 0, P0);
  matern_poly_num(p, 1, P1);
  matern_poly_num(p, 2, P2);
  LPGP_CHECK(coef_used + 3 * (p + 1) <= MAXCOEF, "lower_kdesc: coefficient table overflow");
  double* Q0 = coef + coef_used;
  double* Q1 = Q0 + (p + 1);
  double* Q2 = Q1 + (p + 1);
  for (int k = 0; k <= p; ++k) Q0[k] = Q1[k] = Q2[k] = 0.0;
  // P_1 // s   (P_1(0) = 0 for p >= 1), rounded to double per coefficient like the reference's
  // RationalPolynomial -> np.double conversion
  long double P1s[16] = {0};
  if (p >= 1) for (int k = 0; k < p; ++k) P1s[k] = P1[k + 1];
  for (int k = 0; k <= p; ++k) {
    Q0[k] = (double)(c00 * (long double)(double)(P0[k] / D) - tr * (long double)(double)(P1s[k] / D));
    Q1[k] = (double)(P1s[k] / D);
  }
  if (p >= 2) {
    // -(P_2 - P_1 // s) // s^2   (its two lowest coefficients vanish for p >= 2)
    for (int k = 0; k + 2 <= p; ++k) Q2[k] = -(double)((P2[k + 2] - P1s[k + 2]) / D);
  }
  G.ncls = 3;
  G.parity[0] = 0; G.parity[1] = 1; G.parity[2] = 1;
  G.coef_off[0] = coef_used;
  G.coef_off[1] = coef_used + (p + 1);
  G.coef_off[2] = coef_used + 2 * (p + 1);
  coef_used += 3 * (p + 1);
  G.has_lin = first ? 1 : 0;
  G.has_quad = second ? 1 : 0;
  for (int i = 0; i < LPGP_MAXD; ++i) {
    G.w[i] = (double)w[i];
    // symmetric part only: u^T B u sees nothing else
    for (int j = 0; j < LPGP_MAXD; ++j) G.B[i * LPGP_MAXD + j] = (double)(0.5L * (B[i][j] + B[j][i]));
  }
  return 0;
}

// Probabilists' Hermite He_n, ascending coefficients, degree n.
static void hermite_poly(int n, long double* out /* n+1 */) {
  long double a[16] = {1}, b[16];
  int deg = 0;
  for (int it = 0; it < n; ++it) {
    for (int k = 0; k <= deg + 1; ++k) b[k] = 0;
    for (int k = 0; k <= deg; ++k) b[k + 1] += a[k];            // u * He
    for (int k = 1; k <= deg; ++k) b[k - 1] -= k * a[k];        // - He'
    ++deg;
    for (int k = 0; k <= deg; ++k) a[k] = b[k];
  }
  for (int k = 0; k <= n; ++k) out[k] = a[k];
}

int lower_kdesc(const lpgp_kdesc* kd, int ngroups, DevDesc* out) {
  LPGP_CHECK(kd != nullptr && ngroups >= 1 && ngroups <= LPGP_MAXG, "lower_kdesc: bad ngroups %d", ngroups);
  std::memset(out, 0, offsetof(DevDesc, coef));      // header and groups; the coefficient table (64 KB) is written where it is used
  const int d = kd[0].d;
  LPGP_CHECK(d >= 1 && d <= LPGP_MAXD, "lower_kdesc: d=%d out of range", d);
  out->d = d;
  out->ngroups = ngroups;
  int coef_used = 0;
  for (int g = 0; g < ngroups; ++g) {
    const lpgp_kdesc& K = kd[g];
    LPGP_CHECK(K.d == d, "lower_kdesc: group %d has d=%d != %d", g, K.d, d);
    LPGP_CHECK(K.nterms >= 1 && K.nterms <= LPGP_MAXT, "lower_kdesc: nterms=%d", K.nterms);
    DevGroup& G = out->g[g];
    G.scale = K.scale;
    if (K.family[0] == LPGP_MATERN_ISO) {
      int rc = lower_iso_group(K, d, G, out->coef, coef_used);
      if (rc != 0) return rc;
      continue;
    }
    long double a[LPGP_MAXD];
    for (int j = 0; j < d; ++j) {
      LPGP_CHECK(K.lengthscale[j] > 0, "lower_kdesc: lengthscale must be positive");
      if (K.family[j] == LPGP_MATERN_HALFINT) {
        LPGP_CHECK(K.p[j] >= 0 && K.p[j] <= 6, "lower_kdesc: Matern p=%d unsupported", K.p[j]);
        // probnum Matern._scale_factors = sqrt(2 nu) / lengthscale, in fp64 like the reference
        double as = std::sqrt(2.0 * (K.p[j] + 0.5)) / K.lengthscale[j];
        a[j] = as;
        G.expkind[j] = 1;
      } else if (K.family[j] == LPGP_EXPQUAD) {
        a[j] = 1.0 / K.lengthscale[j];
        G.expkind[j] = 2;
      } else {
        LPGP_CHECK(false, "lower_kdesc: unknown family %d", K.family[j]);
      }
      G.a[j] = (double)a[j];
    }
    // degrees
    for (int j = 0; j < d; ++j) {
      int deg = 0;
      for (int t = 0; t < K.nterms; ++t) {
        int n = K.terms[t].n0[j] + K.terms[t].n1[j];
        LPGP_CHECK(K.terms[t].n0[j] >= 0 && K.terms[t].n1[j] >= 0 && n <= 12,
                   "lower_kdesc: derivative order out of range");
        int dg = (K.family[j] == LPGP_MATERN_HALFINT) ? K.p[j] : n;
        if (dg > deg) deg = dg;
      }
      G.deg[j] = deg;
    }
    int tsize = 1;
    for (int j = 0; j < d; ++j) tsize *= (G.deg[j] + 1);
    // accumulate per parity class
    std::vector<std::vector<long double>> cls(1 << d);
    for (int t = 0; t < K.nterms; ++t) {
      const lpgp_term& T = K.terms[t];
      int parity = 0;
      long double pref = T.coef;
      long double q[LPGP_MAXD][16];
      int qdeg[LPGP_MAXD];
      for (int j = 0; j < d; ++j) {
        int n = T.n0[j] + T.n1[j];
        if (n & 1) parity |= (1 << j);
        pref *= std::pow(a[j], n);
        if (K.family[j] == LPGP_MATERN_HALFINT) {
          if (T.n1[j] & 1) pref = -pref;
          matern_poly(K.p[j], n, q[j]);
          qdeg[j] = K.p[j];
        } else {
          if (T.n0[j] & 1) pref = -pref;
          hermite_poly(n, q[j]);
          qdeg[j] = n;
        }
      }
      auto& C = cls[parity];
      if (C.empty()) C.assign(tsize, 0.0L);
      // tensor product of the per-dim polynomials
      int idx[LPGP_MAXD] = {0, 0, 0, 0};
      for (;;) {
        long double v = pref;
        int lin = 0;
        for (int j = 0; j < d; ++j) {
          v *= q[j][idx[j]];
          lin = lin * (G.deg[j] + 1) + idx[j];
        }
        C[lin] += v;
        int j = d - 1;
        while (j >= 0) {
          if (++idx[j] <= qdeg[j]) break;
          idx[j] = 0;
          --j;
        }
        if (j < 0) break;
      }
    }
    G.ncls = 0;
    for (int c = 0; c < (1 << d); ++c) {
      if (cls[c].empty()) continue;
      bool nz = false;
      for (long double v : cls[c]) nz |= (v != 0.0L);
      if (!nz) continue;
      LPGP_CHECK(coef_used + tsize <= MAXCOEF, "lower_kdesc: coefficient table overflow");
      G.parity[G.ncls] = c;
      G.coef_off[G.ncls] = coef_used;
      for (int i = 0; i < tsize; ++i) out->coef[coef_used + i] = (double)cls[c][i];
      coef_used += tsize;
      ++G.ncls;
    }
  }
  return 0;
}

double desc_diag(const DevDesc& desc) {
  double v = 0.0;
  for (int g = 0; g < desc.ngroups; ++g)
    for (int c = 0; c < desc.g[g].ncls; ++c)
      if (desc.g[g].parity[c] == 0) v += desc.g[g].scale * desc.coef[desc.g[g].coef_off[c]];
  return v;
}

}  // namespace lpgp
