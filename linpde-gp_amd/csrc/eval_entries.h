// Evaluation of one lowered kernel descriptor (`DevDesc`) on AE coordinate differences per call:
// the arithmetic core of assemble_kernel / matvec_kernel (assemble.hip).  Kept in a header so that
// the host-only AddressSanitizer build (`build.sh --host-asan`, csrc/hosttest/) runs the SAME code on
// the CPU against the lowering under test.  Under hipcc this is device code; the host compiler sees
// plain C++ (that build is test infrastructure: nothing in the product calls the host instantiation).
#pragma once

#include <cmath>
#include <cstring>

#include "lpgp_desc.h"
#include "exp_table.h"

#if defined(__HIPCC__)
#define LPGP_HD __device__ __forceinline__
#else
#define LPGP_HD inline
#endif

namespace lpgp {

constexpr int AE = 8;           // entries per thread per pass (1 row x 8 cols)

LPGP_HD unsigned lpgp_hi32(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (unsigned)__double2hiint(v);
#else
  unsigned long long b;
  std::memcpy(&b, &v, 8);
  return (unsigned)(b >> 32);
#endif
}

// v with its sign bit XORed by `s` (s = 0 or 0x80000000)
LPGP_HD double lpgp_xor_sign(double v, unsigned s) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __hiloint2double((int)(((unsigned)__double2hiint(v)) ^ s), __double2loint(v));
#else
  unsigned long long b;
  std::memcpy(&b, &v, 8);
  b ^= ((unsigned long long)s) << 32;
  std::memcpy(&v, &b, 8);
  return v;
#endif
}

// Where the exponential of a Matern dimension comes from.  A half-integer Matern factor decays like e^{-a |x - x'|}, and
//   e^{-a |x - x'|} = min( e^{-a (x - x0)} e^{+a (x' - x0)},  e^{+a (x - x0)} e^{-a (x' - x0)} )       (any origin x0)
// -- one of the two products is e^{-a |x - x'|} <= 1, the other its reciprocal -- so the per-ENTRY exponential of a tile of
// the block collapses into per-POINT exponentials (64 rows + 64 columns instead of 4096 entries) and two multiplies and a
// minimum per entry and dimension.  A provider with `enabled` hands out that minimum for (group, dimension, entry);
// `NoFactors` (the default) evaluates exp(-sum r_d) per entry as before.  Dimensions of squared-exponential kind keep their
// per-entry exponential (e^{-(u-u')^2/2} does not split without an exponential of the cross term), as do isotropic groups.
// Rounding: the argument a (x - x0) carries a relative error of one ulp, i.e. |a (x - x0)| eps absolute in the exponent, so the
// provider is only used while |a (x - x0)| stays below a small bound on the tile (assemble.hip: FACT_TMAX).
// The per-point factors E+ = e^{-a (x - x0)}, E- = e^{+a (x - x0)} with the argument carried in double-double: t = a (x - x0)
// = t_hi + t_lo exactly (TwoSum of the difference, FMA residual of the product), E+- = exp(-+t_hi) (1 -+ t_lo).  Without
// the low part the argument's rounding error, |t| eps, becomes a RELATIVE error of the entry -- also for near pairs x ~ x'
// far from the origin, where the direct evaluation exp(-a |x - x'|) is exact to an ulp; on Gram matrices of condition 1e9
// that moved the posterior by 3e-8 (tests/test_gpu_random.py, seed 107).  With it the entry carries the two exp roundings
// and one multiply, whatever the distance to the origin.
LPGP_HD void lpgp_exp_factors(double a, double x, double x0, double& ep, double& em, double& t_abs) {
  const double d_hi = x - x0;
  const double bb = d_hi - x;
  const double d_lo = (x - (d_hi - bb)) + (-x0 - bb);          // x - x0 = d_hi + d_lo exactly
  const double t_hi = a * d_hi;
  const double t_lo = fma(a, d_hi, -t_hi) + a * d_lo;
  const double e1 = exp(-t_hi), e2 = exp(t_hi);
  ep = fma(-t_lo, e1, e1);
  em = fma(t_lo, e2, e2);
  t_abs = fabs(t_hi);
}

// e^{-s} for s >= 0: the per-entry exponential of every assembly / matrix-free kernel.  The library exp costs 33 vector
// operations per entry inside these kernels (degree-11 Horner whose coefficients the compiler re-materialises per entry, range
// selects) out of ~71; this one costs 16 and one table read: s = r - k ln2/256 with |r| <= ln2/512, e^{-s} = 2^{k >> 8} T[k & 255]
// (1 + p(r)), T[j] = 2^{j/256} held as head + tail (exp_table.h), p a degree-4 polynomial of e^r - 1 (error 2.4e-18): the only
// rounding that matters is the final addition -- 0.51 ulp measured against quad precision over 2e7 arguments (the library's
// exp: 0.51 on the host, <= 1 documented on the device).  s is clamped at 800 (result 0); a NaN / infinite s reaches the entry
// through the polynomial factor that multiplies the exponential (its Horner chain starts from 0 * r), so it needs no branch
// here.  `tab`: 256 {head, tail} pairs -- LDS on the device (staged per workgroup), the constant itself on the host.
#if defined(__HIPCC__)
__device__ const double g_exp_table[2 * EXP_TAB_N] = {LPGP_EXP_TABLE_VALUES};
#else
static const double g_exp_table[2 * EXP_TAB_N] = {LPGP_EXP_TABLE_VALUES};
#endif

struct ExpTab {
  const double* t;
};

LPGP_HD double lpgp_exp_neg(double s, const ExpTab& tab) {
  const double t = -fmin(s, 800.0);
  const double kf = rint(t * EXP_N_OVER_LN2);
  double r = fma(kf, -EXP_LN2_N_HI, t);
  r = fma(kf, -EXP_LN2_N_LO, r);
  const int k = (int)kf;
  const double* tj = tab.t + 2 * (k & (EXP_TAB_N - 1));
  const double Th = tj[0], Tl = tj[1];
  double p = fma(r, EXP_C4, EXP_C3);
  p = fma(r, p, EXP_C2);
  p = fma(r, p, EXP_C1);
  p *= r;
  const double v = Th + fma(Th, p, Tl);
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ldexp(v, k >> EXP_TAB_BITS);
#else
  return std::ldexp(v, k >> EXP_TAB_BITS);
#endif
}

struct NoFactors {
  static constexpr bool enabled = false;
  LPGP_HD double pair(int, int, int) const { return 1.0; }
};

// Where the polynomial coefficients come from.  `MemCoef` reads the descriptor's table (scalar loads on the device).  Round 3
// tried a provider that keeps the whole table in ONE register pair spread over the lanes of the wave (a single coalesced load
// per kernel, two v_readlane per coefficient: no load-to-use latency inside the Horner loops): 2.81 against 2.89 TB/s on the
// 4096 x 16384 cross-covariance -- the scalar loads are not what bounds the kernel -- and it was removed again.
struct MemCoef {
  const double* base;
  LPGP_HD double operator()(int idx) const { return base[idx]; }
};

template <int D, int NE, class Fac, class Coef>
LPGP_HD void eval_entries(const DevDesc* __restrict__ desc,
                                             const double (&dx)[D][NE], double (&res)[NE], const Fac& fac, const Coef& coef,
                                             const ExpTab& tab) {
  constexpr int AE = NE;          // (entries per call: the name the body uses)
#pragma unroll
  for (int e = 0; e < AE; ++e) res[e] = 0.0;
  for (int g = 0; g < desc->ngroups; ++g) {
    const DevGroup& G = desc->g[g];
    if (G.iso) {
      // isotropic Matern: e^{-s} [Q0(s) + (w.u) Q1(s) + (u^T B u) Q2(s)],  u = a .* dx, s = |u|
      double s2[AE], lin[AE], quad[AE];
#pragma unroll
      for (int e = 0; e < AE; ++e) { s2[e] = 0.0; lin[e] = 0.0; quad[e] = 0.0; }
      double u[D][AE];
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const double a = G.a[j], wj = G.w[j];
#pragma unroll
        for (int e = 0; e < AE; ++e) {
          u[j][e] = a * dx[j][e];
          s2[e] = fma(u[j][e], u[j][e], s2[e]);
          lin[e] = fma(wj, u[j][e], lin[e]);
        }
      }
      if (G.has_quad) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
          double bu[AE];
#pragma unroll
          for (int e = 0; e < AE; ++e) bu[e] = 0.0;
#pragma unroll
          for (int j = 0; j < D; ++j) {
            const double bij = G.B[i * LPGP_MAXD + j];
#pragma unroll
            for (int e = 0; e < AE; ++e) bu[e] = fma(bij, u[j][e], bu[e]);
          }
#pragma unroll
          for (int e = 0; e < AE; ++e) quad[e] = fma(u[i][e], bu[e], quad[e]);
        }
      }
      const int q0 = G.coef_off[0], q1 = G.coef_off[1], q2 = G.coef_off[2];
      double sv[AE], v0[AE], v1[AE], v2[AE];
#pragma unroll
      for (int e = 0; e < AE; ++e) { sv[e] = sqrt(s2[e]); v0[e] = 0.0; v1[e] = 0.0; v2[e] = 0.0; }
      for (int k = G.deg[0]; k >= 0; --k) {
        const double c0 = coef(q0 + k), c1 = coef(q1 + k), c2 = coef(q2 + k);
#pragma unroll
        for (int e = 0; e < AE; ++e) {
          v0[e] = fma(v0[e], sv[e], c0);
          v1[e] = fma(v1[e], sv[e], c1);
          v2[e] = fma(v2[e], sv[e], c2);
        }
      }
#pragma unroll
      for (int e = 0; e < AE; ++e)
        res[e] = fma(G.scale * lpgp_exp_neg(sv[e], tab), fma(quad[e], v2[e], fma(lin[e], v1[e], v0[e])), res[e]);
      continue;
    }
    double r[D][AE];
    unsigned sg[D][AE];
    double expo[AE], ef[AE];
    bool per_entry_exp = !Fac::enabled;       // some dimension of the group still needs exp(-expo) per entry
#pragma unroll
    for (int e = 0; e < AE; ++e) { expo[e] = 0.0; ef[e] = 1.0; }
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const double a = G.a[j];
      const int kind = G.expkind[j];
#pragma unroll
      for (int e = 0; e < AE; ++e) {
        double v = a * dx[j][e];
        sg[j][e] = lpgp_hi32(v) & 0x80000000u;
        r[j][e] = fabs(v);
      }
      if (Fac::enabled && kind == 1) {
#pragma unroll
        for (int e = 0; e < AE; ++e) ef[e] *= fac.pair(g, j, e);
      } else {
        per_entry_exp = true;
#pragma unroll
        for (int e = 0; e < AE; ++e) expo[e] += (kind == 1) ? r[j][e] : 0.5 * r[j][e] * r[j][e];
      }
    }
    double tot[AE];
#pragma unroll
    for (int e = 0; e < AE; ++e) tot[e] = 0.0;
    const int n1 = (D > 1) ? G.deg[D > 1 ? 1 : 0] + 1 : 1;
    const int n2 = (D > 2) ? G.deg[D > 2 ? 2 : 0] + 1 : 1;
    const int n3 = (D > 3) ? G.deg[D > 3 ? 3 : 0] + 1 : 1;
    for (int c = 0; c < G.ncls; ++c) {
      const int cf = G.coef_off[c];
      const int par = G.parity[c];
      double acc0[AE];
#pragma unroll
      for (int e = 0; e < AE; ++e) acc0[e] = 0.0;
      for (int i0 = G.deg[0]; i0 >= 0; --i0) {
        if constexpr (D == 1) {
          const double cv = coef(cf + i0);
#pragma unroll
          for (int e = 0; e < AE; ++e) acc0[e] = fma(acc0[e], r[0][e], cv);
        } else {
          double acc1[AE];
#pragma unroll
          for (int e = 0; e < AE; ++e) acc1[e] = 0.0;
          for (int i1 = n1 - 1; i1 >= 0; --i1) {
            if constexpr (D == 2) {
              const double cv = coef(cf + i0 * n1 + i1);
#pragma unroll
              for (int e = 0; e < AE; ++e) acc1[e] = fma(acc1[e], r[1][e], cv);
            } else {
              double acc2[AE];
#pragma unroll
              for (int e = 0; e < AE; ++e) acc2[e] = 0.0;
              for (int i2 = n2 - 1; i2 >= 0; --i2) {
                if constexpr (D == 3) {
                  const double cv = coef(cf + (i0 * n1 + i1) * n2 + i2);
#pragma unroll
                  for (int e = 0; e < AE; ++e) acc2[e] = fma(acc2[e], r[2][e], cv);
                } else {
                  double acc3[AE];
#pragma unroll
                  for (int e = 0; e < AE; ++e) acc3[e] = 0.0;
                  for (int i3 = n3 - 1; i3 >= 0; --i3) {
                    const double cv = coef(cf + ((i0 * n1 + i1) * n2 + i2) * n3 + i3);
#pragma unroll
                    for (int e = 0; e < AE; ++e) acc3[e] = fma(acc3[e], r[D - 1][e], cv);
                  }
#pragma unroll
                  for (int e = 0; e < AE; ++e) acc2[e] = fma(acc2[e], r[2][e], acc3[e]);
                }
              }
#pragma unroll
              for (int e = 0; e < AE; ++e) acc1[e] = fma(acc1[e], r[1][e], acc2[e]);
            }
          }
#pragma unroll
          for (int e = 0; e < AE; ++e) acc0[e] = fma(acc0[e], r[0][e], acc1[e]);
        }
      }
      // sign of the parity class: prod_{d in class} sign(x_d - x'_d)
#pragma unroll
      for (int e = 0; e < AE; ++e) {
        unsigned s = 0;
#pragma unroll
        for (int j = 0; j < D; ++j) s ^= ((par >> j) & 1) ? sg[j][e] : 0u;
        double v = lpgp_xor_sign(acc0[e], s);
        tot[e] += v;
      }
    }
    if (per_entry_exp) {
#pragma unroll
      for (int e = 0; e < AE; ++e) res[e] = fma(G.scale * (ef[e] * lpgp_exp_neg(expo[e], tab)), tot[e], res[e]);
    } else {
#pragma unroll
      for (int e = 0; e < AE; ++e) res[e] = fma(G.scale * ef[e], tot[e], res[e]);
    }
  }
}

template <int D, int NE = AE, class Fac = NoFactors>
LPGP_HD void eval_entries(const DevDesc* __restrict__ desc, const double (&dx)[D][NE], double (&res)[NE], const ExpTab& tab,
                          const Fac& fac = Fac()) {
  eval_entries<D, NE, Fac, MemCoef>(desc, dx, res, fac, MemCoef{desc->coef}, tab);
}

}  // namespace lpgp
