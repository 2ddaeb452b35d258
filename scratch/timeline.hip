// Per-workgroup life cycle of one GEMM launch: when each workgroup started, began / ended its
// k-loop and issued its stores, and on which CU -- to see how much matrix-pipe time is lost between
// consecutive workgroups of a CU (diagnostic build of gemm.hip with -DLPGP_STAMP).
#define LPGP_STAMP 1
#include "../linpde-gp_amd/csrc/gemm.hip"
#include <vector>
#include <algorithm>
#include <map>
namespace lpgp { void set_error(const char* fmt, ...) {} void prof_begin(lpgp_ctx*, hipStream_t, int, double, double) {} void prof_end(lpgp_ctx*, hipStream_t) {} }
int main(int argc, char** argv) {
  using namespace lpgp;
  const int mt = argc > 1 ? atoi(argv[1]) : 64, nt = argc > 2 ? atoi(argv[2]) : 64, k = argc > 3 ? atoi(argv[3]) : 512, tri = argc > 4 ? atoi(argv[4]) : 0; const double beta = argc > 5 ? atof(argv[5]) : 1.0;
  const int64_t m = (int64_t)mt * 128, n = (int64_t)nt * 128;
  double *A, *B, *C;
  hipMalloc(&A, (size_t)m * k * 8); hipMalloc(&B, (size_t)n * k * 8); hipMalloc(&C, (size_t)m * n * 8);
  std::vector<double> hr((size_t)std::max(m, n) * k);
  unsigned long long x = 88172645463325252ull;
  for (auto& v : hr) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5; }
  hipMemcpy(A, hr.data(), (size_t)m * k * 8, hipMemcpyHostToDevice); hipMemcpy(B, hr.data(), (size_t)n * k * 8, hipMemcpyHostToDevice);
  hipMemset(C, 0, (size_t)m * n * 8);
  lpgp_ctx ctx; ctx.cus = 256; ctx.small_tiles_max = 0;
  const int nv = 8 * ((mt * nt + 7) / 8) * 2 + 4096;
  unsigned long long *st, *tl;
  hipMalloc(&st, (size_t)nv * 64); hipMalloc(&tl, (size_t)nv * 64);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(st, 0, (size_t)nv * 64); hipMemset(tl, 0, (size_t)nv * 64);
    GemmArgs g; g.A = A; g.B = tri ? A : B; g.C = C; g.lda = m; g.ldb = tri ? m : n; g.ldc = m; g.mt = mt; g.nt = nt; g.k = k;
    g.alpha = -1; g.beta = beta; g.tri = tri; g.stamps = st; g.timeline = tl;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    launch_gemm(&ctx, 0, 0, 0, g, -1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep < 2) continue;
    std::vector<unsigned long long> h((size_t)nv * 8);
    hipMemcpy(h.data(), tl, (size_t)nv * 64, hipMemcpyDeviceToHost);
    struct W { unsigned long long s, k0, k1, e; unsigned cu; };
    std::map<unsigned, std::vector<W>> percu;
    unsigned long long t0 = ~0ull, t1 = 0; int cnt = 0;
    for (int b = 0; b < nv; ++b) {
      if (!h[8 * b + 3]) continue;
      ++cnt;
      const unsigned hw = (unsigned)h[8 * b + 4];
      // gfx9 HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]; plus the XCC
      const unsigned key = ((unsigned)h[8 * b + 5] << 16) | (hw & 0xff00u);
      percu[key].push_back({h[8 * b], h[8 * b + 1], h[8 * b + 2], h[8 * b + 3], key});
      t0 = std::min(t0, h[8 * b]); t1 = std::max(t1, h[8 * b + 3]);
    }
    const double fl = tri ? (double)m * (m + 1.0) * k : 2.0 * m * n * k;
    printf("mt=%d nt=%d k=%d tri=%d: %.3f ms (events) %.1f TF; %d tiles on %zu CUs; first start -> last store issue %.1f us\n", mt, nt, k, tri, ms,
           fl / ms / 1e9, cnt, percu.size(), (t1 - t0) * 0.01);
    // per CU: time with >= 1 workgroup inside its k-loop, with 2, and with none
    double sum_any = 0, sum_two = 0, sum_pro = 0, sum_kl = 0, sum_epi = 0; int nw = 0;
    for (auto& kv : percu) {
      std::vector<std::pair<unsigned long long, int>> ev;
      for (auto& w : kv.second) { ev.push_back({w.k0, +1}); ev.push_back({w.k1, -1}); sum_pro += (w.k0 - w.s); sum_kl += (w.k1 - w.k0); sum_epi += (w.e - w.k1); ++nw; }
      std::sort(ev.begin(), ev.end());
      int depth = 0; unsigned long long last = ev[0].first;
      for (auto& e : ev) { if (depth >= 1) sum_any += (e.first - last); if (depth >= 2) sum_two += (e.first - last); depth += e.second; last = e.first; }
    }
    {
      std::vector<double> pro;
      for (auto& kv : percu) for (auto& w : kv.second) pro.push_back((w.k0 - w.s) * 0.01);
      std::sort(pro.begin(), pro.end());
      printf("  prologue us: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f   (beta = %g)\n", pro.front(), pro[pro.size() / 10], pro[pro.size() / 2], pro[pro.size() * 9 / 10], pro.back(), beta);
    }
    {
      std::vector<unsigned long long> hs((size_t)nv * 8);
      hipMemcpy(hs.data(), st, (size_t)nv * 64, hipMemcpyDeviceToHost);
      double a[7] = {0, 0, 0, 0, 0, 0, 0}; int c2 = 0;
      for (int b = 0; b < nv; ++b) if (hs[8 * b + 5]) { for (int j = 0; j < 7; ++j) a[j] += hs[8 * b + j]; ++c2; }
      const int KT = k / 16;
      printf("  core clock in the k-loop %.3f GHz; per stage (core cycles): dma-issue %.0f  frag+mfma %.0f  vmwait %.0f  barrier %.0f ; k-loop %.0f cycles/tile = %.2f cycles per MFMA per wave\n",
             a[5] / a[6] * 0.1, a[0] / c2 / KT, a[1] / c2 / KT, a[2] / c2 / KT, a[3] / c2 / KT, a[5] / c2, a[5] / c2 / (KT * 256.0));
    }
    const double span = (t1 - t0) * 0.01, ncu = (double)percu.size();
    printf("  per workgroup (us): prologue %.1f  k-loop %.1f  epilogue-issue %.1f\n", sum_pro / nw * 0.01, sum_kl / nw * 0.01, sum_epi / nw * 0.01);
    printf("  per CU over the %.1f us span: >=1 workgroup in its k-loop %.1f us (%.1f %%), 2 in k-loop %.1f us (%.1f %%)\n", span,
           sum_any / ncu * 0.01, 100 * sum_any / ncu * 0.01 / span, sum_two / ncu * 0.01, 100 * sum_two / ncu * 0.01 / span);
    // one CU in detail
    auto& v = percu.begin()->second;
    std::sort(v.begin(), v.end(), [](const W& a, const W& b) { return a.s < b.s; });
    printf("  CU %06x:", percu.begin()->first);
    for (size_t i = 0; i < v.size() && i < 12; ++i) printf("  [%.1f %.1f %.1f %.1f]", (v[i].s - t0) * 0.01, (v[i].k0 - t0) * 0.01, (v[i].k1 - t0) * 0.01, (v[i].e - t0) * 0.01);
    printf("\n");
  }
  return 0;
}
