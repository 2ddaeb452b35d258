"""A few steps of one small problem for a kernel trace: python scratch/small_trace.py <c1|p32> [steps]"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import linpde_gp_amd as lp
from linpde_gp_amd import problems
wl = {"c1": lambda: problems.poisson_1d(512, n_bdry_repeats=16, noise_var=1e-4, m=256), "p32": lambda: problems.poisson_2d(n_side=32, m_side=16),
      "heat": problems.heat_reference}[sys.argv[1]]()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lp.config.lazy_factorization = True
dev = problems.upload(wl)
prior = problems.build_prior(wl)
ctx = lp._engine.default_context()
for _ in range(5):
    u, m, v = problems.condition_and_predict(wl, prior=prior, device_arrays=dev); u = None
ctx.sync()
t0 = time.perf_counter()
for _ in range(n):
    u, m, v = problems.condition_and_predict(wl, prior=prior, device_arrays=dev); u = None
ctx.sync()
print(f"{sys.argv[1]}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per step")
