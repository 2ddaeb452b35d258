"""Host side of a c3 bench step (lazy mode, the timed region's calls): cProfile over 20 steps.  Usage: python3 scratch/r6_host_profile.py [workload]"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "linpde-gp_amd"))
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import problems, _engine

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
wl = {"c3": problems.poisson_2d, "c2": problems.poisson_1d}[which]()
ctx = _engine.default_context()
lp.config.lazy_factorization = True
lp.config.gram_capacity_hint = wl.n_total
dev = problems.upload(wl)
prior = problems.build_prior(wl)
for _ in range(3): problems.condition_and_predict(wl, prior=prior, device_arrays=dev)
ctx.sync()
n = 20
t0 = time.perf_counter()
for _ in range(n): problems.condition_and_predict(wl, prior=prior, device_arrays=dev)
ctx.sync()
print(f"{which}: {1e3 * (time.perf_counter() - t0) / n:.3f} ms per step")
pr = cProfile.Profile(); pr.enable()
for _ in range(n): problems.condition_and_predict(wl, prior=prior, device_arrays=dev)
ctx.sync(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print("\n".join(l[:160] for l in s.getvalue().splitlines()[4:44]))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(22)
print("\n".join(l[:160] for l in s.getvalue().splitlines()[4:36]))
