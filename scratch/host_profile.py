"""Where the HOST time of a c3 step goes (Python + ctypes), cProfile over 20 steps; the device runs asynchronously, so
functions that wait for it (stream synchronisation inside the factorisation's status read-back, result read-backs) show up
with their waiting time."""
import cProfile, pstats, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import problems
wl = problems.poisson_2d() if len(sys.argv) < 2 else getattr(problems, sys.argv[1])()
dev = problems.upload(wl)
prior = problems.build_prior(wl)
for _ in range(3):
    u, m, v = problems.condition_and_predict(wl, prior=prior, device_arrays=dev); u = None
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(20):
    u, m, v = problems.condition_and_predict(wl, prior=prior, device_arrays=dev); u = None
pr.disable()
print(f"{(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per step under cProfile")
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
