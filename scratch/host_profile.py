# where the HOST time of a small step goes (N_tot = 1152: five conditionings + prediction)
import cProfile, pstats, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "linpde-gp_amd")
import linpde_gp_amd as lp
from linpde_gp_amd import _engine, problems
ctx = _engine.default_context()
wl = problems.poisson_2d(32, m_side=16)
lp.config.gram_capacity_hint = wl.n_total
dev = problems.upload(wl); prior = problems.build_prior(wl)
for _ in range(5): problems.condition_and_predict(wl, prior=prior, device_arrays=dev)
ctx.sync(); t0 = time.perf_counter()
for _ in range(50): problems.condition_and_predict(wl, prior=prior, device_arrays=dev)
ctx.sync(); print("ms per step", (time.perf_counter() - t0) / 50 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(50): problems.condition_and_predict(wl, prior=prior, device_arrays=dev)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
