"""cProfile (cumulative) of the host side of a small step (N_tot = 1 152), lazy mode, points resident."""
import cProfile, pstats, sys, io
sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import linpde_gp_amd as lp
from linpde_gp_amd import problems
lp.config.lazy_factorization = True
wl = problems.poisson_2d(n_side=32, m_side=16)
dev = problems.upload(wl); prior = problems.build_prior(wl)
for _ in range(20):
    u, m, v = problems.condition_and_predict(wl, prior=prior, device_arrays=dev); u = None
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    u, m, v = problems.condition_and_predict(wl, prior=prior, device_arrays=dev); u = None
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
