"""cProfile (tottime) of the host side of small steps, lazy mode, points resident: python scratch/host_profile_tottime.py <c1|p32>"""
import cProfile, pstats, sys, io
sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import linpde_gp_amd as lp
from linpde_gp_amd import problems
lp.config.lazy_factorization = True
wl = problems.poisson_1d(512, n_bdry_repeats=16, noise_var=1e-4, m=256) if sys.argv[1] == "c1" else problems.poisson_2d(n_side=32, m_side=16)
dev = problems.upload(wl); prior = problems.build_prior(wl)
for _ in range(20):
    u, m, v = problems.condition_and_predict(wl, prior=prior, device_arrays=dev); u = None
pr = cProfile.Profile(); pr.enable()
for _ in range(300):
    u, m, v = problems.condition_and_predict(wl, prior=prior, device_arrays=dev); u = None
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(32); print(s.getvalue()[:7000])
