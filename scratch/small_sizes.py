"""Small problems -- the reference's real operating range: ms per condition+predict step of c1, the
reference's own heat problem and small 2-D Poisson grids, points resident (as in bench.py) and handed over as NumPy arrays
(the reference's calling convention), lazy and eager status; then a cProfile of the N_tot = 1 152 step."""
import cProfile, pstats, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import problems

def per_step(wl, dev, n=50):
    prior = problems.build_prior(wl)
    for _ in range(5):
        u, m, v = problems.condition_and_predict(wl, prior=prior, device_arrays=dev); u = None
    lp._engine.default_context().sync()
    t0 = time.perf_counter()
    for _ in range(n):
        u, m, v = problems.condition_and_predict(wl, prior=prior, device_arrays=dev); u = None
    lp._engine.default_context().sync()
    return (time.perf_counter() - t0) / n * 1e3

cases = [("c1 poisson1d 512+32", problems.poisson_1d(512, n_bdry_repeats=16, noise_var=1e-4, m=256)),
         ("heat_reference 2105", problems.heat_reference()),
         ("poisson2d 32x32 (1152)", problems.poisson_2d(n_side=32, m_side=16)),
         ("poisson2d 48x48 (2496)", problems.poisson_2d(n_side=48, m_side=24)),
         ("poisson2d 64x64 (4352)", problems.poisson_2d(n_side=64, m_side=32))]
for name, wl in cases:
    row = []
    for lazy in (True, False):
        lp.config.lazy_factorization = lazy
        row.append(per_step(wl, problems.upload(wl)))
        row.append(per_step(wl, None))
    print(f"{name:28s} N_tot={wl.n_total:5d} M={wl.Xtest.shape[0]:5d}: lazy resident {row[0]:.3f} ms, lazy numpy {row[1]:.3f}; eager resident {row[2]:.3f}, eager numpy {row[3]:.3f}", flush=True)
lp.config.lazy_factorization = True
wl = cases[2][1]
dev = problems.upload(wl)
prior = problems.build_prior(wl)
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    u, m, v = problems.condition_and_predict(wl, prior=prior, device_arrays=dev); u = None
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
