import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import _engine, problems
ctx = _engine.default_context()
wl = problems.poisson_2d(128, m_side=64)
lp.config.gram_capacity_hint = wl.n_total
dev = problems.upload(wl); prior = problems.build_prior(wl)
def run(n=4):
    ts = []
    for _ in range(n):
        ctx.sync(); t0 = time.perf_counter()
        problems.condition_and_predict(wl, prior=prior, device_arrays=dev); ctx.sync()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
run(2)
print("default", round(run(), 2))
for nbb in (1024, 768):
    for mt in (32, 48, 64, 96):
        ctx.set_option("nb_big", nbb); ctx.set_option("nb_big_min_tiles", mt)
        print("nb_big", nbb, "min_tiles", mt, round(run(), 2))
ctx.set_option("nb_big", 0)
print("default again", round(run(), 2))
