import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import _engine, problems
ctx = _engine.default_context()
for name, wl in (("c3", problems.poisson_2d(128, m_side=64)), ("c2", problems.poisson_1d())):
    lp.config.gram_capacity_hint = wl.n_total
    dev = problems.upload(wl); prior = problems.build_prior(wl)
    for th in (256, 64, 128, 384, 512, 768, 256):
        ctx.set_option("small_tiles_max", th)
        best = 1e9
        for rep in range(5):
            ctx.sync(); t0 = time.perf_counter()
            problems.condition_and_predict(wl, prior=prior, device_arrays=dev)
            ctx.sync(); best = min(best, time.perf_counter() - t0)
        print(f"{name} small_tiles_max={th}: {best*1e3:.2f} ms", flush=True)
