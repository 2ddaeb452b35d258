import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import _engine, problems
ctx = _engine.default_context()
wl = problems.poisson_2d(128, m_side=64)
lp.config.gram_capacity_hint = wl.n_total
dev = problems.upload(wl); prior = problems.build_prior(wl)
u, m, v = problems.condition_and_predict(wl, prior=prior, device_arrays=dev)
for rep in range(2):
    ctx.sync(); t0 = time.perf_counter(); m, v = u.predict(dev["test"]); ctx.sync(); print(f"predict {1e3*(time.perf_counter()-t0):.2f} ms")
