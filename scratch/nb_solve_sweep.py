import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import _engine, problems, randvars
ctx = _engine.default_context()
wl = problems.poisson_2d(128, m_side=64)
lp.config.gram_capacity_hint = wl.n_total
dev = problems.upload(wl)
prior = problems.build_prior(wl)
for nbs in (512, 256, 384, 640, 768, 1024, 1536, 2048, 512):
    best = 1e9
    for rep in range(4):
        ctx.set_option("nb", 512)
        u = prior
        for i, o in enumerate(wl.observations):
            n = o.X.shape[0]
            b = None if o.noise_var is None else randvars.Normal(np.zeros(o.X_as_given()[1].shape), np.full(n, o.noise_var))
            u = u.condition_on_observations(o.X_as_given()[1], X=dev["obs"][i], L=problems.operator_of(o.op, 2), b=b)
        ctx.sync(); ctx.set_option("nb", nbs); t0 = time.perf_counter()
        m, v = u.predict(dev["test"]); ctx.sync(); best = min(best, time.perf_counter() - t0)
    print(f"nb_solve={nbs}: predict {best*1e3:.2f} ms", flush=True)
