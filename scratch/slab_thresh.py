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
def run(tag, reps=4):
    best = [1e9, 1e9]
    for rep in range(reps):
        u = prior
        for i, o in enumerate(wl.observations):
            n = o.X.shape[0]
            b = None if o.noise_var is None else randvars.Normal(np.zeros(o.X_as_given()[1].shape), np.full(n, o.noise_var))
            if i == len(wl.observations) - 1:
                ctx.sync(); t0 = time.perf_counter()
            u = u.condition_on_observations(o.X_as_given()[1], X=dev["obs"][i], L=problems.operator_of(o.op, 2), b=b)
        ctx.sync(); tc = time.perf_counter() - t0; t0 = time.perf_counter()
        m, v = u.predict(dev["test"]); ctx.sync(); tv = time.perf_counter() - t0
        best = [min(best[0], tc), min(best[1], tv)]
    print(f"{tag}: last conditioning {best[0]*1e3:.2f} ms   predict {best[1]*1e3:.2f} ms", flush=True)
for th in (1 << 30, 0, 32, 64, 96, 1 << 30):
    ctx.set_option("trsm_slab_max_tiles", th); ctx.set_option("trsv_slab_max_tiles", th)
    run(f"slab while <= {th} tile rows remain")
