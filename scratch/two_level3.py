import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import _engine, problems, randvars
ctx = _engine.default_context()
n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 128
wl = problems.poisson_2d(n_side, m_side=n_side // 2)
lp.config.gram_capacity_hint = wl.n_total
dev = problems.upload(wl)
prior = problems.build_prior(wl)
def run(tag, reps=3):
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
    print(f"{tag}: last conditioning {best[0]*1e3:.2f} ms   predict {best[1]*1e3:.2f} ms   mean_max {m.max():.15f}", flush=True)
cfgs = [(0, 0), (1024, 96), (1024, 80), (1024, 64), (1024, 48), (1024, 32), (1536, 64), (2048, 64), (2048, 88), (0, 0)]
if len(sys.argv) > 2:
    cfgs = [(0, 0), (1024, 64), (1024, 128), (2048, 128), (2048, 256)]
for nbo, mn in cfgs:
    ctx.set_option("nb_outer", nbo); ctx.set_option("nb_outer_min_tiles", mn)
    run(f"nb_outer={nbo} min={mn}", reps=3 if n_side <= 128 else 2)
