"""Where the c3 step goes, phase by phase (each phase bracketed by a device sync: the sum exceeds the pipelined step)."""
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
for rep in range(3):
    ts = []
    u = prior
    ctx.sync(); tstart = t0 = time.perf_counter()
    for i, o in enumerate(wl.observations):
        n = o.X.shape[0]
        b = None if o.noise_var is None else randvars.Normal(np.zeros(o.X_as_given()[1].shape), np.full(n, o.noise_var))
        u = u.condition_on_observations(o.X_as_given()[1], X=dev["obs"][i], L=problems.operator_of(o.op, 2), b=b)
        ctx.sync(); ts.append(time.perf_counter() - t0); t0 = time.perf_counter()
    m, v = u.predict(dev["test"]); ctx.sync(); tv = time.perf_counter() - t0
    print("cond ms:", [round(t * 1e3, 2) for t in ts], "predict", round(tv * 1e3, 2), "sum", round((time.perf_counter() - tstart) * 1e3, 2))
# python-side overhead of a step with the device work removed: time the host part of predict's setup
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
u2, m, v = problems.condition_and_predict(wl, prior=prior, device_arrays=dev)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(18)
