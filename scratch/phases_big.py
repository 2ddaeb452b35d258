import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import _engine, problems, randvars
ctx = _engine.default_context()
for kv in os.environ.get("LPGP_OPTS", "").split(","):
    if "=" in kv:
        k_, v_ = kv.split("="); ctx.set_option(k_, int(v_)); print("option", k_, v_)
n_side = int(os.environ.get("N_SIDE", "256")); m_side = int(os.environ.get("M_SIDE", "128"))
wl = problems.poisson_2d(n_side, m_side=m_side)
lp.config.gram_capacity_hint = wl.n_total
dev = problems.upload(wl)
prior = problems.build_prior(wl)
for rep in range(2):
    ts = []
    u = prior
    t0 = time.perf_counter()
    for i, o in enumerate(wl.observations):
        n = o.X.shape[0]
        b = None if o.noise_var is None else randvars.Normal(np.zeros(o.X_as_given()[1].shape), np.full(n, o.noise_var))
        u = u.condition_on_observations(o.X_as_given()[1], X=dev["obs"][i], L=problems.operator_of(o.op, 2), b=b)
        ctx.sync(); ts.append(time.perf_counter() - t0); t0 = time.perf_counter()
    m, v = u.predict(dev["test"]); ctx.sync(); tv = time.perf_counter() - t0
    N = wl.n_total; M = m_side * m_side
    print(f"cond ms: {[round(t * 1e3, 1) for t in ts]} ({N**3/3/ts[-1]/1e12:.1f} TF incl. assembly)  mean+var {tv*1e3:.1f} ms ({(N*N*M)/tv/1e12:.1f} TF)")
    del u
