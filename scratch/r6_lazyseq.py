import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "linpde-gp_amd"))
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import problems, _engine
wl = problems.poisson_1d(512, n_bdry_repeats=16, noise_var=1e-4, m=256) if (len(sys.argv) < 2 or sys.argv[1] == "c1") else problems.poisson_1d()
ctx = _engine.default_context()
lp.config.gram_capacity_hint = wl.n_total
dev = problems.upload(wl); prior = problems.build_prior(wl)
lp.config.lazy_factorization, lp.config.variance_with_mean = True, True
def conditioned():
    u = prior
    for i, o in enumerate(wl.observations):
        Y = o.Y if o.grid is None else o.Y.reshape(tuple(len(f) for f in o.grid))
        b = None if o.noise_var is None else lp.randvars.Normal(np.zeros(Y.shape), np.full(o.X.shape[0], o.noise_var))
        u = u.condition_on_observations(Y, X=dev["obs"][i], L=problems.operator_of(o.op, wl.d), b=b)
    return u
T = {"cond": [], "mean": [], "std": [], "del": []}
for it in range(30):
    t0 = time.perf_counter(); u = conditioned(); t1 = time.perf_counter(); m = u.mean(dev["test"]); t2 = time.perf_counter(); s = u.std(dev["test"]); t3 = time.perf_counter()
    u = None; m = None; s = None; t4 = time.perf_counter()
    if it >= 5:
        T["cond"].append(t1 - t0); T["mean"].append(t2 - t1); T["std"].append(t3 - t2); T["del"].append(t4 - t3)
print({k: round(float(np.median(v)) * 1e3, 3) for k, v in T.items()})
