import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "linpde-gp_amd"))
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import problems, _engine
wl = problems.poisson_1d(512, n_bdry_repeats=16, noise_var=1e-4, m=256)
if len(sys.argv) > 1:
    from oracle import workloads as owl
    ref = owl.run(wl)
ctx = _engine.default_context()
lp.config.gram_capacity_hint = wl.n_total
dev = problems.upload(wl); prior = problems.build_prior(wl)
def conditioned():
    u = prior
    for i, o in enumerate(wl.observations):
        Y = o.Y if o.grid is None else o.Y.reshape(tuple(len(f) for f in o.grid))
        b = None if o.noise_var is None else lp.randvars.Normal(np.zeros(Y.shape), np.full(o.X.shape[0], o.noise_var))
        u = u.condition_on_observations(Y, X=dev["obs"][i], L=problems.operator_of(o.op, wl.d), b=b)
    return u
def sequence():
    u = conditioned(); t0 = time.perf_counter(); m = u.mean(dev["test"]); t1 = time.perf_counter(); s = u.std(dev["test"]); t2 = time.perf_counter(); return t1 - t0, t2 - t1
def run(tag, lazy, vwm, n=11):
    lp.config.lazy_factorization, lp.config.variance_with_mean = lazy, vwm
    ts, tm, tsd = [], [], []
    for _ in range(n):
        t0 = time.perf_counter(); a, b = sequence(); ctx.sync(); ts.append((time.perf_counter() - t0) * 1e3); tm.append(a * 1e3); tsd.append(b * 1e3)
    print(tag, "total", round(float(np.median(ts)), 3), "mean()", round(float(np.median(tm)), 3), "std()", round(float(np.median(tsd)), 3))
lp.config.lazy_factorization = True
for _ in range(5): problems.condition_and_predict(wl, prior=prior, device_arrays=dev)
run("eager", False, False); run("eager vwm", False, True); run("lazy vwm", True, True)
