"""Host time of the pieces of `predict` before its first kernel (c3): where do the ~150 us between the end of the factorisation
and the first prediction kernel go?"""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import problems, _engine
wl = problems.poisson_2d()
dev = problems.upload(wl)
prior = problems.build_prior(wl)
acc = {}
def tic(name, t0):
    t1 = time.perf_counter(); acc.setdefault(name, []).append((t1 - t0) * 1e6); return t1
for rep in range(12):
    u, m, v = problems.condition_and_predict(wl, prior=prior, device_arrays=dev, want_var=False)   # conditioning (+ a cheap predict)
    ctx = u._state.ctx; ctx.sync()
    x = dev["test"]
    t = time.perf_counter()
    u._check_current(); t = tic("_check_current", t)
    X, batch = u._flat(x); t = tic("_flat", t)
    u._state.residual_key = None
    r = u._residual(); t = tic("_residual (numpy)", t)
    u._state.mat.set_residual(r); t = tic("set_residual (H2D)", t)
    u._state.residual_key = len(u._blocks)
    pts = _engine.as_points(ctx, x, X); t = tic("as_points", t)
    rhs = u._cross(pts); t = tic("_cross (rhs create + 5 launches)", t)
    pm = u._prior_mean_at(X, X.shape[0]); t = tic("_prior_mean_at", t)
    kxx = np.full(X.shape[0], u._prior_diag()); t = tic("kxx", t)
    out = rhs.predict(pm, kxx, want_mean=True, want_var=True); t = tic("rhs.predict (blocking)", t)
    u = None
for k, v in acc.items():
    print(f"{k:36s} median {sorted(v)[len(v)//2]:9.1f} us")
