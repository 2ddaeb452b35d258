"""One GPU, a Gram matrix of half the device's HBM: noisy values at n scattered points (default 131 072: 137 GB in fp64),
conditioning + prediction through the host API, checked by size-independent properties:
 * the representer weights solve the system: || (K + s^2 I) w - r || / || r ||, with K w formed MATRIX-FREE by the kernel-product
   kernel (`CovarianceFunction.linop`, an evaluation path that shares nothing with the assembled matrix or its factor);
 * the posterior variance is within [0, prior variance] and the posterior mean at a subset of the training points reproduces
   (K w)(x) there.
usage: capacity_run.py [n] [m]"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import problems

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
m = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
wl = problems.scattered_2d(n=n, m=m, noise_var=1e-2, seed=1)
prior = problems.build_prior(wl)
o = wl.observations[0]
ctx = lp._engine.default_context()
t0 = time.perf_counter()
u = prior.condition_on_observations(o.Y, X=o.X, b=lp.randvars.Normal(np.zeros(n), o.noise_var))
mean, var = u.predict(wl.Xtest)
ctx.sync()
t1 = time.perf_counter()
flops = n**3 / 3 + n * n * m
print(f"n = {n}  m = {m}: Gram matrix {8 * n * n / 1e9:.1f} GB; condition + predict {t1 - t0:.2f} s = {flops / (t1 - t0) / 1e12:.1f} TFLOP/s (first call, pools cold)", flush=True)
mean1, var1 = mean, var
del u                                                   # (one matrix of this size at a time: its storage returns to the context's pool)
t0 = time.perf_counter()
u = prior.condition_on_observations(o.Y, X=o.X, b=lp.randvars.Normal(np.zeros(n), o.noise_var))
mean, var = u.predict(wl.Xtest)
ctx.sync()
t1 = time.perf_counter()
print(f"second call: {t1 - t0:.2f} s = {flops / (t1 - t0) / 1e12:.1f} TFLOP/s; identical results: {np.array_equal(mean, mean1) and np.array_equal(var, var1)}", flush=True)
w = u.representer_weights
Kw = prior.cov.linop(o.X, o.X) @ w                         # matrix-free
r = o.Y
res = np.linalg.norm(Kw + o.noise_var * w - r) / np.linalg.norm(r)
pv = float(prior.cov(wl.Xtest[:1], wl.Xtest[:1]).ravel()[0]) if hasattr(prior.cov, "__call__") else float("nan")
idx = np.arange(0, n, max(1, n // 512))[:512]
m_tr = u.mean(o.X[idx])
print(f"|| (K + s^2 I) w - r || / || r || = {res:.2e}  (matrix-free K w)")
print(f"posterior variance in [{var.min():.3e}, {var.max():.3e}], prior variance {pv:.3f}; mean at 512 training points vs (K w)(x): {np.abs(m_tr - Kw[idx]).max() / np.abs(Kw).max():.2e}")
ok = res < 1e-10 and var.min() >= 0.0 and var.max() <= pv * (1 + 1e-12) and np.abs(m_tr - Kw[idx]).max() / np.abs(Kw).max() < 1e-9
print("PASS" if ok else "FAIL")
sys.exit(0 if ok else 1)
