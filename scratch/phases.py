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
for nb in (512, 256, 1024):
  ctx.set_option("nb", nb)
  for rep in range(2):
    ts = []
    u = prior
    t0 = time.perf_counter()
    for i, o in enumerate(wl.observations):
        n = o.X.shape[0]
        b = None if o.noise_var is None else randvars.Normal(np.zeros(o.X_as_given()[1].shape), np.full(n, o.noise_var))
        u = u.condition_on_observations(o.X_as_given()[1], X=dev["obs"][i], L=problems.operator_of(o.op, 2), b=b)
        ctx.sync(); ts.append(time.perf_counter() - t0); t0 = time.perf_counter()
    m = u.predict(dev["test"], return_var=False); ctx.sync(); tm = time.perf_counter() - t0; t0 = time.perf_counter()
    m, v = u.predict(dev["test"]); ctx.sync(); tv = time.perf_counter() - t0
    print(f"nb={nb} cond ms:", [round(t * 1e3, 2) for t in ts], "mean-only", round(tm * 1e3, 2), "mean+var", round(tv * 1e3, 2))
# inside the last conditioning: assemble / potrf / weights
ctx.set_option("nb", 512)
mat = _engine.GramMatrix(ctx, wl.n_total)
from linpde_gp_amd.randprocs import covfuncs
import linpde_gp_amd.randprocs._gaussian_process as G
pts = [d_._lpgp_points for d_ in dev["obs"]]
coeffs = [o.op for o in wl.observations]
t0 = time.perf_counter()
for bi in range(5):
    mat.add_block(pts[bi].n)
    for bj in range(bi + 1):
        k = covfuncs.DifferentiatedCovarianceFunction(prior.cov, *G._combine(prior.cov, coeffs[bi], coeffs[bj]))
        mat.assemble(k.lower(), pts[bi], None if bi == bj else pts[bj], bi, bj)
    if wl.observations[bi].noise_var: mat.add_diag(bi, None, wl.observations[bi].noise_var)
ctx.sync(); ta = time.perf_counter() - t0; t0 = time.perf_counter()
info = mat.potrf(); ctx.sync(); tp = time.perf_counter() - t0; t0 = time.perf_counter()
w = mat.solve_weights(np.ones(wl.n_total)); ctx.sync(); tw = time.perf_counter() - t0
print(f"one-shot: assemble {ta*1e3:.2f} ms, potrf {tp*1e3:.2f} ms (info {info}), weights {tw*1e3:.2f} ms")
for la in (0, 1):
    ctx.set_option("lookahead", la)
    mat2 = _engine.GramMatrix(ctx, wl.n_total)
    for bi in range(5):
        mat2.add_block(pts[bi].n)
        for bj in range(bi + 1):
            k = covfuncs.DifferentiatedCovarianceFunction(prior.cov, *G._combine(prior.cov, coeffs[bi], coeffs[bj]))
            mat2.assemble(k.lower(), pts[bi], None if bi == bj else pts[bj], bi, bj)
        if wl.observations[bi].noise_var: mat2.add_diag(bi, None, wl.observations[bi].noise_var)
    ctx.sync(); t0 = time.perf_counter(); mat2.potrf(); ctx.sync()
    print(f"lookahead={la}: potrf {1e3*(time.perf_counter()-t0):.2f} ms")
    del mat2
