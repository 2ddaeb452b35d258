import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import _engine, config, domains
from linpde_gp_amd.linfuncops import diffops
cf = lp.randprocs.covfuncs
ctx = _engine.default_context()
k = 4.0 * cf.TensorProduct(cf.Matern((), nu=2.5), cf.Matern((), nu=2.5))
D = -1.0 * diffops.Laplacian((2,))
kk = D(D(k, argnum=1), argnum=0)
n = 128
g = np.linspace(-1, 1, n); g2 = np.linspace(-0.99, 0.99, n)
X0 = _engine.to_device(domains.TensorProductGrid(g, g)); X1 = _engine.to_device(domains.TensorProductGrid(g2, g2))
P0, P1 = X0._lpgp_points, X1._lpgp_points
mat = _engine.GramMatrix(ctx, 2 * n * n)
mat.add_block(n * n); mat.add_block(n * n)
def timed(fn, reps=5):
    fn(); ctx.sync(); ctx.profile_reset(); ctx.profile_enable(["assemble"])
    for _ in range(reps): fn()
    ctx.sync(); p = ctx.profile_get()["assemble"]; ctx.profile_enable(False)
    return p["ms"] / reps, p["bytes"] / reps
for grid in (True, False):
    if not grid: P0.grid_factors = None; P1.grid_factors = None
    ms, by = timed(lambda: mat.assemble(kk.lower(), P1, P0, 1, 0))
    print(f"grid={grid} full 16384x16384 block: {ms:.3f} ms  {by/ms/1e6:.0f} GB/s")
    ms, by = timed(lambda: mat.assemble(kk.lower(), P0, None, 0, 0))
    print(f"grid={grid} lower 16384x16384 block: {ms:.3f} ms  {by/ms/1e6:.0f} GB/s")
