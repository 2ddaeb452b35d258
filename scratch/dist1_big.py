# world-size-1 RCCL path at the weak-scaling sizes vs the plain single-GPU path (same GPU)
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import _dist, _engine, problems
n_side = int(os.environ.get("N_SIDE", "182")); m_side = n_side // 2
wl = problems.poisson_2d(n_side=n_side, m_side=m_side)
lp.config.gram_capacity_hint = wl.n_total
ctx = _engine.default_context()
dev = problems.upload(wl); prior = problems.build_prior(wl)
def run():
    t0 = time.perf_counter(); u, mean, var = problems.condition_and_predict(wl, prior=prior, device_arrays=dev); ctx.sync()
    return mean, var, time.perf_counter() - t0
run(); m0, v0, t = run(); print(f"single: {t*1e3:.1f} ms  {wl.total_flops()/t/1e12:.1f} TF")
os.environ["LPGP_FORCE_RCCL"] = "1"
ctx.dist_init(_dist.Comm(0, 1))
run(); m1, v1, t = run(); print(f"rccl world=1: {t*1e3:.1f} ms  {wl.total_flops()/t/1e12:.1f} TF")
print("mean diff", np.max(np.abs(m1 - m0)) / np.max(np.abs(m0)), "var diff", np.max(np.abs(v1 - v0)) / np.max(np.abs(v0)))
