"""Where the per-entry assembly kernel's time goes: LPGP_ASM_DIAG = 0 (product) / 1 (stores only) / 2 (evaluation only) /
4 (per-entry exp instead of the per-point factors) on the 4096 x 16384 cross-covariance of c3 (run once per setting)."""
import os, sys; sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
from linpde_gp_amd import _engine, problems
ctx = _engine.default_context()
wl = problems.poisson_2d()
k = problems.build_prior(wl).cov
D = problems.operator_of(wl.observations[-1].op, 2)
Xobs = np.ascontiguousarray(wl.observations[-1].X); Xt = wl.Xtest
kk = D(k, argnum=1)
for rep in range(3):
    ctx.profile_reset(); ctx.profile_enable(["assemble"])
    M = kk.matrix(Xt, Xobs)
    ctx.sync(); p = ctx.profile_get()["assemble"]; ctx.profile_enable(False)
print(f"LPGP_ASM_DIAG={os.environ.get('LPGP_ASM_DIAG', '0')}: {p['ms']:.3f} ms -> {p['bytes'] / p['ms'] / 1e6:.0f} GB/s", flush=True)
