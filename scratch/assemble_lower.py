"""Rate of the Gram (lower-triangle) assembly launch on scattered points vs the rectangular cross-covariance launch."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import _engine, problems
ctx = _engine.default_context()
for n in (16384, 8192):
    wl = problems.scattered_2d(n=n, m=4096)
    prior = problems.build_prior(wl)
    o = wl.observations[0]
    b = lp.randvars.Normal(np.zeros(n), np.full(n, o.noise_var))
    for ct in (4, 1):
        ctx.set_option("asm_ct", ct)
        best = 1e9
        for rep in range(4):
            ctx.profile_reset(); ctx.profile_enable(["assemble"])
            u = prior.condition_on_observations(o.Y, X=o.X, b=b)
            ctx.sync(); p = ctx.profile_get()["assemble"]; ctx.profile_enable(False)
            best = min(best, p["ms"]); u = None
        print(f"n = {n}, asm_ct {ct}: Gram lower triangle {p['bytes'] / 1e9:.3f} GB in {p['launches']} launch(es): best {best:.4f} ms -> {p['bytes'] / best / 1e6:.0f} GB/s", flush=True)
ctx.set_option("asm_ct", 4)
