"""Per-entry assembly in steady state: the 16384 x 4096 cross-covariance block of the scattered-points workload assembled 1, 5
and 200 times back to back (HIP-event profiling slot "assemble": summed kernel time / launches)."""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import _engine, problems
ctx = _engine.default_context()
wl = problems.scattered_2d(n=16384, m=4096)
u, _, _ = problems.condition_and_predict(wl, want_var=False)
pts = _engine.as_points(ctx, None, wl.Xtest)
for reps in (1, 5, 200, 5, 1):
    time.sleep(0.3)
    ctx.profile_reset(); ctx.profile_enable(["assemble"])
    keep = [u._cross(pts) for _ in range(reps)]
    ctx.sync(); p = ctx.profile_get()["assemble"]; ctx.profile_enable(False)
    keep = None
    print(f"{reps:4d} back-to-back assemblies: {p['ms'] / p['launches']:.4f} ms per launch -> {p['bytes'] / p['ms'] / 1e6:.0f} GB/s", flush=True)
