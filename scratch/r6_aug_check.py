import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "linpde-gp_amd"))
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import problems, _engine
ctx = _engine.default_context()
print("ride_aug", ctx.get_option("ride_aug"), "cap hint", lp.config.gram_capacity_hint)
lp.config.lazy_factorization = True
wl = problems.poisson_2d(n_side=int(os.environ.get("NSIDE", 64)), m_side=32)
dev = problems.upload(wl)
out = {}
for aug in (0, 1):
    ctx.set_option("ride_aug", aug)
    ctx.profile_reset(); ctx.profile_enable(True)
    u, mean, var = problems.condition_and_predict(wl, device_arrays=dev)
    prof = ctx.profile_get(); ctx.profile_enable(False)
    print("aug", aug, {k: (v["launches"], round(v["ms"], 3)) for k, v in prof.items() if v["launches"]})
    out[aug] = (mean, var)
print("mean diff", np.max(np.abs(out[0][0] - out[1][0])) / np.max(np.abs(out[0][0])), "var diff", np.max(np.abs(out[0][1] - out[1][1])) / np.max(np.abs(out[0][1])))
