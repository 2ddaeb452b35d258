"""EXPERIMENT: conditional refinement of the tile solves (per-tile flag from the tile Cholesky: refine iff max l_ii > thr * min l_ii).
Per workload: tiles refined / not, and the distance of mean and variance from the always-refined run."""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import problems
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); import _hooks      # test hooks: liblpgp_testhooks.so
names = sys.argv[1:] or ["poisson2d", "poisson1d", "heat"]
for name in names:
    wl = {"poisson2d": lambda: problems.poisson_2d(), "poisson1d": lambda: problems.poisson_1d(), "heat": lambda: problems.heat_1d(),
          "c1": lambda: problems.poisson_1d(512, n_bdry_repeats=16, noise_var=1e-4, m=256), "heatref": problems.heat_reference}[name]()
    u, m, v = problems.condition_and_predict(wl)
    ctx = lp._engine.default_context()
    ctx.sync()
    print(name, "N_tot", wl.n_total, "mean max", float(np.abs(m).max()), "var max", float(np.abs(v).max()), flush=True)
    np.save(f"/tmp/x_{name}_{os.environ.get('LPGP_X_REFINE_THR','0')}_m.npy", m)
    np.save(f"/tmp/x_{name}_{os.environ.get('LPGP_X_REFINE_THR','0')}_v.npy", v)
    import ctypes as C
    from linpde_gp_amd._lib import lib
    out = (C.c_int32 * 8)()
    _hooks.lib.lpgp_debug_tile_xcc(ctx._h, out, 0)
