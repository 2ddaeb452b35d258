"""Round 6: matrix-free conditioning at N = 32 768 -- the device-resident iteration (lpgp_pcg_step, operands in HBM) against the
host loop of round 5 (NumPy vector algebra around lpgp_kernel_matvec).  Same preconditioner, same tolerance."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "linpde-gp_amd"))
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import problems

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
wl = problems.scattered_2d(n=n, m=48, noise_var=1e-2, seed=3)
o = wl.observations[0]
prior = problems.build_prior(wl)
b = lp.randvars.Normal(np.zeros(o.X.shape[0]), np.full(o.X.shape[0], o.noise_var))
lp.config.matrix_free = True
lp.config.matrix_free_rtol = 1e-11
res = {}
for dev in (False, True, False, True):
    lp.config.matrix_free_device_iteration = dev
    t0 = time.perf_counter()
    free = prior.condition_on_observations(o.Y, o.X, b=b)
    free._preconditioner()
    t1 = time.perf_counter()
    w = free.representer_weights
    t2 = time.perf_counter()
    info = free.last_solve_info
    m, v = free.predict(wl.Xtest)
    t3 = time.perf_counter()
    print(f"N={n} device_iteration={dev}: preconditioner {t1 - t0:.3f} s, weights {t2 - t1:.3f} s ({info['iterations']} iterations, "
          f"{(t2 - t1) / max(info['iterations'], 1) * 1e3:.2f} ms each, rel {np.max(info['rel_residual']):.1e}), predict 48 points {t3 - t2:.3f} s "
          f"({free.last_solve_info['iterations']} iterations for the variance block)")
    res[dev] = (w, m, v)
w0, m0, v0 = res[False]; w1, m1, v1 = res[True]
print("device vs host: weights", np.max(np.abs(w1 - w0)) / np.max(np.abs(w0)), "mean", np.max(np.abs(m1 - m0)) / np.max(np.abs(m0)), "var", np.max(np.abs(v1 - v0)) / np.max(np.abs(v0)))
