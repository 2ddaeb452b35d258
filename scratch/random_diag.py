# For the random problems of tests/test_gpu_random.py: who is closer to the exact posterior mean of the fp64 Gram matrix --
# the oracle (LAPACK) or the device?  "Exact": representer weights refined with long-double residuals until they stop moving.
import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "linpde-gp_amd"); sys.path.insert(0, "tests")
import numpy as np, scipy.linalg
import linpde_gp_amd as lp
from oracle import gp as ogp
import test_gpu_random as T

for seed in [int(a) for a in sys.argv[1:]] or range(100, 124):
    u, okern, oblocks, mean_const, d, rng = T._random_problem(lp, seed)
    post = ogp.condition(okern, oblocks, mean_const=mean_const)
    Xt = rng.uniform(-1.0, 1.0, size=(57, d))
    mean, var = u.predict(Xt if d > 1 else Xt[:, 0])
    G, chol = post.G, post.chol
    r = ogp.residual(oblocks, mean_const)
    w = post.weights.astype(np.longdouble)
    Gl = G.astype(np.longdouble)
    for it in range(12):
        res = (r.astype(np.longdouble) - Gl @ w).astype(np.double)
        dw = scipy.linalg.cho_solve((chol, True), res)
        w = w + dw
        if np.max(np.abs(dw)) < 1e-17 * np.max(np.abs(w)): break
    K = ogp.cross_cov(okern, oblocks, Xt)
    m_true = np.asarray(mean_const + (K.astype(np.longdouble) @ w), dtype=np.double)
    sc = np.max(np.abs(m_true))
    cond = ogp.cond2_estimate(G, chol)
    print(f"seed {seed} d={d} N={G.shape[0]} cond~{cond:.1e}: oracle-vs-exact {np.max(np.abs(post.mean(Xt) - m_true)) / sc:.2e}  "
          f"device-vs-exact {np.max(np.abs(mean - m_true)) / sc:.2e}  device-vs-oracle {np.max(np.abs(mean - post.mean(Xt))) / sc:.2e}  "
          f"weights: oracle {np.max(np.abs(post.weights - w)) / np.max(np.abs(w)):.1e} device {np.max(np.abs(u.representer_weights - w)) / np.max(np.abs(w)):.1e}", flush=True)
