"""Where the distance between the device's posterior and the oracle's refined posterior comes from, per seed of
tests/test_gpu_random.py: (entries) the exact posterior of the device-evaluated Gram / cross-covariance matrices against the
exact posterior of NumPy's -- both refined with long-double residuals, no device solver involved -- and (solver) the device's
posterior against the exact posterior of its own matrices.  Usage: python scratch/random_decompose.py [seeds...]"""
import dataclasses
import sys

import numpy as np
import scipy.linalg

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
sys.path.insert(0, "linpde-gp_amd")
import linpde_gp_amd as lp                                     # noqa: E402
import test_gpu_random as t                                    # noqa: E402
from oracle import gp as ogp                                   # noqa: E402


def main(seeds):
    print("seed    n   cond2   | LAPACK  | device total | = entries effect + solver | max |G_dev - G_np| / max|G|   (all relative to max |mean|)")
    for seed in seeds:
        u, okern, oblocks, mean_const, d, rng = t._random_problem(lp, seed)
        post = ogp.condition(okern, oblocks, mean_const=mean_const)
        Xt = rng.uniform(-1.0, 1.0, size=(57, d))
        mean, var = u.predict(Xt if d > 1 else Xt[:, 0])
        m_np, v_np = ogp.refined_posterior(post, Xt)
        G_dev, K_dev = t.device_matrices(u, oblocks, Xt, d)
        post_dev = dataclasses.replace(post, G=G_dev, chol=scipy.linalg.cholesky(G_dev, lower=True))
        m_dev, v_dev = ogp.refined_posterior(post_dev, Xt, K=K_dev)
        sc = np.max(np.abs(m_np))
        K_np = ogp.cross_cov(okern, oblocks, Xt)
        print(f"{seed}  {post.G.shape[0]:4d}  {ogp.cond2_estimate(post.G, post.chol):.1e} | {np.max(np.abs(post.mean(Xt) - m_np)) / sc:.1e} | "
              f"{np.max(np.abs(mean - m_np)) / sc:.1e}     | {np.max(np.abs(m_dev - m_np)) / sc:.1e} + {np.max(np.abs(mean - m_dev)) / sc:.1e}       | "
              f"G {np.max(np.abs(G_dev - post.G)) / np.max(np.abs(post.G)):.1e}  K {np.max(np.abs(K_dev - K_np)) / np.max(np.abs(K_np)):.1e}"
              f"   var: total {np.max(np.abs(var - v_np)) / np.max(np.abs(v_np)):.1e} = {np.max(np.abs(v_dev - v_np)) / np.max(np.abs(v_np)):.1e} + {np.max(np.abs(var - v_dev)) / np.max(np.abs(v_np)):.1e}")


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or list(range(100, 124)))
