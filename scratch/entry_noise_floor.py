"""How far does the EXACT posterior of a randomised test problem move when every Gram / cross-covariance entry is re-rounded
in its last bit?  (The device and NumPy evaluate the same kernel formula with different, equally accurate exponentials:
their Gram matrices differ by <= 1 ulp per entry.)  For each seed: the refined posterior of the oracle's matrix against the
refined posterior of the same matrix with entries multiplied by (1 + delta), delta uniform in +-2^-53 (symmetric), five
draws; printed relative to the maximum of the posterior mean.  CPU only; needs no device."""
import dataclasses
import sys

import numpy as np
import scipy.linalg

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
sys.path.insert(0, "linpde-gp_amd")
from oracle import gp as ogp                                   # noqa: E402


def problem(seed):
    """the oracle side of `tests/test_gpu_random.py::_random_problem` (same random stream); the device-side conditioning is
    replaced by a no-op so that this runs without a GPU"""
    import test_gpu_random as t
    import linpde_gp_amd as lp
    lp.GaussianProcess.condition_on_observations = lambda self, *a, **k: self
    return t._random_problem(lp, seed)


def main(seeds):
    for seed in seeds:
        try:
            u, okern, oblocks, mean_const, d, rng = problem(seed)
        except Exception as e:                                   # host objects need the library to be loadable
            print("seed", seed, "skipped:", e); continue
        post = ogp.condition(okern, oblocks, mean_const=mean_const)
        Xt = rng.uniform(-1.0, 1.0, size=(57, d))
        m0, v0 = ogp.refined_posterior(post, Xt)
        e_lap = np.max(np.abs(post.mean(Xt) - m0)) / np.max(np.abs(m0))
        cond = ogp.cond2_estimate(post.G, post.chol)
        shifts = []
        prng = np.random.default_rng(1000 + seed)
        for _ in range(5):
            E = prng.uniform(-2.0**-53, 2.0**-53, size=post.G.shape)
            E = np.tril(E) + np.tril(E, -1).T
            G2 = post.G * (1.0 + E)
            chol2 = scipy.linalg.cholesky(G2, lower=True)
            post2 = dataclasses.replace(post, G=G2, chol=chol2)
            m2, v2 = ogp.refined_posterior(post2, Xt)
            shifts.append(np.max(np.abs(m2 - m0)) / np.max(np.abs(m0)))
        print(f"seed {seed}: n = {post.G.shape[0]:4d}  cond2 >= {cond:.1e}  LAPACK from refined {e_lap:.1e}  "
              f"exact posterior mean moves by {min(shifts):.1e} .. {max(shifts):.1e} under last-bit re-rounding of the entries")


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [107, 102, 113])
