"""Maximum size on one device ("sized for 288 GB"): a Gram matrix of 137 GB -- 131 072 noisy values at scattered
points, one block, every entry through the per-entry assembly kernel -- conditioned on and predicted from through the host API,
checked by size-independent properties (the oracle would need 137 GB of host memory and hours): the representer weights solve the
system, with K w formed MATRIX-FREE by the kernel-product kernel, which shares nothing with the assembled matrix or its factor.
Named zz: last in the run, after everything else has released its device memory."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_half_the_hbm_in_one_gram_matrix():
    import linpde_gp_amd as lp
    from linpde_gp_amd import problems
    ctx = lp._engine.default_context()
    hbm = ctx.device_info()["hbm_bytes"]
    n, m = 131072, 4096
    if hbm and hbm < 200e9:
        pytest.skip(f"device has {hbm / 1e9:.0f} GB of HBM: the 137-GB Gram matrix of this test needs an MI355X")
    wl = problems.scattered_2d(n=n, m=m, noise_var=1e-2, seed=1)
    prior = problems.build_prior(wl)
    o = wl.observations[0]
    u = prior.condition_on_observations(o.Y, X=o.X, b=lp.randvars.Normal(np.zeros(n), o.noise_var))
    mean, var = u.predict(wl.Xtest)
    w = u.representer_weights
    Kw = prior.cov.linop(o.X, o.X) @ w
    res = np.linalg.norm(Kw + o.noise_var * w - o.Y) / np.linalg.norm(o.Y)
    assert res < 1e-10, f"|| (K + s^2 I) w - r || / || r || = {res:.2e}"
    prior_var = float(np.ravel(prior.cov(wl.Xtest[:1], wl.Xtest[:1]))[0])
    assert var.min() >= 0.0 and var.max() <= prior_var
    # the posterior mean is K(x, X) w: at training points it equals the matrix-free product
    idx = np.arange(0, n, n // 256)
    assert np.abs(u.mean(o.X[idx]) - Kw[idx]).max() <= 1e-9 * np.abs(Kw).max()
    # and at the prediction points it agrees with the mean the fused prediction returned (two code paths: weights / residual solve)
    assert np.abs(u.mean(wl.Xtest[:256]) - mean[:256]).max() <= 1e-8 * np.abs(mean).max()
