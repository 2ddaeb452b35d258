"""Body of tests/test_gpu_spawn.py, run as its OWN process (the front must start its workers before the calling process
makes any GPU call; the pytest process has long done so).  A reference-style script: ONE process, plain
`prior.condition_on_observations(...)` calls -- the factor lives sharded over the workers' ranks."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "linpde-gp_amd"))

import numpy as np  # noqa: E402

import linpde_gp_amd as lp  # noqa: E402
from linpde_gp_amd import problems  # noqa: E402
from oracle import workloads as owl  # noqa: E402


def main():
    transport = sys.argv[1] if len(sys.argv) > 1 else "rccl"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    group = lp.spawn(n, devices=[0] * n, transport=transport, rccl_loopback=(transport == "rccl"))
    assert group.world == n and tuple(group.info["grid"]) == (n, 1), group.info
    wl = problems.poisson_2d(n_side=34, n_bdry=30, m_side=9)          # N_tot = 1276: three panels of 512, ragged blocks
    prior = problems.build_prior(wl)
    u = prior
    for o in wl.observations:
        X, Y = o.X_as_given()
        b = None if o.noise_var is None else lp.randvars.Normal(np.zeros(Y.shape), np.full(o.X.shape[0], o.noise_var))
        u = u.condition_on_observations(Y, X=lp.to_device(X), L=problems.operator_of(o.op, wl.d), b=b)
    assert type(u).__name__ == "RemoteConditionalGaussianProcess"
    mean, var = u.predict(wl.Xtest)
    ref = owl.run(wl)
    em = np.max(np.abs(mean - ref["mean"])) / np.max(np.abs(ref["mean"]))
    ev = np.max(np.abs(var - ref["var"])) / np.max(np.abs(ref["var"]))
    assert em <= 1e-8 and ev <= 1e-8, (em, ev)                          # the criterion of tests/conftest.py
    np.testing.assert_allclose(u.mean(wl.Xtest), ref["mean"], rtol=0, atol=1e-8 * np.max(np.abs(ref["mean"])))
    np.testing.assert_allclose(u.std(wl.Xtest[:5]) ** 2, ref["var"][:5], rtol=0, atol=1e-8 * np.max(np.abs(ref["var"])))
    np.testing.assert_allclose(u.representer_weights, ref["weights"], rtol=0, atol=1e-6 * np.max(np.abs(ref["weights"])))
    C = u.cov.matrix(wl.Xtest[:6])
    np.testing.assert_allclose(np.diag(C), ref["var"][:6], rtol=0, atol=1e-8 * np.max(np.abs(ref["var"])))
    nrm = u(wl.Xtest[:6])
    np.testing.assert_allclose(nrm.mean, ref["mean"][:6], rtol=0, atol=1e-8 * np.max(np.abs(ref["mean"])))
    # the reference's validation errors surface in the calling process, and the posterior it was called on stays usable
    try:
        u.condition_on_observations(np.zeros(4), np.zeros((3, 2)))
        raise AssertionError("shape mismatch accepted")
    except ValueError:
        pass
    Xbad = np.array([[0.2, 0.1], [0.2, 0.1], [0.5, 0.3]])
    try:
        u.condition_on_observations(np.zeros(3), Xbad, b=lp.randvars.Normal(np.zeros(3), -1e-3 * np.eye(3)))
        raise AssertionError("non-PD block accepted")
    except np.linalg.LinAlgError:
        pass
    m2 = u.mean(wl.Xtest)
    np.testing.assert_allclose(m2, ref["mean"], rtol=0, atol=1e-8 * np.max(np.abs(ref["mean"])))
    del u
    group.close()
    print(f"SPAWN-OK transport={transport} ranks={n} mean {em:.2e} var {ev:.2e}")


if __name__ == "__main__":
    main()
