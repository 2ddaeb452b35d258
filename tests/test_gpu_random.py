"""Randomised conditioning problems, device vs CPU oracle: kernels (sums of tensor products of half-integer Matern / ExpQuad
factors with random length scales), observation blocks (1-4 per problem, random sizes, identity / first-derivative /
Laplacian-type / heat-type operators, no / scalar / diagonal noise), one- and two-dimensional inputs, prediction of values
and of a derivative read-out.  Fixed seeds: every case is reproducible, and the list is the same on every box.
Bar: random points come close to each other, so these Gram matrices reach cond 1e8 - 1e10, where LAPACK itself is ~1e-8
from the exact posterior of the fp64 matrix (`scratch/random_diag.py`: oracle 1.6e-8, device 1.5e-9 at seed 113; oracle
5.6e-9, device 1.3e-8 at seed 102) and a device-vs-LAPACK comparison at 1e-8 decides nothing.  The yardstick is therefore
the posterior REFINED with long-double residuals (`oracle.gp.refined_posterior`): the device must be within 1e-8 of it
(relative to the maximum), or within 4x the distance LAPACK itself keeps from it, or within 1/20 of cond2(G) * 2^-53 -- the
forward-error scale of ANY backward-stable fp64 solve.  The third term exists because the second is a lottery at the top of
the condition range (`profiles/r03_random_decompose.txt`: at cond 3e8 - 9e9 LAPACK lands between 5e-10 and 1.6e-8 from the
exact posterior of its own matrix, the device between 7e-11 and 1.3e-8 from that of its own; seed 107, cond 8.5e9: LAPACK
2.4e-9, device 1.3e-8 from its own matrix plus 1.3e-8 from entries that differ from NumPy's by two units in the last place
-- every re-rounding of the exponential redraws both numbers).  It moves the bar only for cond2 > 1.8e9.  Besides the
end-to-end distance the test checks its two parts separately: the matrices the device evaluates against NumPy's entry by
entry, and the device's posterior against the exact posterior of ITS OWN matrices (the solver alone)."""
import dataclasses
import os

import numpy as np
import pytest
import scipy.linalg

from conftest import POSTERIOR_RTOL
from oracle import covfuncs as ocf
from oracle import gp as ogp

pytestmark = pytest.mark.gpu


# LPGP_RANDOM_SCALE=k multiplies the block sizes (3..159 points in the suite) for one-off surveys of the multi-panel paths
_SIZE_SCALE = int(os.environ.get("LPGP_RANDOM_SCALE", "1"))


def _random_problem(lp, seed):
    from linpde_gp_amd.linfuncops import diffops
    cf = lp.randprocs.covfuncs
    rng = np.random.default_rng(seed)
    d = int(rng.integers(1, 3))
    ngroups = int(rng.integers(1, 3))
    k = None
    okern = []
    for _ in range(ngroups):
        scale = float(rng.uniform(0.5, 3.0))
        facs, ofacs = [], []
        for _dim in range(d):
            ell = float(rng.uniform(0.4, 1.5))
            if rng.uniform() < 0.25:
                facs.append(cf.ExpQuad((), lengthscales=ell)); ofacs.append(("expquad", ell))
            else:
                nu = float(rng.choice([2.5, 3.5, 4.5]))          # twice differentiable under both arguments
                facs.append(cf.Matern((), nu=nu, lengthscales=ell)); ofacs.append(("matern", nu, ell))
        g = scale * (cf.TensorProduct(*facs) if d > 1 else facs[0])
        k = g if k is None else k + g
        okern.append((scale, ofacs))
    shape = (d,) if d > 1 else ()
    mean_const = float(rng.uniform(-1.0, 1.0))
    prior = lp.GaussianProcess(lp.functions.Constant(shape, mean_const), k)

    def operator():
        kind = rng.integers(0, 4)
        if kind == 0:
            return None
        if kind == 1:
            mi = [0] * d; mi[int(rng.integers(0, d))] = 1
            return float(rng.uniform(0.5, 2.0)) * diffops.PartialDerivative(diffops.MultiIndex(tuple(mi) if d > 1 else 1))
        if kind == 2:
            return float(rng.uniform(-2.0, -0.5)) * diffops.Laplacian(shape) + float(rng.uniform(0.0, 1.0)) * lp.linfuncops.Identity(shape)
        if d == 2:
            return diffops.HeatOperator((2,), alpha=float(rng.uniform(0.05, 0.5)))
        return -1.0 * diffops.Laplacian(shape)

    u, oblocks = prior, []
    for _b in range(int(rng.integers(1, 5))):
        n = int(rng.integers(3, 160)) * _SIZE_SCALE
        X = rng.uniform(-1.0, 1.0, size=(n, d))
        Y = rng.standard_normal(n)
        L = operator()
        coeffs = {(0,) * d: 1.0} if L is None else L.coefficients_dict()
        noise_kind = rng.integers(0, 3)
        if noise_kind == 0:                       # random points may nearly coincide: a small nugget keeps G factorable
            nv = 1e-6
            b = lp.randvars.Normal(np.zeros(n), nv * np.eye(n)); nm = np.zeros(n)
        elif noise_kind == 1:
            nv = float(rng.uniform(1e-4, 1e-2))
            nm = rng.standard_normal(n) * 0.1
            b = lp.randvars.Normal(nm, nv * np.eye(n))
        else:
            nv = rng.uniform(1e-4, 1e-2, size=n)
            nm = np.zeros(n)
            b = lp.randvars.Normal(nm, np.diag(nv))
        Xarg = X if d > 1 else X[:, 0]
        u = u.condition_on_observations(Y, Xarg, L=L, b=b)
        oblocks.append(ogp.ObsBlock(X, coeffs, Y, nm, nv))
    return u, okern, oblocks, mean_const, d, rng


def device_matrices(u, oblocks, Xt, d, Ltest=None):
    """The Gram matrix and the prediction cross-covariance AS THE DEVICE EVALUATES THEM (`CovarianceFunction.matrix` of the
    differentiated kernels: the same kernels, the same bits as the block assembly), with the oracle's noise added."""
    from linpde_gp_amd.problems import operator_of
    k = u.prior.cov
    ops = [operator_of(b.L, d) for b in oblocks]
    pts = [b.X if d > 1 else b.X[:, 0] for b in oblocks]
    Xt_ = Xt if d > 1 else Xt[:, 0]

    def lkl(Li, Lj, X0, X1):
        kk = k if Lj is None else Lj(k, argnum=1)
        kk = kk if Li is None else Li(kk, argnum=0)
        return np.asarray(kk.matrix(X0, X1))

    G = np.block([[lkl(ops[i], ops[j], pts[i], pts[j]) for j in range(len(oblocks))] for i in range(len(oblocks))])
    G = np.tril(G) + np.tril(G, -1).T                    # the factorisation reads the lower triangle
    off = 0
    for b in oblocks:
        nc = ogp._noise_cov_dense(b)
        if nc is not None:
            G[off:off + b.n, off:off + b.n] += nc
        off += b.n
    Lt = None if Ltest is None else operator_of(Ltest, d)
    K = np.concatenate([lkl(Lt, ops[j], Xt_, pts[j]) for j in range(len(oblocks))], axis=1)
    return G, K


# which of the three terms of the bar each check of each seed NEEDED (1: the plain 1e-8 criterion; 2: 4 x LAPACK's own distance;
# 3: the forward-error scale 0.05 cond2 2^-53), filled by the test below and judged by test_zz_bound_usage at the end of the module
_BOUND_USAGE = {}
_current_seed = [None]


def _assert_as_good_as_lapack(what, dev, lapack, exact, cond2=0.0):
    scale = float(np.max(np.abs(exact)))
    e_dev, e_lap = float(np.max(np.abs(dev - exact))), float(np.max(np.abs(lapack - exact)))
    terms = (POSTERIOR_RTOL * scale, 4.0 * e_lap, 0.05 * cond2 * 2.0**-53 * scale)
    bound = max(terms)
    needed = 1 if e_dev <= terms[0] else (2 if e_dev <= terms[1] else 3)
    rec = _BOUND_USAGE.setdefault(_current_seed[0], {"needed": 1, "worst": 0.0, "cond2": cond2, "what": ""})
    if needed > rec["needed"] or (needed == rec["needed"] and e_dev / scale > rec["worst"]):
        rec.update(needed=max(needed, rec["needed"]), worst=max(e_dev / scale, rec["worst"]), what=what)
    assert e_dev <= bound, (f"{what}: device {e_dev / scale:.2e} from the refined posterior, LAPACK {e_lap / scale:.2e} "
                            f"(bound {bound / scale:.2e}, relative to the maximum; cond2 >= {cond2:.1e})")
    # the third term may be the binding one only where it exceeds the plain criterion at all (cond2 > 1.8e9)
    assert needed < 3 or cond2 > 1.8e9, f"{what}: needed the cond2 term at cond2 = {cond2:.1e}"


ENTRY_RTOL = 4e-15          # device-evaluated matrices vs NumPy's, relative to the largest entry (measured <= 1e-15)


def _seed_range():
    """Seeds 100..147 in the suite; LPGP_RANDOM_SEEDS="lo:hi" runs another range of the same generator (one-off surveys)."""
    spec = os.environ.get("LPGP_RANDOM_SEEDS")
    if not spec:
        # + four seeds of the round-4 survey of 1000..1399 on which the device was 9-37 x LAPACK's distance from the exact
        # posterior until the panel step INSIDE the tile Cholesky was refined (csrc/potrf.hip, phase B; MEASUREMENTS.md)
        return list(range(100, 148)) + [1088, 1215, 1226, 1388]
    lo, hi = (int(v) for v in spec.split(":"))
    return range(lo, hi)


@pytest.mark.parametrize("seed", _seed_range())
def test_random_problem_matches_oracle(seed):
    import linpde_gp_amd as lp
    from linpde_gp_amd.linfuncops import diffops
    _current_seed[0] = seed
    u, okern, oblocks, mean_const, d, rng = _random_problem(lp, seed)
    post = ogp.condition(okern, oblocks, mean_const=mean_const)
    Xt = rng.uniform(-1.0, 1.0, size=(57, d))
    mean, var = u.predict(Xt if d > 1 else Xt[:, 0])
    m_exact, v_exact = ogp.refined_posterior(post, Xt)
    cond2 = ogp.cond2_estimate(post.G, post.chol)
    _assert_as_good_as_lapack("mean", mean, post.mean(Xt), m_exact, cond2)
    _assert_as_good_as_lapack("variance", var, post.var(Xt), v_exact, cond2)
    # the same problem through the opt-in throughput mode -- deferred factorisation, the prediction riding inside it, resident
    # panel chain -- against the same exact posterior with the same bar (round 5)
    saved = lp.config.lazy_factorization
    lp.config.lazy_factorization = True
    try:
        u_f = _random_problem(lp, seed)[0]
        mean_f, var_f = u_f.predict(Xt if d > 1 else Xt[:, 0])
        assert u_f._state.deferred is False and u_f._state.pending is False
    finally:
        lp.config.lazy_factorization = saved
    _assert_as_good_as_lapack("mean (fused pipeline)", mean_f, post.mean(Xt), m_exact, cond2)
    _assert_as_good_as_lapack("variance (fused pipeline)", var_f, post.var(Xt), v_exact, cond2)
    del u_f
    # the two parts of that distance: (entries) the matrices as the device evaluates them ...
    G_dev, K_dev = device_matrices(u, oblocks, Xt, d)
    K_np = ogp.cross_cov(okern, oblocks, Xt)
    assert np.max(np.abs(G_dev - post.G)) <= ENTRY_RTOL * np.max(np.abs(post.G))
    assert np.max(np.abs(K_dev - K_np)) <= ENTRY_RTOL * np.max(np.abs(K_np))
    # ... and (solver) the device's posterior against the exact posterior of its own matrices
    post_dev = dataclasses.replace(post, G=G_dev, chol=scipy.linalg.cholesky(G_dev, lower=True))
    m_own, v_own = ogp.refined_posterior(post_dev, Xt, K=K_dev)
    _assert_as_good_as_lapack("mean vs the exact posterior of the device's own matrices", mean, post.mean(Xt) - m_exact + m_own, m_own, cond2)
    _assert_as_good_as_lapack("variance vs the exact posterior of the device's own matrices", var, post.var(Xt) - v_exact + v_own, v_own, cond2)
    # a derivative read-out of the posterior (`LinearFunctionOperator(ConditionalGaussianProcess)`, _conditional.py:432-450)
    mi = [0] * d; mi[seed % d] = 1
    Du = diffops.PartialDerivative(diffops.MultiIndex(tuple(mi) if d > 1 else 1))(u)
    dm, dv = Du.predict(Xt if d > 1 else Xt[:, 0])
    Ltest = {tuple(mi): 1.0}
    dm_exact, dv_exact = ogp.refined_posterior(post, Xt, Ltest)
    _assert_as_good_as_lapack("derivative mean", dm, post.mean(Xt, Ltest), dm_exact, cond2)
    _assert_as_good_as_lapack("derivative variance", dv, post.var(Xt, Ltest), dv_exact, cond2)


def test_zz_bound_usage():
    """VERDICT r3 / ADVICE r3: the bar of the test above is the max of three terms; this reports, per seed, which one was
    NEEDED, and fails the module if more than 10 % of the seeds needed the third (so that an accuracy regression at high
    condition cannot hide behind it).  The report reaches the driver's log as a pytest warning (shown with -q too)."""
    import warnings
    if len(_BOUND_USAGE) < 40:
        pytest.skip("the randomised problems did not all run in this session")
    by = {k: sorted(s_ for s_, r in _BOUND_USAGE.items() if r["needed"] == k) for k in (1, 2, 3)}
    worst = max(_BOUND_USAGE.items(), key=lambda kv: kv[1]["worst"])
    warnings.warn(f"random problems: {len(by[1])} seeds inside the plain 1e-8 criterion, {len(by[2])} needed 4 x LAPACK's own distance "
                  f"{by[2]}, {len(by[3])} needed the cond2 term {by[3]}; worst: seed {worst[0]} at {worst[1]['worst']:.2e} "
                  f"({worst[1]['what']}, cond2 >= {worst[1]['cond2']:.1e})")
    assert len(by[3]) <= 0.10 * len(_BOUND_USAGE), by[3]
