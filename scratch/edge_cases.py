import sys, traceback
sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd.linfuncops import diffops
cf = lp.randprocs.covfuncs
prior = lp.GaussianProcess(lp.functions.Zero((2,)), cf.TensorProduct(cf.Matern((), nu=2.5), cf.Matern((), nu=2.5)))
rng = np.random.default_rng(0)
def attempt(name, fn):
    try:
        r = fn()
        print(name, "->", "ok", getattr(r, "shape", r if not isinstance(r, tuple) else tuple(getattr(x, "shape", x) for x in r)))
    except Exception as e:
        print(name, "->", type(e).__name__, str(e)[:150])
X = rng.uniform(-1, 1, (5, 2)); Y = rng.standard_normal(5)
u = prior.condition_on_observations(Y, X=X)
attempt("predict 0 points", lambda: u.predict(np.zeros((0, 2))))
attempt("predict 1 point", lambda: u.predict(np.zeros((1, 2))))
attempt("mean scalar point", lambda: u.mean(np.zeros(2)))
attempt("var batch (3,4)", lambda: u.var(rng.uniform(-1, 1, (3, 4, 2))))
attempt("cov 0x0", lambda: u.cov.matrix(np.zeros((0, 2))))
attempt("condition on 0 observations", lambda: prior.condition_on_observations(np.zeros(0), X=np.zeros((0, 2))).predict(X))
attempt("append 0 observations", lambda: u.condition_on_observations(np.zeros(0), X=np.zeros((0, 2))).predict(X))
attempt("single observation", lambda: prior.condition_on_observations(np.array([1.0]), X=np.zeros((1, 2))).predict(X))
attempt("duplicate points, no noise", lambda: prior.condition_on_observations(np.ones(2), X=np.zeros((2, 2))).predict(X))
attempt("duplicate points with noise", lambda: prior.condition_on_observations(np.ones(2), X=np.zeros((2, 2)), b=lp.randvars.Normal(np.zeros(2), 1e-4 * np.eye(2))).predict(X))
attempt("kernel matrix 0 x 3", lambda: prior.cov.matrix(np.zeros((0, 2)), X[:3]))
attempt("kernel call broadcasting", lambda: prior.cov(X[:, None], X[None, :3]))
attempt("nan input", lambda: prior.condition_on_observations(np.array([np.nan]), X=np.zeros((1, 2))).predict(X))
attempt("129 obs (tile + 1)", lambda: prior.condition_on_observations(rng.standard_normal(129), X=rng.uniform(-1, 1, (129, 2)), b=lp.randvars.Normal(np.zeros(129), 1e-6 * np.eye(129))).predict(X))
