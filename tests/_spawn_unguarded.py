# A reference-style script WITHOUT an `if __name__ == "__main__":` guard: module-level statements only.
# argv[1] = "call": lp.spawn(2, ...) explicitly; "env": nothing but LPGP_SPAWN=2 in the environment, picked up at the
# first conditioning.  Two workers share GPU 0 through the direct-peer transport.
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "linpde-gp_amd"))

import numpy as np  # noqa: E402

import linpde_gp_amd as lp  # noqa: E402
from linpde_gp_amd import problems  # noqa: E402
from oracle import workloads as owl  # noqa: E402

mode = sys.argv[1]
if mode == "call":
    lp.spawn(2, devices=[0, 0], transport="ipc")
wl = problems.poisson_2d(n_side=20, n_bdry=16, m_side=7)
u = problems.build_prior(wl)
for o in wl.observations:
    X, Y = o.X_as_given()
    b = None if o.noise_var is None else lp.randvars.Normal(np.zeros(Y.shape), np.full(o.X.shape[0], o.noise_var))
    u = u.condition_on_observations(Y, X=np.asarray(X), L=problems.operator_of(o.op, wl.d), b=b)
assert type(u).__name__ == "RemoteConditionalGaussianProcess", type(u).__name__
mean, var = u.predict(wl.Xtest)
ref = owl.run(wl)
em = np.max(np.abs(mean - ref["mean"])) / np.max(np.abs(ref["mean"]))
ev = np.max(np.abs(var - ref["var"])) / np.max(np.abs(ref["var"]))
assert em <= 1e-8 and ev <= 1e-8, (em, ev)
print(f"SPAWN-OK unguarded mode={mode} mean {em:.2e} var {ev:.2e}")
