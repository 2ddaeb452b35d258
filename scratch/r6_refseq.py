"""The reference's own call sequence at c3 in the package's DEFAULT mode -- u = prior.condition_on_observations(...) per block,
u.mean(x), u.std(x)  (experiments/0001_poisson_dirichlet_2d.ipynb cell 22; _conditional.py:44,96-110,193-197,223-231) -- for
rocprofv3 --kernel-trace --stats (profiles/r06_refseq_c3_kernel_stats.csv: the single-vector solve behind u.mean is
trsv_fwd_resident_kernel / trsv_bwd_resident_kernel, one launch each per sequence).
Usage: python3 scratch/r6_refseq.py [reps] [workload]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "linpde-gp_amd"))
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import problems, _engine

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
which = sys.argv[2] if len(sys.argv) > 2 else "c3"
wl = {"c3": problems.poisson_2d, "c2": problems.poisson_1d}[which]()
ctx = _engine.default_context()
lp.config.gram_capacity_hint = wl.n_total
dev = problems.upload(wl)
prior = problems.build_prior(wl)


def sequence():
    u = prior
    for i, o in enumerate(wl.observations):
        Y = o.Y if o.grid is None else o.Y.reshape(tuple(len(f) for f in o.grid))
        b = None if o.noise_var is None else lp.randvars.Normal(np.zeros(Y.shape), np.full(o.X.shape[0], o.noise_var))
        u = u.condition_on_observations(Y, X=dev["obs"][i], L=problems.operator_of(o.op, wl.d), b=b)
    t0 = time.perf_counter()
    m = u.mean(dev["test"])
    t1 = time.perf_counter()
    s = u.std(dev["test"])
    return m, s, t1 - t0


sequence(); ctx.sync()
t0 = time.perf_counter()
tm = []
for _ in range(reps):
    m, s, dtm = sequence()
    tm.append(dtm)
ctx.sync()
dt = (time.perf_counter() - t0) / reps * 1e3
print(f"{which}: N_tot={wl.n_total} default mode (lazy_factorization={lp.config.lazy_factorization}) trsv_resident={ctx.get_option('trsv_resident')}: "
      f"sequence {dt:.2f} ms, of which u.mean(x) (weights solve + cross-covariance + column dots) {np.median(tm) * 1e3:.2f} ms")
