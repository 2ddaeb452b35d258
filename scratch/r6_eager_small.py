"""Where the DEFAULT (eager) mode's time goes on notebook-size problems (heat_reference: N_tot 2 105, c1: 544): per-call wall
times of the reference's sequence and a cProfile of the host side.   Usage: python3 scratch/r6_eager_small.py [workload] [reps]"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "linpde-gp_amd"))
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import problems, _engine

which = sys.argv[1] if len(sys.argv) > 1 else "heat_reference"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
wl = {"heat_reference": problems.heat_reference, "c1": lambda: problems.poisson_1d(512, n_bdry_repeats=16, noise_var=1e-4, m=256),
      "c2": problems.poisson_1d}[which]()
ctx = _engine.default_context()
lp.config.gram_capacity_hint = wl.n_total
dev = problems.upload(wl)
prior = problems.build_prior(wl)
ops = [problems.operator_of(o.op, wl.d) for o in wl.observations]


def sequence(stamps=None):
    u = prior
    for i, o in enumerate(wl.observations):
        Y = o.Y if o.grid is None else o.Y.reshape(tuple(len(f) for f in o.grid))
        b = None if o.noise_var is None else lp.randvars.Normal(np.zeros(Y.shape), np.full(o.X.shape[0], o.noise_var))
        t0 = time.perf_counter()
        u = u.condition_on_observations(Y, X=dev["obs"][i], L=ops[i], b=b)
        if stamps is not None: stamps.append(("cond%d(n=%d)" % (i, o.X.shape[0]), time.perf_counter() - t0))
    t0 = time.perf_counter(); m = u.mean(dev["test"])
    if stamps is not None: stamps.append(("mean", time.perf_counter() - t0))
    t0 = time.perf_counter(); s = u.std(dev["test"])
    if stamps is not None: stamps.append(("std", time.perf_counter() - t0))
    return m, s


for lazy in (False, True):
    lp.config.lazy_factorization = lazy
    for _ in range(5): sequence()
    ctx.sync()
    acc = {}
    t0 = time.perf_counter()
    for _ in range(reps):
        st = []
        sequence(st)
        for k, v in st: acc.setdefault(k, []).append(v)
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps * 1e3
    print(f"{which} N_tot={wl.n_total} lazy={lazy}: sequence {dt:.3f} ms; per call (median, us): " + ", ".join(f"{k} {np.median(v) * 1e6:.0f}" for k, v in acc.items()))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(reps): sequence()
    ctx.sync(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
    print("\n".join(l[:170] for l in s.getvalue().splitlines()[4:34]))
