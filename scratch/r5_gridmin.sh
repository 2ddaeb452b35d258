#!/bin/bash
cd "$GRAFT_REPO_ROOT"
cat > /tmp/gm.py <<'PY'
import sys, time, statistics
sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import linpde_gp_amd as lp
from linpde_gp_amd import problems
lp.config.lazy_factorization = True
cases = [("p32 (1024)", problems.poisson_2d(n_side=32, m_side=16)), ("heat (2000)", problems.heat_reference()), ("p48 (2304)", problems.poisson_2d(n_side=48, m_side=24)),
         ("p64 (4096)", problems.poisson_2d(n_side=64, m_side=32)), ("p80 (6400)", problems.poisson_2d(n_side=80, m_side=40)), ("p96 (9216)", problems.poisson_2d(n_side=96, m_side=48))]
ctx = lp._engine.default_context()
def run(wl, thr, n):
    lp.config.grid_assembly_min_points = thr
    dev = problems.upload(wl); prior = problems.build_prior(wl)
    for _ in range(3):
        u, m, v = problems.condition_and_predict(wl, prior=prior, device_arrays=dev); u = None
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(n):
        u, m, v = problems.condition_and_predict(wl, prior=prior, device_arrays=dev); u = None
    ctx.sync(); return (time.perf_counter() - t0) / n * 1e3
for name, wl in cases:
    res = {0: [], 1 << 30: []}
    for rep in range(7):
        for thr in res: res[thr].append(run(wl, thr, 40))
    print(f"{name}: Kronecker median {statistics.median(res[0]):.3f} min {min(res[0]):.3f} | per-entry median {statistics.median(res[1 << 30]):.3f} min {min(res[1 << 30]):.3f}", flush=True)
PY
python3 /tmp/gm.py
