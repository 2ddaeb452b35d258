#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r5_08; rm -rf $D; mkdir -p $D
timeout 300 python - > $D/first.log 2>&1 <<'PY'
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import problems
from oracle import workloads as owl
for make in (lambda: problems.poisson_1d(512, n_bdry_repeats=16, noise_var=1e-4, m=256), lambda: problems.poisson_2d(n_side=24, n_bdry=24, m_side=12),
             lambda: problems.poisson_2d(n_side=40, n_bdry=37, m_side=17), lambda: problems.poisson_1d(3000, m=300), lambda: problems.poisson_1d()):
    wl = make()
    for lazy in (False, True):
        lp.config.lazy_factorization = lazy
        t0 = time.time()
        u, m, v = problems.condition_and_predict(wl)
        ref = owl.run(wl)
        em = np.max(np.abs(m - ref["mean"])) / np.max(np.abs(ref["mean"])); ev = np.max(np.abs(v - ref["var"])) / np.max(np.abs(ref["var"]))
        print(f"{wl.name} N={wl.n_total} lazy={lazy}: mean {em:.2e} var {ev:.2e} ({time.time()-t0:.1f}s)", flush=True)
PY
echo "first rc=$?"; cat $D/first.log | tail -12
