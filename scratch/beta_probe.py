import sys; sys.path.insert(0,'.'); sys.path.insert(0,'linpde-gp_amd')
import numpy as np
from linpde_gp_amd import _engine
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); import _hooks      # test hooks: liblpgp_testhooks.so
ctx = _engine.default_context()
rng = np.random.default_rng(0)
n = 16384
C = np.zeros((n, n), order="F")
for k in (512, 1024):
    P = rng.standard_normal((n, k))
    for beta in (1.0, 0.0):
        for tri in (1, 0):
            _, ms = _hooks.test_gemm(ctx, 0, 0, tri, -1.0, P, P, beta, C, k, reps=5)
            fl = n * (n + 1.0) * k if tri else 2.0 * n * n * k
            print(f"n={n} k={k} tri={tri} beta={beta}: {ms:.3f} ms {fl/ms/1e9:.1f} TF", flush=True)
