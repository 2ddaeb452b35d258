"""Isolated timing of the refined tile solve (lpgp_test_tile_step), warm."""
import sys, os
import numpy as np, scipy.linalg
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "linpde-gp_amd"))
from linpde_gp_amd import _engine
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); import _hooks      # test hooks: liblpgp_testhooks.so
ctx = _engine.default_context()
rng = np.random.default_rng(0)
L = np.tril(rng.standard_normal((128, 128))) * 0.1 + 4 * np.eye(128)
Linv = scipy.linalg.solve_triangular(L, np.eye(128), lower=True)
for which, ns in ((0, (128, 1024, 4096, 8192, 16896)), (1, (128, 1152, 4224))):
    for n in ns:
        XV = rng.standard_normal((n, 128) if which == 0 else (128, n))
        ts = []
        for rep in range(6):
            _, ms = _hooks.test_tile_step(ctx, which, XV, L, Linv)
            ts.append(ms)
        print(f"which={which} n={n:6d} WGs={n//32:4d}: min {min(ts)*1e3:7.1f} us  median {sorted(ts)[3]*1e3:7.1f} us")
