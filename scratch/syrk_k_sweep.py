import sys; sys.path.insert(0,'.'); sys.path.insert(0,'linpde-gp_amd')
import numpy as np
from linpde_gp_amd import _engine
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); import _hooks      # test hooks: liblpgp_testhooks.so
ctx = _engine.default_context()
rng = np.random.default_rng(0)
for n in (16384, 12288, 8192, 6144, 4096):
    C = np.zeros((n, n), order="F")
    for k in (512, 1024, 2048):
        P = rng.standard_normal((n, k))
        _, ms = _hooks.test_gemm(ctx, 0, 0, 1, -1.0, P, P, 1.0, C, k, reps=5)
        fl = n * (n + 1.0) * k
        tiles = (n // 128) * (n // 128 + 1) // 2
        print(f"SYRK n={n} k={k}: {ms:.3f} ms {fl/ms/1e9:.1f} TF  tiles={tiles} rounds={tiles/512:.2f}", flush=True)
