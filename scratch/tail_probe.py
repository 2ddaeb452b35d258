import sys; sys.path.insert(0,'.'); sys.path.insert(0,'linpde-gp_amd')
import numpy as np
from linpde_gp_amd import _engine
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); import _hooks      # test hooks: liblpgp_testhooks.so
ctx = _engine.default_context()
rng = np.random.default_rng(0)
k = 512
nt = 32
B = rng.standard_normal((nt * 128, k))
for mt in (48, 49, 52, 56, 60, 64, 65, 66, 72, 80, 81):
    A = rng.standard_normal((mt * 128, k))
    C = np.zeros((mt * 128, nt * 128), order="F")
    _, ms = _hooks.test_gemm(ctx, 0, 0, 0, -1.0, A, B, 1.0, C, k, reps=8)
    tiles = mt * nt
    print(f"tiles={tiles} rounds={tiles/512:.3f}: {ms*1e3:.1f} us  {2.0*mt*128*nt*128*k/ms/1e9:.1f} TF  per-round {ms*1e3/np.ceil(tiles/512):.1f} us", flush=True)
