"""The same rank-512 SYRK (16384^2, random operands) timed over 1, 2, 5, 20, 100, 400 back-to-back launches between two events:
does the rate depend on how long the burst is?  (and on an idle pause in front of it)"""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
from linpde_gp_amd import _engine
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); import _hooks      # test hooks: liblpgp_testhooks.so
ctx = _engine.default_context()
rng = np.random.default_rng(0)
m, k = 16384, 512
A = rng.standard_normal((m, k)); C = np.zeros((m, m), order="F")
fl = m * (m + 1.0) * k
for pause in (0.0, 0.5):
    for reps in (1, 2, 5, 20, 100, 400, 5, 1):
        time.sleep(pause)
        _, ms = _hooks.test_gemm(ctx, 0, 0, 1, -1.0, A, A, 1.0, C, k, reps=reps)
        print(f"pause {pause:.1f} s, {reps:4d} launches: {ms:.3f} ms each -> {fl / ms / 1e9:.1f} TFLOP/s", flush=True)
