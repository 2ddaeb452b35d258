"""A burst of 120 rank-512 SYRKs after 0.6 s of idle, twice, with per-launch HIP-event times printed: how the rate develops
inside the burst.  Run beside scratch/clock_trace to see the shader clock over the same seconds."""
import sys, time, ctypes as C; sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
from linpde_gp_amd import _engine
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); import _hooks      # test hooks: liblpgp_testhooks.so
ctx = _engine.default_context()
rng = np.random.default_rng(0)
m, k = 16384, 512
A = rng.standard_normal((m, k)); Cm = np.zeros((m, m), order="F")
fl = m * (m + 1.0) * k
t00 = time.time()
for burst in range(2):
    time.sleep(0.6)
    for reps in (3, 3, 3, 3, 8, 20, 40, 40):
        _, ms = _hooks.test_gemm(ctx, 0, 0, 1, -1.0, A, A, 1.0, Cm, k, reps=reps)
        print(f"t = {time.time() - t00:6.2f} s: {reps:3d} launches at {fl / ms / 1e9:.1f} TFLOP/s", flush=True)
