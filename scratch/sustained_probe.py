"""Steady-state rate of the rank-K SYRK (16384^2, random operands): 300 back-to-back launches between two events, after the
clock has recovered from the onset of load (the first ~40 ms of a burst run at 1.7-1.9 GHz: scratch/burst_clock.py,
scratch/clock_trace.hip).  LPGP_TEST_GEMM_STREAM=1: on the CU-masked update stream (LPGP_RESERVE_CUS)."""
import os, sys; sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
from linpde_gp_amd import _engine
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); import _hooks      # test hooks: liblpgp_testhooks.so
ctx = _engine.default_context()
rng = np.random.default_rng(0)
m = 16384
C = np.zeros((m, m), order="F")
out = []
for k in (512, 1024, 2048):
    A = rng.standard_normal((m, k))
    _, ms = _hooks.test_gemm(ctx, 0, 0, 1, -1.0, A, A, 1.0, C, k, reps=max(60, int(300 * 512 / k)))
    out.append(f"K = {k}: {ms:.3f} ms = {m * (m + 1.0) * k / ms / 1e9:.1f} TFLOP/s")
print(f"stream {os.environ.get('LPGP_TEST_GEMM_STREAM', '0')} reserve {os.environ.get('LPGP_RESERVE_CUS', '8')}: " + ";  ".join(out), flush=True)
