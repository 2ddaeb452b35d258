"""Core clock and socket power (rocm-smi, read-only) while rank-512 / rank-2048 SYRKs on random and on all-zero operands run
back to back: what the chip does under the trailing update's load.  One sample per ~0.3 s from a thread beside the launches."""
import subprocess, sys, threading, time, re
sys.path.insert(0, '.'); sys.path.insert(0, 'linpde-gp_amd')
import numpy as np
from linpde_gp_amd import _engine
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); import _hooks      # test hooks: liblpgp_testhooks.so
ctx = _engine.default_context()
stop = False
samples = []
def sampler():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        s = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out); p = re.search(r"Power \(W\): ([\d.]+)", out)
        samples.append((time.perf_counter(), int(s.group(1)) if s else -1, float(p.group(1)) if p else -1.0))
        time.sleep(0.2)
rng = np.random.default_rng(0)
m = 16384
for label, k, zero in (("random, K = 512", 512, False), ("random, K = 2048", 2048, False), ("zeros,  K = 512", 512, True)):
    A = np.zeros((m, k)) if zero else rng.standard_normal((m, k))
    C = np.zeros((m, m), order="F")
    _hooks.test_gemm(ctx, 0, 0, 1, -1.0, A, A, 1.0, C, k, reps=2)          # warm
    samples.clear(); stop = False
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter()
    _, ms = _hooks.test_gemm(ctx, 0, 0, 1, -1.0, A, A, 1.0, C, k, reps=int(4000 / (2.6 * k / 512)))
    t1 = time.perf_counter()
    stop = True; th.join()
    inside = [(s, p) for (t, s, p) in samples if t0 + 0.5 < t < t1 - 0.3]
    clk = [s for s, _ in inside]; pw = [p for _, p in inside]
    print(f"{label}: {m * (m + 1.0) * k / ms / 1e9:.1f} TFLOP/s over {t1 - t0:.1f} s; rocm-smi sclk {min(clk)}-{max(clk)} MHz (mean {sum(clk)/len(clk):.0f}), "
          f"socket power {min(pw):.0f}-{max(pw):.0f} W (mean {sum(pw)/len(pw):.0f}), {len(inside)} samples", flush=True)
