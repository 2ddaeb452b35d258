"""Isolated timing of the fused panel solve (lpgp_test_panel_solve), warm."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "linpde-gp_amd"))
from linpde_gp_amd import _engine
ctx = _engine.default_context()
rng = np.random.default_rng(0)
for nt in (1, 2, 4):
    n = nt * 128
    L = np.tril(rng.standard_normal((n, n))) * 0.05 + 4 * np.eye(n)
    for cols in (128, 1152, 4224, 8448, 16512):
        V = rng.standard_normal((n, cols))
        ts = [_engine.test_panel_solve(ctx, V, L)[1] for _ in range(6)]
        print(f"NT={nt} cols={cols:6d} WGs={'%4d' % (cols // 16 if cols // 16 <= 512 else cols // 32)}: min {min(ts)*1e3:7.1f} us")
