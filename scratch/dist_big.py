# ad-hoc: a larger multi-rank job on ONE GPU through the host-staged transport (see tests/test_gpu_dist.py)
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import test_gpu_dist as t
world = int(sys.argv[1]) if len(sys.argv) > 1 else 3
t._run_ranks(world, "poisson_2d(n_side=72, n_bdry=50, m_side=20)", 512, 29801)
print("ok", world)
