"""Distributed code path (csrc/dist.hip: Pr x Pc process grid, 2-D block-cyclic tiles, sharded factor) on ONE GPU.

* world_size 1 through RCCL (ncclCommInitRank; a 1 x 1 grid runs the distributed factorisation, the panel gather
  and the panel-streaming solves without a peer) must reproduce the single-GPU result;
* 2, 3 and 4 ranks -- grids 2 x 1, 1 x 2, 3 x 1 and 2 x 2 -- as separate processes SHARING the GPU: RCCL refuses two
  ranks on one device, so these jobs use the host-staged test transport (`lpgp_dist_init_host`: every message goes
  D2H -> control plane -> H2D); everything else is the product path: sharded storage and assembly (per-entry and
  tensor-grid kernels), diagonal-block broadcast, Pr-fold parallel panel solve, panel gather, staircase update of the
  local tiles with look-ahead, block append at boundaries inside a block, streamed solves, prediction sharded over
  the ranks, gathered results.  Every rank must match the oracle, and the factor collected from all ranks must
  reproduce the oracle's Gram matrix.
The 8-GPU node with RCCL over xGMI is only available to the driver; a NumPy mirror of the algorithm also runs on
gloo ranks in tests/test_dist_cpu.py."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "linpde-gp_amd"))
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import _dist, _engine, problems
from oracle import workloads as owl
os.environ["LPGP_FORCE_RCCL"] = "1"
comm = _dist.Comm(0, 1)
ctx = _engine.default_context()
ctx.dist_init(comm)
assert ctx.world == 1 and ctx.comm is comm and ctx.grid == (1, 1)
wl = problems.poisson_2d(n_side=40, n_bdry=33, m_side=9)      # ragged sizes, 3 panels of 512
u, mean, var = problems.condition_and_predict(wl)
ref = owl.run(wl)
em = np.max(np.abs(mean - ref["mean"])) / np.max(np.abs(ref["mean"]))
ev = np.max(np.abs(var - ref["var"])) / np.max(np.abs(ref["var"]))
w = u.representer_weights                                      # streamed forward + backward solve
ew = np.max(np.abs(w - ref["weights"])) / np.max(np.abs(ref["weights"]))
B = np.random.default_rng(1).normal(size=(wl.n_total, 3))
from oracle import gp as ogp
G_ref = ogp.gram(wl.kernel, owl.blocks_of(wl))
es = np.max(np.abs(G_ref @ u.gram.solve(B) - B))
print("DIST1", em, ev, ew, es)
assert em < 1e-8 and ev < 1e-8 and ew < 1e-6 and es < 1e-4      # (residual of a backward-stable solve: eps |G| |x|, |x| up to cond |B| / |G|)
"""


def test_world1_rccl_path_matches_oracle():
    out = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "DIST1" in out.stdout


MULTI = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "linpde-gp_amd"))
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import _dist, _engine, problems
from oracle import workloads as owl
comm = _dist.Comm.from_env()
ctx = _engine.default_context()            # LPGP_DEVICE=0 on every rank: the ranks share the GPU
ctx.set_option("nb", %(nb)d)
ctx.dist_init(comm, transport="host", grid=%(grid)r)
assert ctx.world == comm.world and ctx.rank == comm.rank and ctx.grid == %(grid)r
wl = problems.%(workload)s
u, mean, var = problems.condition_and_predict(wl)
ref = owl.run(wl)
em = np.max(np.abs(mean - ref["mean"])) / np.max(np.abs(ref["mean"]))
ev = np.max(np.abs(var - ref["var"])) / np.max(np.abs(ref["var"]))
# the factor is sharded: collecting it streams every panel to every rank (collective)
Lf = u.gram.cholesky(True)
G = Lf @ Lf.T
from oracle import gp as ogp
G_ref = ogp.gram(wl.kernel, owl.blocks_of(wl))
eg = np.max(np.abs(G - G_ref)) / np.max(np.abs(G_ref))
w = u.representer_weights
ew = np.max(np.abs(w - ref["weights"])) / np.max(np.abs(ref["weights"]))
m_only = u.mean(wl.Xtest)                    # mean through the weights (prediction points sharded, results gathered)
em2 = np.max(np.abs(m_only - ref["mean"])) / np.max(np.abs(ref["mean"]))
st = ctx.dist_stats()
print("RANK", comm.rank, "of", comm.world, em, ev, eg, ew, em2, st, flush=True)
assert em < 1e-8 and ev < 1e-8 and eg < 1e-12 and ew < 1e-5 and em2 < 1e-8       # (the weights carry the conditioning of G)
assert st["bytes_sent"] > 0 and st["bytes_received"] > 0
comm.barrier()
comm.close()
"""


def _run_ranks(world, grid, workload, nb, port):
    env = dict(os.environ, WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LPGP_DEVICE="0")
    env.pop("LOCAL_RANK", None)
    procs = [subprocess.Popen([sys.executable, "-c", MULTI % {"root": ROOT, "workload": workload, "nb": nb, "grid": grid}],
                              env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}: " + so[-1500:] + se[-3000:]
        assert f"RANK {r} of {world}" in so


@pytest.mark.parametrize("grid,workload,nb,port", [
    ((2, 1), "poisson_2d(n_side=40, n_bdry=33, m_side=9)", 512, 29711),      # 5 conditionings: appends inside a block
    ((1, 2), "poisson_2d(n_side=40, n_bdry=33, m_side=9)", 512, 29721),
    ((3, 1), "poisson_2d(n_side=40, n_bdry=33, m_side=9)", 256, 29731),      # 9 blocks of 256 over 3 ranks
    ((2, 2), "poisson_2d(n_side=40, n_bdry=33, m_side=9)", 256, 29741),      # the 2 x 2 grid of north_star
    ((2, 2), "heat_1d(nt=40, nx=24, m_side=8)", 256, 29751),                 # mixed functional / differential blocks
    ((1, 3), "heat_1d(nt=40, nx=24, m_side=8)", 128, 29761),
])
def test_multi_rank_on_one_gpu_host_transport(grid, workload, nb, port):
    _run_ranks(grid[0] * grid[1], grid, workload, nb, port)
