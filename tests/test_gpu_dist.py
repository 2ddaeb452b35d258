"""Distributed code path on ONE GPU.

* world_size 1 through RCCL (ncclCommInitRank, panel pack / ncclBroadcast / all-reduce of info)
  must reproduce the single-GPU factorisation;
* world_size 2 and 3 as separate processes SHARING the GPU: RCCL refuses two ranks on one device,
  so these jobs use the host-staged test transport (`lpgp_dist_init_host`: every panel goes D2H ->
  control plane -> H2D); everything else is the product path: ownership-filtered assembly, cyclic
  panel ownership with block append, pack / unpack, replicated factor, sharded prediction,
  gathered results.  Every rank must match the oracle.
The 8-GPU node with RCCL over xGMI is only available to the driver; a NumPy mirror of the
algorithm also runs on gloo ranks in tests/test_dist_cpu.py."""
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
os.environ["LPGP_DIST_SELFTEST"] = "1"       # the owner also runs the receive path (panel wiped, then unpacked)
comm = _dist.Comm(0, 1)
ctx = _engine.default_context()
ctx.dist_init(comm)
assert ctx.world == 1 and ctx.comm is comm
wl = problems.poisson_2d(n_side=40, n_bdry=33, m_side=9)      # ragged sizes, 3 panels of 512
u, mean, var = problems.condition_and_predict(wl)
ref = owl.run(wl)
em = np.max(np.abs(mean - ref["mean"])) / np.max(np.abs(ref["mean"]))
ev = np.max(np.abs(var - ref["var"])) / np.max(np.abs(ref["var"]))
print("DIST1", em, ev)
assert em < 1e-8 and ev < 1e-8
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
ctx.dist_init(comm, transport="host")
assert ctx.world == comm.world and ctx.rank == comm.rank
ctx.set_option("nb", %(nb)d)
wl = problems.%(workload)s
u, mean, var = problems.condition_and_predict(wl)
ref = owl.run(wl)
em = np.max(np.abs(mean - ref["mean"])) / np.max(np.abs(ref["mean"]))
ev = np.max(np.abs(var - ref["var"])) / np.max(np.abs(ref["var"]))
# the factor is replicated: every rank holds all of it
Lf = u.gram.cholesky(True)
G = Lf @ Lf.T
from oracle import gp as ogp
G_ref = ogp.gram(wl.kernel, owl.blocks_of(wl))
eg = np.max(np.abs(G - G_ref)) / np.max(np.abs(G_ref))
print("RANK", comm.rank, "of", comm.world, em, ev, eg, flush=True)
assert em < 1e-8 and ev < 1e-8 and eg < 1e-12
comm.barrier()
comm.close()
"""


def _run_ranks(world, workload, nb, port):
    env = dict(os.environ, WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LPGP_DEVICE="0")
    env.pop("LOCAL_RANK", None)
    procs = [subprocess.Popen([sys.executable, "-c", MULTI % {"root": ROOT, "workload": workload, "nb": nb}],
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


@pytest.mark.parametrize("world,workload,nb,port", [
    (2, "poisson_2d(n_side=40, n_bdry=33, m_side=9)", 512, 29711),      # 4 panels, ragged, 5 blocks (append)
    (3, "poisson_2d(n_side=40, n_bdry=33, m_side=9)", 256, 29731),      # 8 panels over 3 ranks
    (2, "heat_1d(nt=40, nx=24, m_side=8)", 256, 29751),                 # mixed functional / differential blocks
])
def test_multi_rank_on_one_gpu_host_transport(world, workload, nb, port):
    _run_ranks(world, workload, nb, port)


def test_assembly_ownership_filter():
    """A rank of a P-rank job assembles only the tile columns of the panels it owns (panel i of
    512 columns belongs to rank i % P); checked on one GPU by assembling AS rank r of P = 3 into
    a matrix pre-filled by a full assembly of a different kernel."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "linpde-gp_amd"))
    import linpde_gp_amd as lp
    from linpde_gp_amd import _engine
    cf = lp.randprocs.covfuncs
    ctx = _engine.default_context()
    rng = np.random.default_rng(5)
    n = 1700                                            # 4 panels of 512 columns, ragged
    X = rng.uniform(-1, 1, size=(n, 2))
    _check_ownership(lp, _engine, ctx, X, _engine.Points(ctx, X))
    # the same through the tensor-grid (Kronecker) assembly: 34 x 50 grid
    from linpde_gp_amd import domains
    Xg = _engine.to_device(domains.TensorProductGrid(np.linspace(-1, 1, 34), np.linspace(-1, 1, 50)))
    assert Xg._lpgp_points.grid_factors is not None
    _check_ownership(lp, _engine, ctx, np.asarray(Xg).reshape(-1, 2), Xg._lpgp_points)


def _check_ownership(lp, _engine, ctx, X, pts):
    cf = lp.randprocs.covfuncs
    n = X.shape[0]
    k_a = cf.TensorProduct(cf.Matern((), nu=2.5), cf.Matern((), nu=2.5))
    k_b = 3.0 * cf.TensorProduct(cf.Matern((), nu=1.5), cf.Matern((), nu=1.5))
    Ga, Gb = k_a.matrix(X), k_b.matrix(X)
    try:
        for rank in range(3):
            mat = _engine.GramMatrix(ctx, n)
            mat.add_block(n)
            mat.assemble(k_b.lower(), pts, None, 0, 0)             # everything: kernel b
            ctx.set_option("test_assemble_as", 3 * 1000 + rank)
            mat.assemble(k_a.lower(), pts, None, 0, 0)             # owned panels: kernel a
            ctx.set_option("test_assemble_as", 0)
            G = mat.todense("gram")
            nb = int(os.environ.get("LPGP_NB", "512"))          # panel width of the factorisation
            owner = (np.arange(n) // nb) % 3
            for j0 in range(0, n, nb):
                cols = slice(j0, min(j0 + nb, n))
                want = Ga if owner[j0] == rank else Gb
                blk = np.tril(G)[:, cols]
                np.testing.assert_allclose(blk, np.tril(want)[:, cols], rtol=0, atol=1e-12)
            del mat
    finally:
        ctx.set_option("test_assemble_as", 0)
