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
ctx.dist_init(comm, transport=%(transport)r, grid=%(grid)r)
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
ld = u.gram.logabsdet()                      # from the replicated diagonal blocks: local, no communication
assert abs(ld - 2.0 * np.sum(np.log(np.diag(Lf)))) < 1e-10 * abs(ld) and abs(ld - np.linalg.slogdet(G_ref)[1]) < 1e-6 * abs(ld)
w = u.representer_weights
ew = np.max(np.abs(w - ref["weights"])) / np.max(np.abs(ref["weights"]))
m_only = u.mean(wl.Xtest)                    # mean through the weights (prediction points sharded, results gathered)
em2 = np.max(np.abs(m_only - ref["mean"])) / np.max(np.abs(ref["mean"]))
st = ctx.dist_stats()
print("RANK", comm.rank, "of", comm.world, em, ev, eg, ew, em2, st, flush=True)
assert em < 1e-8 and ev < 1e-8 and eg < 1e-12 and ew < 1e-5 and em2 < 1e-8       # (the weights carry the conditioning of G)
assert st["bytes_received"] > 0             # (a rank that owns no tile of a small matrix sends nothing)
comm.barrier()
comm.close()
"""


def _run_ranks(world, grid, workload, nb, port, transport="host", window_mb=None, rank_env=None):
    env = dict(os.environ, WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LPGP_DEVICE="0")
    env.pop("LOCAL_RANK", None)
    # every rank runs the NumPy / LAPACK oracle: without a cap each of them starts one BLAS thread per host core (256 on the
    # GPU box) and eight ranks spend their time spinning against each other -- 178 CPU-minutes for an N_tot = 864 problem
    env.update(OPENBLAS_NUM_THREADS="4", OMP_NUM_THREADS="4", MKL_NUM_THREADS="4")
    if window_mb is not None:
        env["LPGP_IPC_WINDOW_MB"] = str(window_mb)
    procs = [subprocess.Popen([sys.executable, "-c", MULTI % {"root": ROOT, "workload": workload, "nb": nb, "grid": grid,
                                                              "transport": transport}],
                              env=dict(env, RANK=str(r), **(rank_env(r) if rank_env else {})),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    if transport == "rccl" and any("lpgp_dist_init failed" in so + se or "lpgp_dist_unique_id failed" in so + se for so, se in outs):
        # the bring-up itself did not work on this box (no loopback interface, RCCL without its socket transport ...):
        # nothing of the product path ran
        pytest.skip("RCCL could not be brought up over loopback sockets here: " + (outs[0][0] + outs[0][1])[-300:])
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
    # 6 144 scattered points: the sharded per-entry assembly on a LARGE block (several column tiles per workgroup, `asm_ct`,
    # with the ownership test per tile), 48 tile columns over two ranks
    ((2, 1), "scattered_2d(n=6144, m=1024)", 512, 29771),
])
def test_multi_rank_on_one_gpu_host_transport(grid, workload, nb, port):
    _run_ranks(grid[0] * grid[1], grid, workload, nb, port)


@pytest.mark.parametrize("grid,workload,nb,port", [
    # the grids of an 8-GPU node, eight processes on the one GPU: 8 x 1 (the default there) with 7 blocks of 128 -- rank 7
    # owns NOTHING -- and 2 x 4 (north_star's example).  (Opt-in until round 3 because a case took 300 s -- which turned out
    # to be eight oracles with 256 BLAS threads each spinning against one another, not the GPU: see _run_ranks.)
    ((8, 1), "poisson_2d(n_side=28, n_bdry=20, m_side=7)", 128, 29871),
    ((2, 4), "poisson_2d(n_side=28, n_bdry=20, m_side=7)", 128, 29881),
    ((4, 2), "heat_1d(nt=30, nx=20, m_side=6)", 128, 29885),              # the P/2 x 2 grid bench.py tries at 8 GPUs; mixed blocks
])
def test_eight_ranks_on_one_gpu_host_transport(grid, workload, nb, port):
    _run_ranks(8, grid, workload, nb, port)


def test_eight_ranks_on_one_gpu_single_stream():
    """The same 8 x 1 job with ONE hardware queue per process (LPGP_SINGLE_STREAM=1: chain and update on the panel stream,
    every cross-stream event a no-op): the schedule must not depend on the streams being distinct."""
    _run_ranks(8, (8, 1), "poisson_2d(n_side=28, n_bdry=20, m_side=7)", 128, 29891, rank_env=lambda r: {"LPGP_SINGLE_STREAM": "1"})


@pytest.mark.parametrize("grid,workload,nb,port,window_mb", [
    ((2, 1), "poisson_2d(n_side=40, n_bdry=33, m_side=9)", 512, 29911, 64),     # a panel fits the window: one round per exchange
    ((2, 2), "poisson_2d(n_side=40, n_bdry=33, m_side=9)", 256, 29921, 1),      # 1-MiB windows: pieces travel in slices
    ((3, 1), "heat_1d(nt=40, nx=24, m_side=8)", 256, 29931, 2),
])
def test_multi_rank_on_one_gpu_direct_peer_transport(grid, workload, nb, port, window_mb):
    """The same jobs with the DIRECT-PEER transport (`lpgp_dist_init_ipc`): panel pieces pushed device to device into
    IPC-mapped receive windows of the peers -- the data path of a multi-GPU node without RCCL, here between processes
    that share the one GPU; the control plane only carries barriers."""
    _run_ranks(grid[0] * grid[1], grid, workload, nb, port, transport="ipc", window_mb=window_mb)


def _rccl_as_if_on_separate_hosts(r):
    """RCCL refuses two ranks on one GPU ("Duplicate GPU detected") -- unless it believes they sit on different HOSTS: a
    distinct NCCL_HOSTID per rank turns the job into a multi-node one whose ranks talk through RCCL's socket transport over
    the loopback interface.  Not the xGMI data path, but the product's RCCL code path end to end: communicator bring-up
    over the control plane, the grouped ncclSend / ncclRecv exchanges of `gather_panel` on the panel stream beside the
    update stream, ncclAllReduce of the factorisation status."""
    return {"NCCL_HOSTID": f"lpgp-test-host-{r}", "NCCL_SOCKET_IFNAME": "lo", "NCCL_IB_DISABLE": "1", "NCCL_NET": "Socket",
            "NCCL_DEBUG": "WARN", "LPGP_FORCE_RCCL": "1"}


@pytest.mark.parametrize("grid,workload,nb,port,collective", [
    ((2, 1), "poisson_2d(n_side=40, n_bdry=33, m_side=9)", 512, 30011, None),
    ((2, 2), "poisson_2d(n_side=40, n_bdry=33, m_side=9)", 256, 30021, None),
    ((3, 1), "heat_1d(nt=40, nx=24, m_side=8)", 256, 30031, None),
    ((1, 2), "poisson_2d(n_side=40, n_bdry=33, m_side=9)", 256, 30061, "bcast"),    # LPGP_DIST_COLLECTIVE=bcast: ncclBroadcast per piece
])
def test_multi_rank_rccl_over_loopback_sockets(grid, workload, nb, port, collective):
    env = (lambda r: dict(_rccl_as_if_on_separate_hosts(r), LPGP_DIST_COLLECTIVE=collective)) if collective else _rccl_as_if_on_separate_hosts
    _run_ranks(grid[0] * grid[1], grid, workload, nb, port, transport="rccl", rank_env=env)


SCOPED = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "linpde-gp_amd"))
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import _dist, _engine, problems
from oracle import workloads as owl
comm = _dist.Comm.from_env()
ctx = _engine.default_context()
ctx.set_option("nb", 128)
ctx.dist_init(comm, transport=%(transport)r, grid=%(grid)r)
wl = problems.poisson_2d(n_side=40, n_bdry=33, m_side=9)        # N_tot = 1732: 14 blocks of 128, five conditionings
ref = owl.run(wl)
recv = {}
for scoped in (1, 0):
    ctx.set_option("scoped_gather", scoped)
    ctx.dist_stats(reset=True)
    prior = problems.build_prior(wl)
    u = prior
    for o in wl.observations:
        X, Y = o.X_as_given()
        b = None if o.noise_var is None else lp.randvars.Normal(np.zeros(Y.shape), np.full(o.X.shape[0], o.noise_var))
        u = u.condition_on_observations(Y, X=X, L=problems.operator_of(o.op, wl.d), b=b)
    ctx.sync()
    recv[scoped] = ctx.dist_stats()["bytes_received"]            # the factorisations alone (the streamed solves need every row)
    mean, var = u.predict(wl.Xtest)
    em = np.max(np.abs(mean - ref["mean"])) / np.max(np.abs(ref["mean"]))
    ev = np.max(np.abs(var - ref["var"])) / np.max(np.abs(ref["var"]))
    Lf = u.gram.cholesky(True)
    assert np.all(np.isfinite(Lf)) and em < 1e-8 and ev < 1e-8, (scoped, em, ev)
    del u
print("SCOPED", comm.rank, recv[1], recv[0], flush=True)
comm.barrier()
comm.close()
"""


@pytest.mark.parametrize("grid,port,transport", [((2, 2), 30211, "host"), ((2, 4), 30221, "host"), ((2, 2), 30231, "ipc")])
def test_scoped_panel_gather_on_2d_grids(grid, port, transport):
    """Round 4 (missing 3): on a Pr x Pc grid with Pr, Pc > 1 a panel's rows travel only to the process row and
    the process column whose trailing updates read them, ~ S (1/Pr + 1/Pc) per rank instead of the whole panel S
    (SURVEY.md section 8e).  Every rank runs the same five conditionings with the scoped and with the unscoped gather:
    both against the oracle -- with LPGP_DIST_POISON=1 the panel buffer is NaN wherever a rank received nothing, so an
    update that read such a row would poison the factor -- and the bytes received during the factorisations compared."""
    world = grid[0] * grid[1]
    env = dict(os.environ, WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LPGP_DEVICE="0", LPGP_DIST_POISON="1",
               OPENBLAS_NUM_THREADS="4", OMP_NUM_THREADS="4", MKL_NUM_THREADS="4", LPGP_IPC_WINDOW_MB="8")
    env.pop("LOCAL_RANK", None)
    procs = [subprocess.Popen([sys.executable, "-c", SCOPED % {"root": ROOT, "grid": grid, "transport": transport}], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    got = {}
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}: " + so[-1500:] + se[-3000:]
        line = [ln for ln in so.splitlines() if ln.startswith("SCOPED")][-1].split()
        got[int(line[1])] = (float(line[2]), float(line[3]))
    scoped, full = sum(v[0] for v in got.values()), sum(v[1] for v in got.values())
    assert all(v[0] <= v[1] for v in got.values()), got
    # expected share of the panel bytes, averaged over the ranks: a rank READS 1/Pr + 1/Pc - 1/(Pr Pc) of a panel and owns
    # 1/(Pr Pc) of it, so it receives (1/Pr + 1/Pc - 2/(Pr Pc)) against 1 - 1/(Pr Pc) to everyone -- 2 x 2: 2/3 (nothing saved
    # on the off-diagonal ranks, 1/3 on the diagonal ones), 2 x 4: 4/7 -- plus the diagonal blocks and tile inverses, which
    # every rank receives either way (a quarter of the traffic at this size)
    pr, pc = grid
    share = (1.0 / pr + 1.0 / pc - 2.0 / (pr * pc)) / (1.0 - 1.0 / (pr * pc))
    print(f"\n[scoped gather {pr} x {pc}, {transport}] bytes received in the factorisations, all ranks: {scoped:.3e} scoped vs {full:.3e} to everyone "
          f"= {scoped / full:.3f} (panel share expected {share:.3f}); per rank {got}")
    assert scoped <= (share + 0.5 * (1.0 - share)) * full, (scoped, full, share)
    if grid == (2, 2):
        assert got[0][0] < 0.75 * got[0][1] and got[3][0] < 0.75 * got[3][1]      # ranks (0,0) and (1,1): half of every panel


FAILING = r"""
import os, sys, time
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "linpde-gp_amd"))
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import _dist, _engine, problems
comm = _dist.Comm.from_env()
ctx = _engine.default_context()
ctx.set_option("nb", 128)
ctx.dist_init(comm, transport="rccl", grid=(2, 1))
wl = problems.poisson_2d(n_side=24, n_bdry=20, m_side=5)
t0 = time.time()
try:
    problems.condition_and_predict(wl)
except Exception as exc:      # noqa: BLE001
    print("RAISED", comm.rank, type(exc).__name__, f"{time.time() - t0:.1f}s", str(exc)[:400].replace("\n", " "), flush=True)
    os._exit(0)               # the job is dead: no further collective, no clean shutdown of the aborted communicator
print("NO ERROR", comm.rank, flush=True)
os._exit(3)
"""


def test_a_failing_rank_does_not_leave_its_peer_waiting():
    """Rank 1 fails locally inside the factorisation (test hook LPGP_TEST_FAIL_RANK / _PANEL in `potrf_dist`) and aborts
    its communicator (`dist_fail`); rank 0, whose next receive would wait for rank 1's pieces for ever, polls RCCL's
    asynchronous error state while it waits for its stream (`sync_stream`), aborts too and raises -- both within seconds."""
    world, port = 2, 30051
    env = dict(os.environ, WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LPGP_DEVICE="0",
               LPGP_TEST_FAIL_RANK="1", LPGP_TEST_FAIL_PANEL="2", LPGP_DIST_TIMEOUT_S="120")
    env.pop("LOCAL_RANK", None)
    procs = [subprocess.Popen([sys.executable, "-c", FAILING % {"root": ROOT}],
                              env=dict(env, RANK=str(r), **_rccl_as_if_on_separate_hosts(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=300))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    if any("lpgp_dist_init failed" in so + se or "lpgp_dist_unique_id failed" in so + se for so, se in outs):
        pytest.skip("RCCL could not be brought up over loopback sockets here: " + (outs[0][0] + outs[0][1])[-300:])
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RAISED {r}" in so, f"rank {r} (rc {p.returncode}): " + so[-1500:] + se[-3000:]
    assert "injected failure" in outs[1][0]
    assert "peer of the job failed" in outs[0][0] or "communicator" in outs[0][0], outs[0][0]


CHAIN = r"""
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "linpde-gp_amd"))
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import _dist, _engine
from oracle import covfuncs as ocf, gp as ogp
comm = _dist.Comm.from_env()
ctx = _engine.default_context()
ctx.set_option("nb", 128)
ctx.dist_init(comm, transport=%(transport)r, grid=%(grid)r)
cf = lp.randprocs.covfuncs
okern, ident = [(1.0, [("expquad", 1.0)])], ocf.identity(1)
prior = lp.GaussianProcess(lp.functions.Zero((1,)), cf.ExpQuad((1,), lengthscales=1.0))
rng = np.random.default_rng(3)
X1, Y1 = rng.uniform(-1, 1, (300, 1)), rng.normal(size=300)          # three blocks of 128 over the ranks
X2, Y2 = np.array([[0.31], [-0.62]]), np.array([0.1, 0.2])
noise = lp.randvars.Normal(np.zeros(300), 1e-2 * np.eye(300))
u1 = prior.condition_on_observations(Y1, X1, b=noise)
p1 = ogp.condition(okern, [ogp.ObsBlock(X1, ident, Y1, 0.0, 1e-2)])
Xt = np.linspace(-1, 1, 9)[:, None]
rel = lambda a, b: np.max(np.abs(a - b)) / np.max(np.abs(b))
# a failed conditioning (duplicated point, negative noise) raises on EVERY rank and is rolled back everywhere
try:
    u1.condition_on_observations(np.zeros(3), np.array([[0.2], [0.2], [0.5]]), b=lp.randvars.Normal(np.zeros(3), -1e-3 * np.eye(3)))
    raise SystemExit("not positive definite, but no LinAlgError")
except np.linalg.LinAlgError:
    pass
m, v = u1.predict(Xt)
assert rel(m, p1.mean(Xt)) < 1e-8 and np.max(np.abs(v - p1.var(Xt))) < 1e-9
u2 = u1.condition_on_observations(Y2, X2)
p2 = ogp.condition(okern, [ogp.ObsBlock(X1, ident, Y1, 0.0, 1e-2), ogp.ObsBlock(X2, ident, Y2)])
m2, v2 = u2.predict(Xt)
assert rel(m2, p2.mean(Xt)) < 1e-8 and np.max(np.abs(v2 - p2.var(Xt))) < 1e-9
# the earlier posterior is still served (view on the leading blocks of the sharded factor), and the later one again
m, v = u1.predict(Xt)
assert rel(m, p1.mean(Xt)) < 1e-8 and np.max(np.abs(v - p1.var(Xt))) < 1e-9
np.testing.assert_allclose(u1.representer_weights, p1.weights, rtol=1e-6, atol=1e-8)
m2, v2 = u2.predict(Xt)
assert rel(m2, p2.mean(Xt)) < 1e-8
# a prediction set smaller than the job: every rank predicts all of it
m3 = u2.mean(Xt[:1])
assert abs(m3[0] - p2.mean(Xt[:1])[0]) < 1e-8
print("CHAIN", comm.rank, "ok", flush=True)
comm.barrier()
comm.close()
"""


@pytest.mark.parametrize("grid,port,transport", [((2, 1), 29771, "host"), ((2, 2), 29781, "host"), ((2, 1), 30041, "rccl")])
def test_multi_rank_views_rollback_and_small_prediction_sets(grid, port, transport):
    """On a sharded factor: a failed conditioning raises on every rank and is rolled back, an earlier posterior of the
    chain keeps working after a later conditioning (view), prediction sets smaller than the job are not sharded.  The
    "rccl" case runs the failure path (status all-reduce, no abort: the failure is a numerical one) through RCCL."""
    world = grid[0] * grid[1]
    env = dict(os.environ, WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LPGP_DEVICE="0")
    env.pop("LOCAL_RANK", None)
    rank_env = _rccl_as_if_on_separate_hosts if transport == "rccl" else (lambda r: {})
    procs = [subprocess.Popen([sys.executable, "-c", CHAIN % {"root": ROOT, "grid": grid, "transport": transport}],
                              env=dict(env, RANK=str(r), **rank_env(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    if transport == "rccl" and any("lpgp_dist_init failed" in so + se or "lpgp_dist_unique_id failed" in so + se for so, se in outs):
        pytest.skip("RCCL could not be brought up over loopback sockets here: " + (outs[0][0] + outs[0][1])[-300:])
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}: " + so[-1500:] + se[-3000:]
        assert f"CHAIN {r} ok" in so


CALIBRATE = r"""
import gc, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "linpde-gp_amd"))
import numpy as np
import linpde_gp_amd as lp
from linpde_gp_amd import _dist, _engine, problems
from oracle import workloads as owl
comm = _dist.Comm.from_env()
ctx = _engine.default_context()
ctx.set_option("nb", 128)
ctx.dist_init(comm, transport=%(transport)r)
assert ctx.grid == (4, 1)                                   # the library's default: P x 1
probe = ctx.link_probe(1 << 20, 2)                          # what bench.py reports as config.link_probe
W = comm.world
pair = np.array(probe["pair_gbps"])
assert pair.shape == (W, W) and np.all(np.diag(pair) == 0.0) and np.all(pair[~np.eye(W, dtype=bool)] > 0.0), probe
assert len(probe["one_to_all_gbps"]) == W and min(probe["one_to_all_gbps"]) > 0.0 and min(probe["all_to_all_inbound_gbps"]) > 0.0, probe
wl = problems.poisson_2d(n_side=26, n_bdry=20, m_side=6)     # 6 blocks of 128
ref = owl.run(wl)
for grid, bcast in (((4, 1), 0), ((2, 2), 0), ((4, 1), 1), ((1, 4), 0)):       # bench.py's trials: regrid between problems
    ctx.dist_set_grid(*grid)
    ctx.set_option("dist_bcast", bcast)
    assert ctx.grid == grid
    u, mean, var = problems.condition_and_predict(wl)
    em = np.max(np.abs(mean - ref["mean"])) / np.max(np.abs(ref["mean"]))
    ev = np.max(np.abs(var - ref["var"])) / np.max(np.abs(ref["var"]))
    assert em < 1e-8 and ev < 1e-8, (grid, bcast, em, ev)
    try:
        ctx.dist_set_grid(2, 2) if grid != (2, 2) else ctx.dist_set_grid(4, 1)
        raise AssertionError("regrid accepted while a matrix of the old grid is alive")
    except lp._lib.LpgpError as exc:
        assert "still alive" in str(exc)
    del u
    gc.collect()
print("CALIBRATE-OK", comm.rank, probe["pair_median_gbps"], flush=True)
comm.barrier()
comm.close()
"""


@pytest.mark.parametrize("transport,port", [("ipc", 30111), ("rccl", 30121)])
def test_link_probe_and_regrid_between_problems(transport, port):
    """What `bench.py --gpus N` does before its timed region on a multi-GPU node ("first contact"): the
    link probe through the transport the panels use, then the same problem on several process grids and with both
    collectives, the grid changed AFTER the bring-up while no matrix is alive (`lpgp_dist_set_grid`) -- refused while one
    is.  Four ranks sharing the one GPU; every variant against the oracle on every rank."""
    env = dict(os.environ, WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LPGP_DEVICE="0", LPGP_IPC_WINDOW_MB="8",
               LPGP_SINGLE_STREAM="1", OPENBLAS_NUM_THREADS="4", OMP_NUM_THREADS="4")
    env.pop("LOCAL_RANK", None)
    extra = _rccl_as_if_on_separate_hosts if transport == "rccl" else (lambda r: {})
    procs = [subprocess.Popen([sys.executable, "-c", CALIBRATE % {"root": ROOT, "transport": transport}], env=dict(env, RANK=str(r), **extra(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(4)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    if transport == "rccl" and any("lpgp_dist_init failed" in so + se or "lpgp_dist_unique_id failed" in so + se for so, se in outs):
        pytest.skip("RCCL could not be brought up over loopback sockets here")
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"CALIBRATE-OK {r}" in so, f"rank {r}: " + so[-1500:] + se[-3000:]
