#!/usr/bin/env python3
"""bench.py -- condition + predict on the BASELINE workload (c3: 2-D Poisson-Dirichlet,
N = 16 384 collocation + 4 x 128 boundary observations, M = 64 x 64 prediction points).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over the synthetic workload: block-by-block
conditioning (Gram/cross-block assembly, blocked Cholesky with block append, representer
weights) + prediction (cross-covariance assembly, posterior mean, marginal variance).
Point sets are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.

Order of a run at N = 1: the CPU baseline (the oracle on the bench workload itself, ~30 s of host time) FIRST, then
the GPU section in one piece (warm-up, timed region, per-phase stamps, per-kernel HIP-event passes), so that the
device is busy in one contiguous stretch.

N > 1 (launched by `python -m torch.distributed.run`, env RANK/LOCAL_RANK/WORLD_SIZE/
MASTER_ADDR/MASTER_PORT): one process per GPU, all ranks factor ONE problem whose Gram matrix is
sharded in 2-D block-cyclic tiles over a Pr x Pc process grid; panels travel as grouped RCCL
point-to-point sends over xGMI, prediction points are sharded over the ranks.
The problem grows with N so that the algorithmic flops PER GPU stay those of c3 ("scaling": "weak": grid side
128 -> 144 / 162 / 182 at 2 / 4 / 8 GPUs, prediction grid side = half of it; c3 itself is 33 ms of
factorisation, far too small to shard).  Before the timed region the run CALIBRATES itself, because the
builder never had more than one GPU: `config.link_probe` (measured GB/s of every ordered pair of ranks, of one
rank sending to all, of all to all, through the transport the panels use) and `config.trials` (the same
workload for two steps each on the P x 1 grid with the split panel gather, on the P/2 x 2 grid, and on
P x 1 with one ncclBroadcast per panel piece instead of the point-to-point group; the fastest variant runs
the timed region; LPGP_GRID / LPGP_DIST_COLLECTIVE pin a variant and skip the trials).  After the timed
region the run also factors the BASELINE configuration named for that GPU count -- c4 (256 x 256, N_tot =
66 560) at 8 GPUs, c5 (heat, N_tot = 33 600) at 4 and 2 -- reported under `configs` with size-independent
parity properties (LPGP_BENCH_EXTRA=c4,c5 / none overrides the choice).
LPGP_BENCH_STRONG=1 shards c3 itself ("strong"); LPGP_BENCH_REPLICAS=1 runs one independent c3 per rank.
The product path never imports torch; ranks rendezvous over a plain TCP star.
"""
import argparse
import gc
import glob
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "linpde-gp_amd"))

import numpy as np  # noqa: E402

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak (vendor datasheet)
HBM_PEAK_GBPS = 8000.0         # MI355X_MICROARCH.md: 8 TB/s spec
ROOFLINE_SLOT = "syrk_trailing"
T_START = time.time()


def _blas_info():
    try:
        from threadpoolctl import threadpool_info
        pools = [p for p in threadpool_info() if p.get("user_api") == "blas"]
        threads = max((p.get("num_threads", 1) for p in pools), default=os.cpu_count() or 1)
        blas = ",".join(sorted({f"{p.get('internal_api')}-{p.get('version')}" for p in pools}))
    except Exception:  # pragma: no cover
        threads, blas = os.cpu_count() or 1, "unknown"
    return int(threads), blas


def cpu_baseline(problems, wl, sample_side=0):
    """The oracle (NumPy/SciPy restatement of the reference path: vectorised NumPy assembly, LAPACK
    dpotrf / dpotrs / dtrtrs through scipy.linalg) timed on the host cores ON THE BENCH WORKLOAD ITSELF
    (c3: N_tot = 16 896, ~30 GB of host memory at the peak of the NumPy assembly).  `sample_side` > 0, or
    less than 48 GB of free host memory: the same workload on a smaller grid, labelled as a sample.
    Returns (json dict, oracle result or None if a sample was run)."""
    from oracle import workloads as owl
    threads, blas = _blas_info()
    full = sample_side <= 0
    if full:
        try:
            import psutil
            if psutil.virtual_memory().available < 48 * 2**30 and wl.n_total > 12000:
                full, sample_side = False, 72
        except Exception:  # pragma: no cover
            pass
    w = wl if full else problems.poisson_2d(n_side=sample_side, m_side=32)
    res = owl.run(w, want_cond=full)
    sec = res["seconds"]
    what = ("the bench workload itself" if full else f"BOUNDED SAMPLE of the workload at {sample_side}x{sample_side} collocation")
    return {
        "value": w.total_flops() / sec["total"] / 1e9,
        "unit": "GFLOP/s",
        "cores": threads,
        "kind": "port",
        "sample": (f"{what}: {w.name}, N_tot={w.n_total}, M={w.Xtest.shape[0]}; NumPy/SciPy oracle, BLAS={blas} "
                   f"({threads} threads; the NumPy assembly is single-threaded); {sec['total']:.2f} s total, timed before the GPU section"),
        "n_total": int(w.n_total),
        "seconds": sec["total"],
        "phase_seconds": {k: v for k, v in sec.items() if k != "total"},
        "host_cpus": os.cpu_count(),
    }, (res if full else None)


def parity_report(mean, var, ref, wl):
    """The ONE posterior criterion (tests/conftest.py: `posterior_tolerances`), restated for the JSON line:
    mean 1e-8 of max|mean|; variance 1e-8 of max|var|; no absolute slack."""
    mean_atol = 1e-8 * float(np.max(np.abs(ref["mean"])))
    var_atol = 1e-8 * float(np.max(np.abs(ref["var"])))
    em, ev = float(np.max(np.abs(mean - ref["mean"]))), float(np.max(np.abs(var - ref["var"])))
    return {
        "n_total": int(wl.n_total),
        "mean_rel_err": em / float(np.max(np.abs(ref["mean"]))),
        "var_rel_err": ev / float(np.max(np.abs(ref["var"]))),
        "mean_abs_err": em, "mean_atol": mean_atol, "var_abs_err": ev, "var_atol": var_atol,
        "criterion": "mean <= 1e-8 max|mean|; var <= 1e-8 max|var|",
        "pass": bool(em <= mean_atol and ev <= var_atol),
        "var_max": float(np.max(ref["var"])),
        "gram_cond2_estimate": ref.get("cond2"),      # lower bound, power / inverse iteration on the oracle's G and factor
        "cpu_seconds_full": ref["seconds"],
    }


def parity_by_properties(problems, wl, u, mean, var):
    """Size-independent checks for configurations whose oracle does not fit a bench run (c4: 380 s of host time; the
    full-size comparisons are tests/test_gpu_zz_c4_full.py and test_gpu_configs.py::test_c5_heat_full_size_vs_oracle):
    residual of G w = r on re-evaluated rows of the collocation block, variance inside [0, k(x,x)], the closed-form
    solution where there is one (c5: the reference's own accuracy bar, test_heat.py:25-28), the known maximum of the
    Poisson solution (c3 / c4).  Collective in a multi-GPU job."""
    kxx = float(sum(sc for sc, _ in wl.kernel))
    big = max(o.X.shape[0] for o in wl.observations)
    rows = np.unique(np.array([0, big // 3 + 17, big // 2 + 5, big - 1]))
    res = float(np.max(np.abs(problems.row_residual(u, wl, rows))))
    ymax = float(max(np.max(np.abs(o.Y)) for o in wl.observations))
    out = {"finite": bool(np.all(np.isfinite(mean)) and np.all(np.isfinite(var))),
           "var_min": float(np.min(var)), "var_max": float(np.max(var)), "prior_var": kxx,
           "var_in_bounds": bool(np.all(var > -1e-9 * kxx) and np.all(var < kxx)),
           "gram_row_residual_max": res, "gram_row_residual_tol": 1e-5 * max(ymax, 1.0)}
    ok = out["finite"] and out["var_in_bounds"] and res < out["gram_row_residual_tol"]
    sol = problems.analytic_solution(wl)
    if sol is not None:
        out["analytic_solution_max_err"] = float(np.max(np.abs(mean - sol)))
        out["analytic_solution_tol"] = 3e-2
        ok = ok and out["analytic_solution_max_err"] < 3e-2
    elif wl.name.startswith("poisson2d"):
        out["mean_max"] = float(np.max(mean))
        out["mean_max_expected"] = 0.5894          # max of the solution of -Lap u = 2, u = 0 on the boundary of [-1,1]^2
        ok = ok and abs(out["mean_max"] - 0.5894) < 1e-2
    out["pass"] = bool(ok)
    return out


def _weak_sides(world):
    """Grid side whose algorithmic flops (Workload.algorithmic_work: N^3/3 + 2N^2 + N^2 M + 4NM,
    N = n^2 + 4n, M = (n/2)^2) are closest to world x those of c3; m_side = n_side / 2."""
    def flops_of(n, ms):
        N, M = float(n * n + 4 * n), float(ms * ms)
        return N**3 / 3.0 + 2.0 * N * N + N * N * M + 4.0 * N * M
    target = world * flops_of(128, 64)
    best = min(range(128, 513, 2), key=lambda n: abs(flops_of(n, n // 2) - target))
    return best, best // 2


def csrc_sha16():
    """Identity of the kernel sources this process runs (the built library follows them: __graft_entry__.build()):
    sha256 over csrc/*.hip|h|cpp and include/*.h, first 16 hex digits."""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "linpde-gp_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "linpde-gp_amd", "csrc", "*.h"))
                   + glob.glob(os.path.join(ROOT, "linpde-gp_amd", "csrc", "*.cpp")) + glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def profile_tag(workload, n_side=128, m_side=64):
    """Tag of the committed rocprofv3 collection (`profiles/rNN_bench_<tag>_summary.json`) of a bench workload: the BASELINE
    configurations only; None for any other workload or size (no PMC passes of that command exist)."""
    if workload == "poisson2d":
        return {(128, 64): "c3", (256, 128): "c4"}.get((n_side, m_side))
    return {"poisson1d": "c2", "heat1d": "c5"}.get(workload)


def pmc_traffic(sha, kernel_symbol, profiles_dir=None, tag="c3"):
    """HBM bytes per launch of the roofline kernel from the committed rocprofv3 --pmc passes of the SAME command
    (FETCH_SIZE / WRITE_SIZE cannot be read from inside the process; scratch/collect_profiles.sh collects them and records
    the command and the source identity it ran on; `tag` names the configuration, see profile_tag).  A summary collected on OTHER kernel sources is not reported as
    `traffic` (it is named under `traffic_stale`)."""
    out = {"traffic": None, "traffic_source": None}
    if tag is None:            # a workload without a committed PMC collection: nothing to report, nothing stale to name
        out["traffic_note"] = "no rocprofv3 --pmc collection of this workload is committed (profiles/: c2, c3, c4, c5)"
        return out
    profiles_dir = profiles_dir or os.path.join(ROOT, "profiles")
    for path in sorted(glob.glob(os.path.join(profiles_dir, f"r*_bench_{tag}_summary.json")), reverse=True):
        try:
            with open(path) as f:
                summ = json.load(f)
        except Exception:
            continue
        v = summ.get("syrk_hbm_bytes_per_launch")
        if v is None:
            continue
        at = (summ.get("bench_line") or {}).get("config", {}).get("csrc_sha16")
        rel = os.path.relpath(path, ROOT) if path.startswith(ROOT) else path
        if at == sha and kernel_symbol.split("(")[0].strip() in (summ.get("syrk_kernel") or ""):
            return {"traffic": v, "traffic_source": rel, "traffic_collected_at_csrc": at,
                    "traffic_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench command on these kernel sources; (2*FETCH_SIZE + WRITE_SIZE) KB per dispatch"}
        out.setdefault("traffic_stale", {"value": v, "source": rel, "collected_at_csrc": at, "running_csrc": sha,
                                          "kernel": summ.get("syrk_kernel"),
                                          "note": "collected on other kernel sources (or another roofline kernel) than the running tree: not reported as `traffic`"})
    return out


def visible_gpus():
    """Number of GPUs this process could open, WITHOUT a HIP call (the launcher never touches the GPU): KFD topology
    nodes with SIMDs, narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES.  None when the
    topology cannot be read (the ranks then find out themselves)."""
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return 0 if not os.path.exists("/dev/kfd") else None
    n = 0
    for path in nodes:
        try:
            with open(path) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
        except OSError:
            return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if os.environ.get(var, "").strip():
            n = min(n, len([d for d in os.environ[var].split(",") if d.strip()]))
    return n


def _free_port_pair():
    """MASTER_PORT with MASTER_PORT + 1 free as well (the control plane of `_dist.Comm` listens there)."""
    import socket
    for _ in range(64):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        if port + 1 < 65536:
            try:
                with socket.socket() as s2:
                    s2.bind(("127.0.0.1", port + 1))
                return port
            except OSError:
                continue
    raise SystemExit("bench.py: no free port pair for the control plane")


def launch_ranks(n, argv):
    """`python bench.py --gpus N` called the way the reference is called -- from ONE process
    (/root/reference/src/linpde_gp/randprocs/_gaussian_process/_conditional.py:253-294), no launcher around it.  This
    process starts N fresh children of itself, one per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / a free
    MASTER_PORT: the environment `python -m torch.distributed.run` would give them), BEFORE it has imported the engine
    or made any GPU call -- never a re-exec of a process that touched the GPU.  It relays rank 0's stdout (the ONE JSON
    line), ends the other ranks as soon as one fails, and returns non-zero with ONE error line if a rank failed, if
    fewer than N GPUs are visible, or if the line printed is not a line for N GPUs."""
    import signal
    import subprocess
    import tempfile
    shared = "LPGP_DEVICE" in os.environ        # bring-up on one GPU (tests): every rank opens that device
    have = int(os.environ["LPGP_BENCH_ASSUME_GPUS"]) if "LPGP_BENCH_ASSUME_GPUS" in os.environ else visible_gpus()   # (tests: the ranks' own failure path)
    if have is not None and have < (1 if shared else n):
        sys.stderr.write(f"bench.py: --gpus {n} but {have} GPU(s) visible on this node: nothing measured\n")
        return 2
    port = _free_port_pair()
    procs, errs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   LPGP_BENCH_LAUNCHER="self")
        # rank 0's stderr passes through; the other ranks' is kept and shown only for the rank that failed first
        errs.append(None if r == 0 else tempfile.TemporaryFile())
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env, start_new_session=True,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=errs[r]))

    def end_all():
        for p_ in procs:
            if p_.poll() is None:
                try:
                    os.killpg(p_.pid, signal.SIGKILL)       # exactly the process groups started above
                except OSError:
                    pass

    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    limit = float(os.environ.get("LPGP_BENCH_TIMEOUT", "1500")) + 60.0
    t0, failed = time.time(), None
    try:
        while any(p_.poll() is None for p_ in procs):
            bad = [(r, p_.returncode) for r, p_ in enumerate(procs) if p_.poll() not in (None, 0)]
            if bad or time.time() - t0 > limit:
                failed = bad[0] if bad else ("all", "timeout")
                time.sleep(0.5)          # let the peers print what they know, then end them
                break
            time.sleep(0.05)
    finally:
        end_all()
    reader.join(5.0)
    bad = [(r, p_.returncode) for r, p_ in enumerate(procs) if p_.returncode != 0]
    failed = failed or (bad[0] if bad else None)
    if failed is not None:
        if isinstance(failed[0], int) and errs[failed[0]] is not None:
            errs[failed[0]].seek(0)
            sys.stderr.write("".join(errs[failed[0]].read().decode(errors="replace").splitlines(True)[-30:]))
        sys.stderr.write(f"bench.py: --gpus {n}: rank {failed[0]} ended with status {failed[1]}; the other ranks were stopped, no line printed\n")
        return 1
    text = (out0[0] if out0 else b"").decode(errors="replace")
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    try:
        got = json.loads(lines[-1])["n_gpus"]
    except (IndexError, ValueError, KeyError):
        got = None
    if got != n:
        sys.stderr.write(f"bench.py: --gpus {n}: rank 0 printed no line for {n} GPUs (n_gpus = {got})\n")
        return 1
    sys.stdout.write(text)
    sys.stdout.flush()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-side", type=int, default=128, help="collocation grid side (c3: 128)")
    ap.add_argument("--m-side", type=int, default=64, help="prediction grid side (c3: 64)")
    ap.add_argument("--cpu-side", type=int, default=0,
                    help="0 (default): time the CPU oracle on the bench workload itself; > 0: on a bounded sample of this grid side")
    ap.add_argument("--workload", default="poisson2d", choices=["poisson2d", "poisson1d", "heat1d", "scattered2d", "poisson1d_c1", "heat_reference"],
                    help="poisson2d = c3/c4 (the metric's workload), poisson1d = c2 (N=8192), heat1d = c5 (N=32768 + IC/BC/noisy interior), "
                         "poisson1d_c1 = c1 (N=512 + 32 repeated noisy boundary values, N_tot = 544), heat_reference = the reference's own heat "
                         "problem at its own sizes (tests/linpde_gp/problems/test_heat.py:56-99, N_tot = 2 105, 50 x 50 test grid), "
                         "scattered2d = 16 384 noisy values at scattered points (not a BASELINE config: every block through the "
                         "per-entry assembly kernels, none through the Kronecker path)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--check", action="store_true", help="also compare with the CPU oracle at full size")
    args = ap.parse_args()
    global T_START
    T_START = time.time()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a plain `python bench.py --gpus N`: this process becomes the launcher (no GPU call, no engine import) and exits
        # with the job's status; the N ranks are fresh children that re-enter main() with WORLD_SIZE set
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: a line for another GPU count than the one asked for is never printed")

    # a hung collective must not hold the node until the caller's own limit
    limit = float(os.environ.get("LPGP_BENCH_TIMEOUT", "1500"))
    watchdog = threading.Timer(limit, lambda: (sys.stderr.write(f"bench.py: no result after {limit:.0f} s, aborting\n"),
                                               os._exit(3)))
    watchdog.daemon = True
    watchdog.start()

    import linpde_gp_amd as lp
    from linpde_gp_amd import _dist, _engine, problems
    # The timed region runs the package's throughput mode, an explicit OPT-IN (`lp.config.lazy_factorization = True`): the
    # factorisation is enqueued instead of awaited inside `condition_on_observations`, and `predict` rides inside the last
    # one (`lpgp_potrf_predict`).  Same numbers as the default mode, whose step time is reported beside it
    # (`modes.eager_default`: status read back inside every conditioning, as the reference raises there).  World > 1: the
    # ranks agree on the status collectively inside the conditioning, whatever the flag says.
    bench_lazy = not os.environ.get("LPGP_BENCH_EAGER")          # LPGP_BENCH_EAGER=1: the default mode in the timed region (A/B aid)
    lp.config.lazy_factorization = bench_lazy

    def make_workload(n_side, m_side):
        if args.workload == "poisson1d":
            return problems.poisson_1d()            # c2
        if args.workload == "heat1d":
            return problems.heat_1d()               # c5
        if args.workload == "scattered2d":
            return problems.scattered_2d(n=16384, m=4096)
        if args.workload == "poisson1d_c1":
            return problems.poisson_1d(512, n_bdry_repeats=16, noise_var=1e-4, m=256)      # c1 (tests/test_gpu_configs.py)
        if args.workload == "heat_reference":
            return problems.heat_reference()
        return problems.poisson_2d(n_side=n_side, m_side=m_side)

    # ---- CPU baseline FIRST (N = 1 only): the host cores are idle, and the GPU section afterwards is one busy stretch ----
    cpu_json, ref = None, None
    if world == 1 and not args.no_cpu:
        cpu_json, ref = cpu_baseline(problems, make_workload(args.n_side, args.m_side), args.cpu_side)
    elif world == 1 and args.check:
        from oracle import workloads as owl
        ref = owl.run(make_workload(args.n_side, args.m_side), want_cond=True)

    comm = _dist.Comm.from_env()
    ctx = _engine.default_context()          # device = LOCAL_RANK
    replicas = bool(int(os.environ.get("LPGP_BENCH_REPLICAS", "0")))
    strong = bool(int(os.environ.get("LPGP_BENCH_STRONG", "0")))
    dist_note = None
    force_dist = world == 1 and bool(int(os.environ.get("LPGP_BENCH_FORCE_DIST", "0")))
    if force_dist:
        # measurement aid: ONE rank through the distributed code path (1 x 1 grid: sharded-storage kernels, panel
        # gather / unpack, staircase update, panel-streaming solves -- everything but the wire)
        os.environ["LPGP_FORCE_RCCL"] = "1"
        ctx.dist_init(comm)
        dist_note = "single rank through the DISTRIBUTED code path (LPGP_BENCH_FORCE_DIST=1)"
    transport = os.environ.get("LPGP_DIST_TRANSPORT", "rccl")
    loopback = bool(int(os.environ.get("LPGP_BENCH_RCCL_LOOPBACK", "0")))
    if loopback and world > 1:
        # bring-up aid for a box whose ranks SHARE one GPU: RCCL refuses two ranks on one device unless it takes them for
        # different hosts; a distinct NCCL_HOSTID per rank makes it run its socket transport over the loopback interface.
        # The product's RCCL code path end to end, NOT the xGMI data path and never a benchmark configuration (run it with
        # LPGP_DEVICE=0 so that every rank opens the one GPU).
        os.environ.update(NCCL_HOSTID=f"lpgp-bench-host-{rank}", NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_NET="Socket")
    if world > 1 and not replicas:
        # RCCL communicator: distributed factorisation of ONE problem.  If the communicator cannot be created (on every
        # rank alike) the run falls back to the direct-peer transport (IPC-mapped windows, device-to-device pushes), and
        # if that cannot be set up either, to independent replicas -- and says so.  (A failed lpgp_dist_init leaves a
        # plain single-GPU context behind, so the next attempt starts clean.)
        # LPGP_DIST_TRANSPORT=ipc selects the direct-peer transport; =host: bring-up on a box whose ranks share one GPU
        # (messages staged through the host; never a benchmark configuration)
        first_err = ""
        for attempt in ([transport, "ipc"] if transport == "rccl" else [transport]):
            try:
                ctx.dist_init(comm, transport=attempt)
                ok, err = True, ""
            except Exception as exc:            # noqa: BLE001 (reported in the JSON line)
                ok, err = False, f"{type(exc).__name__}: {exc}"
            oks = comm.allgather((ok, err))
            if all(o for o, _ in oks):
                if attempt != transport:
                    dist_note = f"RCCL communicator creation failed ({first_err[:160]}): direct-peer (IPC) transport instead; "
                    if rank == 0:
                        sys.stderr.write("bench.py: " + dist_note + "\n")
                transport = attempt
                break
            if any(o for o, _ in oks):
                raise SystemExit(f"{attempt} transport set up on some ranks only: " + "; ".join(e for o, e in oks if not o))
            first_err = first_err or next(e for _, e in oks if e)
        else:
            replicas = True
            dist_note = "communicator creation failed (" + first_err[:200] + "): independent replicas instead"
            if rank == 0:
                sys.stderr.write("bench.py: " + dist_note + "\n")
    distributed = world > 1 and not replicas
    info = ctx.device_info()
    # RCCL prints a version banner through C stdio when a communicator is created; on a pipe it would leave the
    # buffer only at exit, i.e. AFTER the JSON line.  Push it out now, on every rank.
    import ctypes
    _libc = ctypes.CDLL(None)
    _libc.fflush(None)

    n_side, m_side = args.n_side, args.m_side
    weak = distributed and not strong and (n_side, m_side) == (128, 64) and args.workload == "poisson2d"
    if weak:
        n_side, m_side = _weak_sides(world)
    wl = make_workload(n_side, m_side)

    def timed_steps(w, prior_, dev_, k):
        """k steps bracketed by barrier + device sync on both sides; seconds, max over ranks."""
        comm.barrier()
        ctx.sync()
        t0_ = time.perf_counter()
        out_ = None
        for _ in range(k):
            out_ = None          # drop the previous posterior BEFORE the next step allocates its factor (the pool then reuses the buffer)
            out_ = problems.condition_and_predict(w, prior=prior_, device_arrays=dev_)
        ctx.sync()
        dt_ = comm.allreduce_max(time.perf_counter() - t0_)
        comm.barrier()
        return dt_, out_

    # ---- first contact with a real fabric: measure it, then let the measurement pick the variant ----
    # Everything before the timed region -- link probe and trials -- shares ONE budget (LPGP_BENCH_BUDGET_S, default 300 s of wall
    # time, agreed on by all ranks): a trial whose predecessor's cost says it would overrun is skipped, and the line says so.
    # (The builder never had more than one GPU; over loopback sockets a trial takes 20-46 s, on xGMI it should take well under
    # one: the budget is there for the fabric nobody has measured.)
    link_probe, trials, chosen = None, None, None
    budget_s = float(os.environ.get("LPGP_BENCH_BUDGET_S", "300"))
    t_cal0 = time.time()

    def budget_left():
        return comm.allreduce_max(-(budget_s - (time.time() - t_cal0))) * -1.0      # the MIN over ranks of what is left

    if distributed and transport != "host":
        try:
            link_probe = ctx.link_probe(32 << 20, 3)
        except Exception as exc:                # noqa: BLE001
            link_probe = {"error": f"{type(exc).__name__}: {exc}"}
    pinned = "LPGP_GRID" in os.environ or "LPGP_DIST_COLLECTIVE" in os.environ or bool(int(os.environ.get("LPGP_BENCH_NO_TRIALS", "0")))
    skipped_trials = []
    if distributed and world >= 4 and world % 2 == 0 and not pinned:
        variants = [("Px1_p2p_split", (world, 1), 0), ("P/2x2_p2p", (world // 2, 2), 0), ("Px1_bcast", (world, 1), 1)]
        if world == 8:
            # north_star's own grid first (BASELINE.json: "2D block-cyclic tiles across the 8 GPUs", SURVEY section 8e: Pr x Pc = 2 x 4), so that
            # a budget that cuts the list still measures it; the product's default (P x 1: one exchange per panel, lpgp.h) second
            variants = [("2x4_p2p", (2, 4), 0), ("Px1_p2p_split", (8, 1), 0), ("P/2x2_p2p", (4, 2), 0), ("Px1_bcast", (8, 1), 1)]
        trials = {}
        lp.config.gram_capacity_hint = wl.n_total
        last_cost = 0.0
        for name, grid, bc in variants:
            left = budget_left()
            if trials and left < 1.5 * last_cost:          # the first variant always runs (it is the default configuration's warm-up anyway)
                skipped_trials.append({"variant": name, "reason": f"{left:.0f} s of the {budget_s:.0f}-s calibration budget left, the previous trial took {last_cost:.0f} s"})
                trials[name] = {"grid": list(grid), "collective": "bcast" if bc else "p2p", "ms_per_step": None, "skipped": skipped_trials[-1]["reason"]}
                continue
            t_tr = time.time()
            ctx.dist_set_grid(*grid)
            ctx.set_option("dist_bcast", bc)
            dev_t = problems.upload(wl)
            prior_t = problems.build_prior(wl)
            problems.condition_and_predict(wl, prior=prior_t, device_arrays=dev_t)       # untimed: allocation, connections
            dt_t, last_t = timed_steps(wl, prior_t, dev_t, 2)
            trials[name] = {"grid": list(grid), "collective": "bcast" if bc else "p2p", "ms_per_step": dt_t / 2 * 1e3}
            del dev_t, prior_t, last_t          # every matrix of this grid must be gone before the next lpgp_dist_set_grid
            gc.collect()
            last_cost = comm.allreduce_max(time.time() - t_tr)
        chosen = min((k_ for k_ in trials if trials[k_]["ms_per_step"] is not None), key=lambda k_: trials[k_]["ms_per_step"])
        _, grid, bc = next(v for v in variants if v[0] == chosen)
        ctx.dist_set_grid(*grid)
        ctx.set_option("dist_bcast", bc)
    calibration_s = time.time() - t_cal0

    # (the augmented form of the fused pipeline keeps the prediction's right-hand side as ROWS below the matrix's blocks: room for them)
    aug_rows = ((wl.Xtest.shape[0] + 1 + 127) // 128) * 128 if (not distributed and ctx.get_option("ride_aug")) else 0
    lp.config.gram_capacity_hint = ((wl.n_total + 127) // 128) * 128 + 128 * len(wl.observations) + aug_rows if aug_rows else wl.n_total
    dev = problems.upload(wl)                # point sets resident in HBM before timing
    prior = problems.build_prior(wl)

    def step():
        u_, mean_, var_ = problems.condition_and_predict(wl, prior=prior, device_arrays=dev)
        return mean_, var_

    for _ in range(args.warmup):
        mean, var = step()
    # ---- timed region: exactly K steps, no instrumentation ----
    dt, last = timed_steps(wl, prior, dev, args.steps)
    _, mean, var = last
    # ---- the two phases of a step (un-instrumented steps; the conditioning phase ends with the read-back of the
    #      factorisation status, so a host stamp between the phases costs nothing): max over ranks ----
    ph = []
    for _ in range(2):
        st_ = []
        problems.condition_and_predict(wl, prior=prior, device_arrays=dev, stamps=st_)
        ph.append((st_[1] - st_[0], st_[2] - st_[1]))
    phase_rows = comm.gather([min(p_[0] for p_ in ph) * 1e3, min(p_[1] for p_ in ph) * 1e3])

    # ---- what a user of the reference would call, and what it costs (N = 1) ----
    # The schedule of the timed region, by name, and the step time of the TWO-PIPELINE schedule beside it on every line: multi-GPU jobs
    # always run the two pipelines (the ranks agree on the factorisation's status inside the conditioning; lpgp_potrf_predict is
    # single-GPU), so a scaling curve must be read against `two_pipeline_ms_per_step` of the N = 1 line, not against its fused
    # `ms_per_step`.
    mode_name = "two_pipelines" if (distributed or not bench_lazy) else "fused_factor_and_predict"
    two_pipeline_ms = dt / args.steps * 1e3 if mode_name == "two_pipelines" else None
    tp_syrk = None
    # (LPGP_BENCH_NO_MODES -- set by the profile collections only, scratch/collect_profiles.sh: the profiled process must contain ONE
    #  schedule, or the per-kernel averages of rocprofv3 --stats mix the two)
    if two_pipeline_ms is None and not os.environ.get("LPGP_BENCH_NO_MODES"):
        saved_lazy = lp.config.lazy_factorization
        try:
            lp.config.lazy_factorization = False
            step(); ctx.sync()
            k_tp = max(3, min(args.steps, 10))
            t0_tp = time.perf_counter()
            for _ in range(k_tp):
                last_tp = step()
            ctx.sync()
            two_pipeline_ms = (time.perf_counter() - t0_tp) / k_tp * 1e3
            del last_tp
            # the roofline kernel in THAT schedule (it does not share the chip with the substitution's updates there): one extra step under HIP events
            ctx.profile_reset(); ctx.profile_enable([ROOFLINE_SLOT])
            step(); ctx.sync()
            tp_syrk = ctx.profile_get()[ROOFLINE_SLOT]
            ctx.profile_enable(False); ctx.profile_reset()
        finally:
            lp.config.lazy_factorization = saved_lazy
    modes, ref_seq, e2e = None, None, None
    if world == 1 and not os.environ.get("LPGP_BENCH_NO_MODES"):
        k_m = max(3, min(args.steps, 10))

        def timed(fn, k=k_m):
            fn()                                   # untimed: first call of this variant
            ctx.sync()
            t0_ = time.perf_counter()
            r_ = None
            for _ in range(k):
                r_ = None
                r_ = fn()
            ctx.sync()
            return (time.perf_counter() - t0_) / k * 1e3, r_

        def conditioned():
            u_ = prior
            for i_, o_ in enumerate(wl.observations):
                Y_ = o_.Y if o_.grid is None else o_.Y.reshape(tuple(len(f_) for f_ in o_.grid))
                b_ = None if o_.noise_var is None else lp.randvars.Normal(np.zeros(Y_.shape), np.full(o_.X.shape[0], o_.noise_var))
                u_ = u_.condition_on_observations(Y_, X=dev["obs"][i_], L=problems.operator_of(o_.op, wl.d), b=b_)
            return u_

        def sequence():
            # notebook 0001 cell 22 (`_conditional.py:193-197,223-231`): u.mean(grid), then u.std(grid)
            u_ = conditioned()
            m_ = u_.mean(dev["test"])
            s_ = u_.std(dev["test"])
            return m_, s_

        saved = (lp.config.lazy_factorization, lp.config.variance_with_mean)
        try:
            lp.config.lazy_factorization, lp.config.variance_with_mean = False, False
            eager_ms, _ = timed(step)
            seq_eager_ms, (m_e, s_e) = timed(sequence)
            lp.config.variance_with_mean = True
            seq_eager_vwm_ms, (m_v, s_v) = timed(sequence)
            lp.config.lazy_factorization, lp.config.variance_with_mean = True, True
            seq_lazy_ms, (m_l, s_l) = timed(sequence)
            lp.config.lazy_factorization, lp.config.variance_with_mean = saved
            # end to end from HOST-resident inputs ("end-to-end = sum incl. H2D of coordinates and D2H of
            # mean/var"): every point set handed over as a NumPy array / TensorProductGrid per step, uploaded inside the step
            e2e_ms, (_, m_h, v_h) = timed(lambda: problems.condition_and_predict(wl, prior=prior))
        finally:
            lp.config.lazy_factorization, lp.config.variance_with_mean = saved
        sd = np.sqrt(np.maximum(var, 0.0))
        scale_m, scale_s = float(np.max(np.abs(mean))), float(np.max(sd))
        modes = {"timed_region": {"lazy_factorization": bool(bench_lazy), "ms_per_step": dt / args.steps * 1e3},
                 "eager_default": {"lazy_factorization": False, "ms_per_step": eager_ms, "steps": k_m,
                                   "note": "the package's DEFAULT: every condition_on_observations reads the factorisation status back and raises "
                                           "LinAlgError itself, as the reference does (_conditional.py:44,83,280-282); predict is a second pipeline"}}
        ref_seq = {
            "calls": "u = prior.condition_on_observations(...) x blocks; u.mean(x); u.std(x)   (experiments/0001_poisson_dirichlet_2d.ipynb cell 22)",
            "default_mode_ms": seq_eager_ms, "default_mode_variance_with_mean_ms": seq_eager_vwm_ms, "lazy_mode_ms": seq_lazy_ms,
            "predict_ms": dt / args.steps * 1e3,
            "overhead_vs_predict_default_mode": seq_eager_ms / eager_ms - 1.0,
            "overhead_vs_predict_lazy_mode": seq_lazy_ms / (dt / args.steps * 1e3) - 1.0,
            "steps": k_m,
            "mean_vs_predict_rel": float(max(np.max(np.abs(m_e - mean)), np.max(np.abs(m_l - mean)), np.max(np.abs(m_v - mean))) / scale_m),
            # (compared on the VARIANCE scale: where the posterior variance is ~0 -- c2: 1e-7 of the prior's -- a rounding error of
            #  the variance is amplified by 1 / (2 std) in the standard deviation, for `predict` and the sequence alike)
            "var_from_std_vs_predict_rel": float(max(np.max(np.abs(s_e**2 - np.maximum(var, 0.0))), np.max(np.abs(s_l**2 - np.maximum(var, 0.0))))
                                                 / max(float(np.max(np.abs(var))), 1e-300)),
            "std_vs_predict_rel": float(max(np.max(np.abs(s_e - sd)), np.max(np.abs(s_l - sd))) / scale_s),
            "note": "default mode: mean(x) solves for the representer weights (two triangular solves with one right-hand side), std(x) "
                    "assembles the cross-covariance again and runs the blocked forward substitution; with lp.config.variance_with_mean "
                    "mean(x) computes the variance in the same pass -- in lazy mode the fused factor-and-predict pipeline -- and std(x) "
                    "is served from it",
        }
        e2e = {"e2e_with_h2d_ms": e2e_ms,
               "h2d_bytes_per_step": int(sum(o_.X.nbytes + o_.Y.nbytes for o_ in wl.observations) + wl.Xtest.nbytes),
               "d2h_bytes_per_step": int(2 * 8 * wl.Xtest.shape[0]),
               "mean_vs_resident_rel": float(np.max(np.abs(m_h - mean)) / scale_m),
               "var_vs_resident_rel": float(np.max(np.abs(v_h - var)) / max(float(np.max(np.abs(var))), 1e-300)),
               "note": "same step with every point set handed over as a host array (TensorProductGrid for the grids) and uploaded "
                       "inside the step; `value` is the resident-input rate (task contract), this is the PCIe-inclusive time"}
    # ---- per-kernel HIP-event timing (same process, same workload, right after the timed
    #      region: event records between launches cost ~25 % wall time, so they stay out of it) ----
    # (at least ~2.5 s of device time per pass on one GPU: more samples per kernel, and a GPU section long enough for an
    #  external activity sampler to see whatever K the caller chose)
    prof_steps = max(1, min(args.steps, 3))
    if os.environ.get("LPGP_BENCH_PROF_STEPS"):
        prof_steps = max(1, int(os.environ["LPGP_BENCH_PROF_STEPS"]))          # (rocprofv3 counter passes: keep them short)
    elif world == 1:
        prof_steps = max(prof_steps, min(200, int(2.5 / max(dt / args.steps, 1e-4))))

    def profiled(which):
        ctx.profile_reset()
        ctx.profile_enable(which)
        for _ in range(prof_steps):
            step()
        ctx.sync()
        out_ = ctx.profile_get()
        ctx.profile_enable(False)
        for p_ in out_.values():
            for k_ in ("ms", "launches", "flops", "bytes"):
                p_[k_] = p_[k_] * (args.steps / prof_steps)
        return out_

    prof_syrk = profiled([ROOFLINE_SLOT])        # dominant kernel alone: least perturbation
    ctx.dist_stats(reset=True)
    prof = profiled(True)                        # every kernel (table)
    prof[ROOFLINE_SLOT] = prof_syrk[ROOFLINE_SLOT]
    # per-rank communication of the table pass: bytes and seconds inside panel exchanges (HIP events on the stream
    # the exchange is enqueued on), gathered so that the driver's scaling curve can be read
    cs = ctx.dist_stats()
    comm_rows = comm.gather([float(cs["bytes_sent"]) / prof_steps, float(cs["bytes_received"]) / prof_steps,
                             float(prof["comm"]["ms"]) / args.steps])

    # ---- real parity at N > 1: the oracle of the timed workload on rank 0's host cores, started NOW on a thread
    #      (LAPACK releases the GIL) so that it runs beside the extras below; joined before the line is printed ----
    oracle_box = {}
    oracle_thread = None
    if distributed and rank == 0 and not os.environ.get("LPGP_BENCH_NO_ORACLE"):
        def _oracle():
            try:
                import psutil
                from oracle import workloads as owl
                workers = max(1, min(16, (os.cpu_count() or 1) // 8))
                need = owl.host_memory_needed(wl, 1024, workers)
                if psutil.virtual_memory().available < 1.3 * need:
                    oracle_box["skipped"] = f"the oracle at N_tot = {wl.n_total} needs {need / 1e9:.0f} GB of host memory"
                    return
                oracle_box["ref"] = owl.run_in_place(wl, chunk=1024, workers=workers)
            except Exception as exc:            # noqa: BLE001 (reported in the line)
                oracle_box["skipped"] = f"{type(exc).__name__}: {exc}"
        oracle_thread = threading.Thread(target=_oracle, daemon=True)
        oracle_thread.start()

    # ---- the BASELINE configuration named for this GPU count, on the same ranks (collective: every rank runs it) ----
    extra = os.environ.get("LPGP_BENCH_EXTRA")
    if extra is None:
        extra = {8: "c4", 4: "c5", 2: "c5"}.get(world, "none") if (distributed and weak) else "none"
    extra_names = [e for e in extra.split(",") if e and e != "none"]
    del dev, prior
    gc.collect()

    def run_extras():
        configs_ = {}
        for name in extra_names:
            w2 = {"c4": lambda: problems.poisson_2d(n_side=256, m_side=128), "c5": problems.heat_1d,
                  "c3": problems.poisson_2d, "c2": problems.poisson_1d}[name]()
            lp.config.gram_capacity_hint = w2.n_total
            dev2, prior2 = problems.upload(w2), problems.build_prior(w2)
            problems.condition_and_predict(w2, prior=prior2, device_arrays=dev2)             # untimed: allocation
            ctx.dist_stats(reset=True)
            k2 = 2 if name == "c4" else 3
            dt2, (u2, mean2, var2) = timed_steps(w2, prior2, dev2, k2)
            cs2 = ctx.dist_stats()
            # the configuration's own roofline kernel and communication time: one more step under HIP events (outside its timed steps)
            ctx.profile_reset()
            ctx.profile_enable([ROOFLINE_SLOT, "comm"])
            problems.condition_and_predict(w2, prior=prior2, device_arrays=dev2)
            ctx.sync()
            pr2 = ctx.profile_get()
            ctx.profile_enable(False)
            comm2 = comm.gather([float(cs2["bytes_sent"]) / k2, float(cs2["bytes_received"]) / k2, float(pr2["comm"]["ms"]) * 1e-3])
            props = parity_by_properties(problems, w2, u2, mean2, var2)
            per_gpu = (world if replicas else 1)
            sy2 = pr2[ROOFLINE_SLOT]
            ach2 = sy2["flops"] / (sy2["ms"] * 1e-3) / 1e12 if sy2["ms"] > 0 else 0.0
            entry = {
                "workload": w2.name, "n_total": int(w2.n_total), "m_predict": int(w2.Xtest.shape[0]), "steps": k2, "n_gpus": world,
                "ms_per_step": dt2 / k2 * 1e3, "value": per_gpu * w2.total_flops() / (dt2 / k2) / 1e9, "unit": "GFLOP/s",
                "frac_of_fp64_mfma_peak_all_gpus": per_gpu * w2.total_flops() / (dt2 / k2) / 1e12 / (FP64_MFMA_PEAK_TFLOPS * world),
                "algorithmic_flops_per_step": w2.total_flops(),
                "roofline": {"kernel": "rank-512 trailing update (this rank's share), HIP events over one extra step", "bound": "mfma",
                             "achieved": ach2, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach2 / FP64_MFMA_PEAK_TFLOPS,
                             "launches_per_step": sy2["launches"], "traffic": None},
                "parity_by_properties": props,
                "comm_per_rank_per_step": None if comm2 is None or not distributed else [
                    {"rank": r_, "bytes_sent": row[0], "bytes_received": row[1], "seconds_in_comm": row[2]} for r_, row in enumerate(comm2)],
            }
            if rank == 0 and name == "c4":
                # the committed full-size oracle posterior of c4 (tests/golden/c4_posterior.npz, tests/golden/make_c4_golden.py)
                try:
                    gold = np.load(os.path.join(ROOT, "tests", "golden", "c4_posterior.npz"))
                    if int(gold["n_total"]) == w2.n_total and int(gold["m"]) == w2.Xtest.shape[0]:
                        entry["parity"] = parity_report(mean2, var2, {"mean": gold["mean"], "var": gold["var"], "seconds": "committed fixture"}, w2)
                        entry["parity"]["oracle"] = "tests/golden/c4_posterior.npz (" + str(gold["provenance"]) + ")"
                except Exception as exc:        # noqa: BLE001
                    entry["parity"] = {"skipped": f"{type(exc).__name__}: {exc}"}
            elif rank == 0 and name == "c5" and not os.environ.get("LPGP_BENCH_NO_ORACLE"):
                try:
                    from oracle import workloads as owl
                    if oracle_thread is not None:
                        oracle_thread.join()                 # one LAPACK job at a time on the host
                    ref2 = owl.run_in_place(w2, chunk=1024, workers=max(1, min(16, (os.cpu_count() or 1) // 8)))
                    entry["parity"] = parity_report(mean2, var2, ref2, w2)
                    entry["parity"]["oracle"] = "oracle.workloads.run_in_place on rank 0's host cores, in this run"
                except Exception as exc:        # noqa: BLE001
                    entry["parity"] = {"skipped": f"{type(exc).__name__}: {exc}"}
            configs_[name] = entry
            del u2, dev2, prior2
            gc.collect()
        return configs_

    # Extras BEFORE the line only if the line can still be printed inside LPGP_BENCH_LINE_DEADLINE_S (default 600 s after the
    # start of this process) by an estimate from the measured step; otherwise the line goes out first and the extras follow
    # as a second JSON object on stderr -- the headline never waits for a configuration nobody has timed on real links.
    deadline_s = float(os.environ.get("LPGP_BENCH_LINE_DEADLINE_S", "600"))
    est_extras_s = 0.0
    for name in extra_names:
        f2 = {"c4": 1.73e14, "c5": 1.73e13, "c3": 2.78e12, "c2": 2.5e11}.get(name, 1e13)
        est_extras_s += (f2 / max(wl.total_flops(), 1.0)) * (dt / args.steps) * 5.0 * 1.5 + 10.0 + (40.0 if name == "c5" else 0.0)
    extras_first = comm.bcast((time.time() - T_START) + est_extras_s < deadline_s if rank == 0 else None) if extra_names else True
    configs = run_extras() if (extra_names and extras_first) else {}

    if rank != 0:
        if extra_names and not extras_first:
            run_extras()
        comm.close()
        return
    ms_per_step = dt / args.steps * 1e3
    flops = wl.total_flops()
    # distributed: all ranks work on ONE problem; replicas: one problem per rank
    value = (world if replicas else 1) * flops / (dt / args.steps) / 1e9

    sha = csrc_sha16()
    syrk = prof[ROOFLINE_SLOT]
    asm = prof["assemble"]
    achieved = syrk["flops"] / (syrk["ms"] * 1e-3) / 1e12 if syrk["ms"] > 0 else 0.0
    roof_symbol = ctx.roofline_kernel_symbol() if hasattr(ctx, "roofline_kernel_symbol") else "gemm_f64_kernel<false, false, 1>"
    out = {
        "metric": "condition+predict fp64 GFLOP/s (algorithmic), N=16384 Poisson-2D",
        "value": value,
        "unit": "GFLOP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        # the N = 1, 2, 4, 8 series of the default mode grows the problem with N (flops per GPU fixed): weak scaling, and the
        # N = 1 line is the first point of that series; "strong" only for LPGP_BENCH_STRONG=1 / an explicit --n-side
        "scaling": "strong" if (strong or (distributed and not weak)) else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": wl.name,
            "n_collocation": int(wl.observations[-1].X.shape[0]) if args.workload in ("poisson1d", "poisson1d_c1") else
                             int(max(o.X.shape[0] for o in wl.observations)),
            "n_other_observations": int(wl.n_total - max(o.X.shape[0] for o in wl.observations)),
            "n_total": wl.n_total,
            "m_predict": int(wl.Xtest.shape[0]),
            "noise_var_per_block": [None if o.noise_var is None else float(o.noise_var) for o in wl.observations],
            "algorithmic_flops_per_step": flops,
            "multi_gpu": ((dist_note or "single GPU") if world == 1 else
                          (dist_note or "independent replicas (one problem per GPU)") if replicas else
                          (dist_note or "") +
                          f"one problem, Gram matrix / factor sharded in 2-D block-cyclic tiles (blocks of 512) over a "
                          f"{ctx.grid[0]} x {ctx.grid[1]} process grid, "
                          + {"host": "HOST-STAGED panel exchange (bring-up transport, not a benchmark configuration)",
                             "ipc": "panel gather by direct-peer pushes into IPC-mapped windows (device-to-device copies, barriers over the control plane)",
                             "rccl": "panel gather by grouped RCCL point-to-point sends"
                                     + (" -- OVER LOOPBACK SOCKETS between ranks that share one GPU (LPGP_BENCH_RCCL_LOOPBACK=1: "
                                        "bring-up aid, not a benchmark configuration)" if loopback else "")}[transport]
                          + ", prediction points sharded over the ranks, factor streamed for the solves"
                          + (f"; weak scaling: grid side {n_side} so that flops per GPU equal c3's" if weak else "")),
            "rccl_ranks": world if (distributed and transport == "rccl") else 0,
            "transport": None if not distributed else transport,
            "process_grid": None if not distributed else list(ctx.grid),
            "link_probe": link_probe,
            "trials": None if trials is None else {"variants": trials, "chosen": chosen, "skipped": skipped_trials,
                                                   "calibration_seconds": calibration_s, "budget_seconds": budget_s,
                                                   "note": "two timed steps of the bench workload per variant, during warm-up; the fastest runs the timed region; "
                                                           "link probe and trials share LPGP_BENCH_BUDGET_S"},
            "comm_per_rank_per_step": None if not distributed else [
                {"rank": r, "bytes_sent": row[0], "bytes_received": row[1], "seconds_in_comm": row[2] * 1e-3}
                for r, row in enumerate(comm_rows)],
            "device": info["name"].strip(),
            "csrc_sha16": sha,
        },
        "roofline": {
            "kernel": f"{roof_symbol} (SYRK: rank-512 trailing update of the blocked Cholesky, remainder half of the look-ahead split, ~90 % of the factorisation flops)",
            "bound": "mfma",
            "achieved": achieved,
            "peak": FP64_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
            **pmc_traffic(sha, roof_symbol, tag=profile_tag(args.workload, args.n_side, args.m_side) if not distributed else None),
            "launches_per_step": syrk["launches"] / max(args.steps, 1),
            "avg_launch_ms": syrk["ms"] / max(syrk["launches"], 1),
            "note": ("in the fused factor-and-predict pipeline this kernel shares the chip with the riding substitution's updates "
                     "(gemm3_f64_kernel<true, 0>) for most of its launches: its own in-situ rate is what `achieved` reports; the "
                     "rate of the whole step -- every algorithmic flop over the step time -- is `step_frac_of_peak`, and "
                     "`modes.eager_default` is the two-pipeline schedule in which it runs beside the panel chain only"),
            "step_frac_of_peak": (flops / (dt / args.steps) / 1e12) / (FP64_MFMA_PEAK_TFLOPS * (1 if not replicas else 1)) / (world if distributed else 1),
        },
        "kernels": {
            name: {
                "ms_per_step": p["ms"] / args.steps,
                "launches_per_step": p["launches"] / args.steps,
                **({"tflops": p["flops"] / (p["ms"] * 1e-3) / 1e12} if p["flops"] > 0 and p["ms"] > 0 else {}),
                **({"gbps": p["bytes"] / (p["ms"] * 1e-3) / 1e9} if p["bytes"] > 0 and p["ms"] > 0 else {}),
            }
            for name, p in prof.items()
        },
        "phase_ms": {"condition": max(r_[0] for r_ in phase_rows), "predict": max(r_[1] for r_ in phase_rows),
                     "note": "host stamps around the conditioning chain (assembly, block appends, factorisation) and the prediction "
                             "(cross-covariance, streamed solve, read-outs) of two extra steps with a device synchronisation between the phases "
                             "(an un-instrumented step has none: the factorisation is enqueued and the prediction follows it on the device, so "
                             "the two phases add up to more than ms_per_step); best of two, max over ranks"},
        "posterior": {"mean_max": float(np.max(mean)), "var_min": float(np.min(var)), "var_max": float(np.max(var))},
    }
    out["mode"] = mode_name
    out["two_pipeline_ms_per_step"] = two_pipeline_ms
    out["two_pipeline_value"] = None if two_pipeline_ms is None else (world if replicas else 1) * flops / (two_pipeline_ms * 1e-3) / 1e9
    if tp_syrk is not None and tp_syrk["ms"] > 0:
        ach_tp = tp_syrk["flops"] / (tp_syrk["ms"] * 1e-3) / 1e12
        out["roofline"]["two_pipeline_mode"] = {
            "achieved": ach_tp, "frac": ach_tp / FP64_MFMA_PEAK_TFLOPS, "launches_per_step": tp_syrk["launches"],
            "note": "the same kernel in the two-pipeline schedule (package default; every N > 1 line): beside the panel chain only, not beside the riding substitution's updates"}
    out["config"]["lazy_factorization"] = bool(lp.config.lazy_factorization) and not distributed
    out["config"]["fused_factor_and_predict"] = bool(lp.config.lazy_factorization) and not distributed
    if modes is not None:
        out["modes"] = modes
        out["reference_sequence"] = ref_seq
        out["e2e_with_h2d_ms"] = e2e["e2e_with_h2d_ms"]
        out["e2e"] = e2e
    if configs:
        out["configs"] = configs
    # assembly kernels, one entry per kernel symbol (HBM-write bound by design; bytes = entries stored x 8,
    # lower triangle only for diagonal blocks)
    asm_kernels = {"assemble": "assemble_fast_kernel<D,N0,N1,MODE> / assemble_kernel<D> (one fused evaluation per entry: boundary / cross blocks, cross-covariance, every block of scattered points)",
                   "assemble_grid": "kron2_kernel<NU> / kron_expand_kernel (tensor-grid blocks: Kronecker expansion of 1-D kernel matrices)"}
    out["roofline_assembly"] = {}
    for key, label in asm_kernels.items():
        p = prof[key]
        if p["ms"] > 0 and p["bytes"] > 0:
            gbps = p["bytes"] / (p["ms"] * 1e-3) / 1e9
            out["roofline_assembly"][key] = {
                "kernel": label, "bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": gbps / HBM_PEAK_GBPS, "algorithmic_bytes_per_step": p["bytes"] / args.steps,
                "launches_per_step": p["launches"] / args.steps,
            }
    del asm
    if args.workload != "poisson2d" or wl.n_total != 16896:
        out["metric"] = f"condition+predict fp64 GFLOP/s (algorithmic), {wl.name} (N_tot={wl.n_total})"
    if cpu_json is not None:
        out["cpu_baseline"] = cpu_json
    if ref is not None:
        out["parity"] = parity_report(mean, var, ref, wl)
    if oracle_thread is not None:
        # the timed workload of this multi-GPU run against the oracle (rank 0's host cores, beside the extras)
        oracle_thread.join(max(30.0, deadline_s - (time.time() - T_START)) if (extra_names and not extras_first) else None)
        if "ref" in oracle_box:
            out["parity"] = parity_report(mean, var, oracle_box["ref"], wl)
            out["parity"]["oracle"] = "oracle.workloads.run_in_place on rank 0's host cores, in this run"
        else:
            out["parity"] = {"skipped": oracle_box.get("skipped", "the oracle had not finished when the line was due")}
    out["config"]["calibration_seconds"] = calibration_s
    assert out["n_gpus"] == args.gpus, (out["n_gpus"], args.gpus)
    _libc.fflush(None)
    sys.stdout.flush()
    print(json.dumps(out), flush=True)
    if extra_names and not extras_first:
        late = run_extras()
        sys.stderr.write(json.dumps({"configs_after_the_line": late, "note": "the BASELINE configuration for this GPU count, run after the line "
                                     "was printed (LPGP_BENCH_LINE_DEADLINE_S)"}) + "\n")
        sys.stderr.flush()
    comm.close()


if __name__ == "__main__":
    main()
