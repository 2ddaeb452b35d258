#!/bin/bash
# two RCCL ranks over loopback on one GPU; rank 1 fails at its third panel step: where does each rank end up?
cd "$GRAFT_REPO_ROOT"
cat > /tmp/fail_probe.py <<'PY'
import os, sys, time, faulthandler
faulthandler.dump_traceback_later(45, exit=True)
sys.path.insert(0, "."); sys.path.insert(0, "linpde-gp_amd")
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
except Exception as exc:
    print("RAISED", comm.rank, type(exc).__name__, f"{time.time() - t0:.1f}s", str(exc)[:400], flush=True)
    os._exit(0)
print("NO ERROR", comm.rank, flush=True)
os._exit(3)
PY
for r in 0 1; do
  RANK=$r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=30071 LPGP_DEVICE=0 LPGP_TEST_FAIL_RANK=1 LPGP_TEST_FAIL_PANEL=2 LPGP_DIST_TIMEOUT_S=20 \
  NCCL_HOSTID=h$r NCCL_SOCKET_IFNAME=lo NCCL_IB_DISABLE=1 NCCL_NET=Socket NCCL_DEBUG=WARN LPGP_FORCE_RCCL=1 LPGP_DIST_TRACE=1 \
  timeout 70 python /tmp/fail_probe.py > gpurun_out/fail_probe_$r.log 2>&1 &
  pids[$r]=$!
done
wait
for r in 0 1; do echo "=== rank $r"; grep -v "alt_rsmi\|^$" gpurun_out/fail_probe_$r.log | tail -n 22 | cut -c1-200; done
