#!/bin/bash
# python bench.py --gpus 8 without a launcher: eight ranks on the ONE GPU of the box, RCCL over loopback sockets -- the code path
# of the driver's 8-GPU command end to end (calibration budget, trials, timed region, oracle parity on rank 0, c4 extra), not its rates
set -u
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r06_lines; mkdir -p $D
export OPENBLAS_NUM_THREADS=32
T0=$(date +%s)
LPGP_DEVICE=0 LPGP_BENCH_RCCL_LOOPBACK=1 timeout 1700 python3 bench.py --gpus 8 --steps 1 --warmup 1 > $D/gpus8.out 2> $D/gpus8.err
RC=$?
T1=$(date +%s)
{ echo "# LPGP_DEVICE=0 LPGP_BENCH_RCCL_LOOPBACK=1 python3 bench.py --gpus 8 --steps 1 --warmup 1   (no launcher; eight ranks on the one GPU, RCCL over loopback sockets:"
  echo "# the code path of the driver's 8-GPU run -- link probe + trials inside LPGP_BENCH_BUDGET_S, weak-scaled point, oracle parity on rank 0's host cores, c4 on the same ranks -- not its rates)"
  echo "# rc=$RC wall=$((T1-T0))s"
  python3 - <<'PY'
import json
try:
    line=[l for l in open('gpurun_out/r06_lines/gpus8.out') if l.startswith('{')][-1]
    d=json.loads(line)
    for k in ("link_probe",):
        d["config"].pop(k, None)
    print(json.dumps(d, indent=1)[:9000])
except Exception as e:
    print("no line:", e)
PY
  echo "# stderr (tail):"; tail -c 3000 $D/gpus8.err
} > $D/bench_gpus8_selflaunch.txt
head -3 $D/bench_gpus8_selflaunch.txt | cut -c1-200
