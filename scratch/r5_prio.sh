#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export LPGP_BENCH_NO_MODES=1
run() { python3 bench.py "$@" --no-cpu 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
python3 -c "
import ctypes
h=ctypes.CDLL('libamdhip64.so'); lo=ctypes.c_int(); hi=ctypes.c_int(); h.hipDeviceGetStreamPriorityRange(ctypes.byref(lo),ctypes.byref(hi)); print('priority range least',lo.value,'greatest',hi.value)"
for rep in 1 2; do
for v in 0 1 2; do
  export LPGP_UPD_ALL_PRIO=$v
  echo "upd_all prio=$v: c3 $(run --steps 10 --warmup 3)  c2 $(run --workload poisson1d --steps 30 --warmup 3)  c5 $(run --workload heat1d --steps 4 --warmup 1)"
done
done
for g in 55 75 100; do
  echo "prio=2 gate=$g: c3 $(LPGP_UPD_ALL_PRIO=2 LPGP_RIDE_GATE_PCT=$g run --steps 10 --warmup 3)"
done
