#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
export LPGP_BENCH_NO_MODES=1
run() { python3 bench.py "$@" --no-cpu 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
timeout 600 python -m pytest tests/test_gpu_fused.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2 3; do
for v in 8388608 0; do
  export LPGP_RIDE_EARLY_REDUCE=$v
  echo "early_reduce=$v c3 $(run --steps 10 --warmup 3)"
done
done
export LPGP_RIDE_EARLY_REDUCE=8388608
echo "c5 on $(run --workload heat1d --steps 5 --warmup 2)"
export LPGP_RIDE_EARLY_REDUCE=0
echo "c5 off $(run --workload heat1d --steps 5 --warmup 2)"
