#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
export LPGP_BENCH_NO_MODES=1
run() { python3 bench.py "$@" --no-cpu 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
for rep in 1 2; do
for v in 96 0; do
  export LPGP_RIDE_VCHAIN=$v
  echo "vchain=$v c3 $(run --steps 10 --warmup 3)  c2 $(run --workload poisson1d --steps 30 --warmup 3)"
done
done
export LPGP_RIDE_VCHAIN=96
echo "c5 $(run --workload heat1d --steps 5 --warmup 2)"
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_chain.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -2
