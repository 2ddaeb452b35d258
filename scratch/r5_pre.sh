#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export LPGP_BENCH_NO_MODES=1
run() { python3 bench.py "$@" --no-cpu 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
for rep in 1 2 3; do
for v in 1 0; do
  export LPGP_RIDE_VCHAIN_PRE=$v
  echo "pre=$v $(python3 scratch/small_trace.py c1 400 | tail -1) $(python3 scratch/small_trace.py p32 300 | tail -1) heat_ref $(run --workload heat_reference --steps 100) c2 $(run --workload poisson1d --steps 30)"
done
done
