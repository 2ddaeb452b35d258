#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export LPGP_BENCH_NO_MODES=1
run() { python3 bench.py "$@" --no-cpu 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
for rep in 1 2; do
for v in 0 40 56 72 200; do
  export LPGP_RIDE_OUTER_TAIL=$v
  echo "outer_tail=$v: c3 $(run --steps 10 --warmup 3)  c5 $(run --workload heat1d --steps 4 --warmup 1)"
done
done
