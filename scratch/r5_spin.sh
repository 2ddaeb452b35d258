#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for v in 0 300 2000; do
  export LPGP_SYNC_SPIN_US=$v
  echo "spin=$v $(python3 scratch/small_trace.py c1 300 2>&1 | tail -1) $(python3 scratch/small_trace.py p32 200 2>&1 | tail -1)"
done
done
