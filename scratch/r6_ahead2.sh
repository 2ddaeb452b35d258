#!/bin/bash
mkdir -p gpurun_out
export LPGP_BENCH_NO_MODES=1
{
for rep in 1 2 3; do
for v in "0 0" "1 0" "1 8" "1 12" "1 16"; do
  set -- $v
  for w in poisson1d poisson2d; do
  echo -n "rep=$rep chain_ahead=$1 min_rows=$2 $w: "
  LPGP_CHAIN_AHEAD=$1 LPGP_CHAIN_AHEAD_MIN_ROWS=$2 timeout 600 python bench.py --workload $w --steps 30 --warmup 4 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_step %.3f' % (d['ms_per_step']))"
  done
done
done
for v in "0 0" "1 12" "0 0" "1 12"; do set -- $v; echo "== ahead=$1 min_rows=$2"; LPGP_CHAIN_AHEAD=$1 LPGP_CHAIN_AHEAD_MIN_ROWS=$2 timeout 600 python scratch/small_sizes.py 2>&1 | head -5 | cut -c1-90; done
} 2>&1 | tee gpurun_out/r6_ahead2.txt
