#!/bin/bash
# resident-chain limit sweep (one launch per panel where at most R tile rows lie below it)
mkdir -p gpurun_out
export LPGP_BENCH_NO_MODES=1
{
for R in 32 40 48 56; do
  for w in poisson1d poisson2d; do
    echo -n "R=$R $w: "
    LPGP_CHAIN_RESIDENT=$R timeout 600 python bench.py --workload $w --steps 30 --warmup 4 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_step %.3f' % d['ms_per_step'])"
  done
done
} 2>&1 | tee gpurun_out/r6_resident_sweep.txt
