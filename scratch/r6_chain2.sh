#!/bin/bash
# round 6: two-kernel resident chain (LPGP_CHAIN_RESIDENT2 = max tile rows below; 0 off) -- tests, then c2 / c3 / c1 / small A/B
mkdir -p gpurun_out
export LPGP_BENCH_NO_MODES=1
{
timeout 900 python -m pytest tests/test_gpu_chain.py -x -q -m gpu 2>&1 | tail -6
for rep in 1 2; do
for R2 in 0 40 56 64; do
  for w in poisson1d poisson2d; do
    echo -n "rep=$rep R2=$R2 $w: "
    LPGP_CHAIN_RESIDENT2=$R2 timeout 600 python bench.py --workload $w --steps 30 --warmup 4 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_step %.3f two_pipeline %.3f' % (d['ms_per_step'], d['two_pipeline_ms_per_step']))"
  done
done
done
for R2 in 0 56; do
  echo -n "R2=$R2 heat1d: "
  LPGP_CHAIN_RESIDENT2=$R2 timeout 900 python bench.py --workload heat1d --steps 8 --warmup 2 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_step %.3f two_pipeline %.3f' % (d['ms_per_step'], d['two_pipeline_ms_per_step']))"
done
} 2>&1 | tee gpurun_out/r6_chain2.txt
