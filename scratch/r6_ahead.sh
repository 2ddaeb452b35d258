#!/bin/bash
mkdir -p gpurun_out
export LPGP_BENCH_NO_MODES=1
{
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_fused.py -x -q -m gpu 2>&1 | tail -5
for rep in 1 2 3; do
for v in 0 1; do
  for w in poisson1d poisson2d; do
  echo -n "rep=$rep chain_ahead=$v $w: "
  LPGP_CHAIN_AHEAD=$v timeout 600 python bench.py --workload $w --steps 30 --warmup 4 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_step %.3f two_pipeline %s frac %.3f' % (d['ms_per_step'], d['two_pipeline_ms_per_step'], d['roofline']['frac']))"
  done
done
done
} 2>&1 | tee gpurun_out/r6_ahead.txt
