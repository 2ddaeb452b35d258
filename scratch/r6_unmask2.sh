#!/bin/bash
mkdir -p gpurun_out
export LPGP_BENCH_NO_MODES=1
{
for rep in 1 2 3 4; do
for v in 0 2.5 3.5; do
  echo -n "rep=$rep unmask_ratio=$v poisson2d: "
  LPGP_UNMASK_RATIO=$v timeout 600 python bench.py --steps 40 --warmup 4 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_step %.3f frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"
done
done
for rep in 1 2; do for v in 0 2.5; do
  echo -n "rep=$rep unmask_ratio=$v heat1d: "; LPGP_UNMASK_RATIO=$v timeout 900 python bench.py --workload heat1d --steps 8 --warmup 2 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_step %.3f' % (d['ms_per_step']))"
done; done
} 2>&1 | tee gpurun_out/r6_unmask2.txt
