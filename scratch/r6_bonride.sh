#!/bin/bash
mkdir -p gpurun_out
export LPGP_BENCH_NO_MODES=1
{
for rep in 1 2; do
for v in 0 1; do
  for w in poisson2d poisson1d; do
  echo -n "rep=$rep b_on_ride=$v $w: "
  LPGP_RIDE_B_ON_RIDE=$v timeout 600 python bench.py --workload $w --steps 30 --warmup 4 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_step %.3f frac %.3f launches %s' % (d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('launches_per_step')))"
  done
done
done
echo -n "b_on_ride=1 heat1d: "; LPGP_RIDE_B_ON_RIDE=1 timeout 900 python bench.py --workload heat1d --steps 8 --warmup 2 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_step %.3f frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"
} 2>&1 | tee gpurun_out/r6_bonride.txt
