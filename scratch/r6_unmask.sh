#!/bin/bash
mkdir -p gpurun_out
export LPGP_BENCH_NO_MODES=1
{
timeout 600 python -m pytest tests/test_gpu_matrix_free.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
for v in 0 1.5 2.5 4; do
  for w in poisson2d; do
  echo -n "rep=$rep unmask_ratio=$v $w: "
  LPGP_UNMASK_RATIO=$v timeout 600 python bench.py --workload $w --steps 30 --warmup 4 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_step %.3f frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"
  done
done
done
for v in 0 2.5; do
  echo -n "unmask_ratio=$v heat1d: "; LPGP_UNMASK_RATIO=$v timeout 900 python bench.py --workload heat1d --steps 8 --warmup 2 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_step %.3f' % (d['ms_per_step']))"
  echo -n "unmask_ratio=$v poisson1d: "; LPGP_UNMASK_RATIO=$v timeout 900 python bench.py --workload poisson1d --steps 30 --warmup 4 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_step %.3f' % (d['ms_per_step']))"
done
} 2>&1 | tee gpurun_out/r6_unmask.txt
