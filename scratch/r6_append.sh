#!/bin/bash
mkdir -p gpurun_out
export LPGP_BENCH_NO_MODES=1
{
timeout 900 python -m pytest tests/test_gpu_matrix_free.py tests/test_gpu_parity.py tests/test_gpu_fused.py -x -q -m gpu 2>&1 | tail -4
for rep in 1 2 3; do
for v in 0 1; do
  for w in poisson2d poisson1d; do
  echo -n "rep=$rep append_split=$v $w: "
  LPGP_APPEND_SPLIT=$v timeout 600 python bench.py --workload $w --steps 30 --warmup 4 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_step %.3f two_pipeline %.3f frac %.3f' % (d['ms_per_step'], d['two_pipeline_ms_per_step'], d['roofline']['frac']))"
  done
done
done
for v in 0 1 0 1; do
  echo -n "append_split=$v heat1d: "; LPGP_APPEND_SPLIT=$v timeout 900 python bench.py --workload heat1d --steps 8 --warmup 2 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_step %.3f two_pipeline %.3f' % (d['ms_per_step'], d['two_pipeline_ms_per_step']))"
done
} 2>&1 | tee gpurun_out/r6_append.txt
