#!/bin/bash
# the substitution's LAST outer updates in slices of K = 512 (LPGP_RIDE_OUTER_SPLIT_ROWS = max tile rows below, 0 = one launch): c3 / c5 / c2
mkdir -p gpurun_out
export LPGP_BENCH_NO_MODES=1
{
LPGP_RIDE_OUTER_SPLIT_ROWS=140 timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2 3; do
for v in 0 20 40 56 72 140; do
  echo -n "rep=$rep split_rows=$v poisson2d: "
  env LPGP_RIDE_OUTER_SPLIT_ROWS=$v timeout 600 python bench.py --workload poisson2d --steps 30 --warmup 4 --no-cpu 2>gpurun_out/r6_outersplit.err | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('ms_per_step %.3f frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))
except Exception as e:
    print('FAILED', open('gpurun_out/r6_outersplit.err').read()[-300:].replace(chr(10), ' | '))"
done
done
for v in 0 40 72; do
  echo -n "split_rows=$v heat1d: "; LPGP_RIDE_OUTER_SPLIT_ROWS=$v timeout 900 python bench.py --workload heat1d --steps 8 --warmup 2 --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_step %.3f' % (d['ms_per_step']))"
  echo -n "split_rows=$v poisson1d: "; LPGP_RIDE_OUTER_SPLIT_ROWS=$v timeout 900 python bench.py --workload poisson1d --steps 30 --warmup 4 --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_step %.3f' % (d['ms_per_step']))"
done
} 2>&1 | tee gpurun_out/r6_outersplit.txt
