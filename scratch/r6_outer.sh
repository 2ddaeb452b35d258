#!/bin/bash
# the substitution's outer blocks (K of its far updates = the lifetime of their workgroups): LPGP_RIDE_OUTER_ROWS 0 / 1024 / 2048 (default) / 4096, c3 (+ c5)
mkdir -p gpurun_out
export LPGP_BENCH_NO_MODES=1
{
for rep in 1 2; do
for v in 2048 1024 1536 0 4096; do
  echo -n "rep=$rep outer_rows=$v poisson2d: "
  env LPGP_RIDE_OUTER_ROWS=$v timeout 600 python bench.py --workload poisson2d --steps 30 --warmup 4 --no-cpu 2>gpurun_out/r6_outer.err | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('ms_per_step %.3f frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))
except Exception as e:
    print('FAILED', open('gpurun_out/r6_outer.err').read()[-300:].replace(chr(10), ' | '))"
done
done
} 2>&1 | tee gpurun_out/r6_outer.txt
