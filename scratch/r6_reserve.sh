#!/bin/bash
# CUs the update streams leave to the panel chain (LPGP_RESERVE_CUS, default 32; the narrow stream LPGP_RESERVE_CUS_NARROW, default 64): c2 / c3 / c5
mkdir -p gpurun_out
export LPGP_BENCH_NO_MODES=1
{
for rep in 1 2; do
for v in "32" "40" "48" "56" "64" "48 LPGP_RESERVE_CUS_NARROW=96" "64 LPGP_RESERVE_CUS_NARROW=96"; do
  for w in poisson1d poisson2d; do
  echo -n "rep=$rep reserve=$v $w: "
  env LPGP_RESERVE_CUS=$v timeout 600 python bench.py --workload $w --steps 30 --warmup 4 --no-cpu 2>gpurun_out/r6_reserve.err | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('ms_per_step %.3f frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))
except Exception as e:
    print('FAILED', open('gpurun_out/r6_reserve.err').read()[-300:].replace(chr(10), ' | '))"
  done
done
done
} 2>&1 | tee gpurun_out/r6_reserve.txt
