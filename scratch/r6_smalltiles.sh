#!/bin/bash
# launches of at most LPGP_SMALL_TILES_MAX 128 x 128 tiles go to the 64 x 64-tile kernel (workgroups that live a quarter as long): c2 / c3
mkdir -p gpurun_out
export LPGP_BENCH_NO_MODES=1
{
for rep in 1 2; do
for v in 256 512 1024 1700 2400; do
  for w in poisson1d poisson2d; do
  echo -n "rep=$rep small_tiles_max=$v $w: "
  env LPGP_SMALL_TILES_MAX=$v timeout 600 python bench.py --workload $w --steps 30 --warmup 4 --no-cpu 2>gpurun_out/r6_smalltiles.err | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('ms_per_step %.3f frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))
except Exception as e:
    print('FAILED', open('gpurun_out/r6_smalltiles.err').read()[-300:].replace(chr(10), ' | '))"
  done
done
done
} 2>&1 | tee gpurun_out/r6_smalltiles.txt
