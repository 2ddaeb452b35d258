#!/bin/bash
# near / far rows for wide panels (LPGP_CHAIN_FAR = max tile rows below, 0 = tile by tile): tests, then A/B of the bench lines
mkdir -p gpurun_out
export LPGP_BENCH_NO_MODES=1
{
timeout 900 python -m pytest tests/test_gpu_chain.py -x -q -m gpu -k "near_and_far" 2>&1 | tail -5
LPGP_CHAIN_FAR=64 LPGP_CHAIN_FAR_NEAR=8 LPGP_CHAIN_FAR_MIN_NEAR=4 timeout 1200 python -m pytest tests/test_gpu_chain.py tests/test_gpu_fused.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4
for rep in 1 2; do
for v in "0" "64 LPGP_CHAIN_FAR_NEAR=8 LPGP_CHAIN_FAR_MIN_NEAR=4" "64 LPGP_CHAIN_FAR_NEAR=8 LPGP_CHAIN_FAR_MIN_NEAR=4 LPGP_CHAIN_AHEAD_MIN_ROWS=0" "64 LPGP_CHAIN_FAR_NEAR=4 LPGP_CHAIN_FAR_MIN_NEAR=4" "64 LPGP_CHAIN_FAR_NEAR=12 LPGP_CHAIN_FAR_MIN_NEAR=4" "64 LPGP_CHAIN_FAR_NEAR=12 LPGP_CHAIN_FAR_MIN_NEAR=8 LPGP_CHAIN_AHEAD_MIN_ROWS=0" "64 LPGP_CHAIN_FAR_NEAR=16 LPGP_CHAIN_FAR_MIN_NEAR=8"; do
  for w in poisson1d poisson2d; do
  echo -n "rep=$rep far=$v $w: "
  env LPGP_CHAIN_FAR=$v timeout 600 python bench.py --workload $w --steps 30 --warmup 4 --no-cpu 2>gpurun_out/r6_far.err | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('ms_per_step %.3f frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))
except Exception as e:
    print('FAILED', open('gpurun_out/r6_far.err').read()[-300:].replace(chr(10), ' | '))"
  done
done
done
} 2>&1 | tee gpurun_out/r6_far.txt
