#!/bin/bash
mkdir -p gpurun_out
export LPGP_BENCH_NO_MODES=1
{
LPGP_CHAIN_BLOCK_FIRST=4096 timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_fused.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4
for rep in 1 2; do
for v in 0 64 128 4096; do
  for w in poisson1d poisson2d; do
  echo -n "rep=$rep block_first=$v $w: "
  LPGP_CHAIN_BLOCK_FIRST=$v timeout 600 python bench.py --workload $w --steps 30 --warmup 4 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_step %.3f frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"
  done
done
done
for v in 0 4096; do
  echo -n "block_first=$v heat1d: "; LPGP_CHAIN_BLOCK_FIRST=$v timeout 900 python bench.py --workload heat1d --steps 8 --warmup 2 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_step %.3f' % (d['ms_per_step']))"
done
} 2>&1 | tee gpurun_out/r6_blockfirst.txt
