#!/bin/bash
# The resident solve with strips shared by two workgroups (LPGP_TRSV_SPLIT_MIN); results under gpurun_out/r6_trsv_split.txt
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_trsv.py -x -q 2>&1 | tail -5
for c in c2 c5s c3; do
  for sm in 0 8 16 32 48 64; do
    echo "split_min=$sm"
    LPGP_TRSV_SPLIT_MIN=$sm timeout 300 python scratch/r6_trsv.py $c
  done
done
} 2>&1 | tee gpurun_out/r6_trsv_split.txt
