#!/bin/bash
# A/B of the resident single-vector solve; results under gpurun_out/r6_trsv.txt
mkdir -p gpurun_out
{
for c in c1 ragged c2 c5s c3; do
  LPGP_TRSV_RESIDENT=0 timeout 300 python scratch/r6_trsv.py $c
  LPGP_TRSV_RESIDENT=1 timeout 300 python scratch/r6_trsv.py $c
  python - <<PY
import numpy as np
a=np.load("gpurun_out/r6_w_${c}_0.npy"); b=np.load("gpurun_out/r6_w_${c}_1.npy")
print("  $c: resident vs per-tile max rel diff", float(np.max(np.abs(a-b))/np.max(np.abs(a))))
PY
done
} 2>&1 | tee gpurun_out/r6_trsv.txt
