#!/bin/bash
# EXPERIMENT driver: thresholds for the conditional refinement
mkdir -p gpurun_out/$1
for thr in 0 30 100 300 1000 1e30; do
  echo "== thr $thr" | tee -a gpurun_out/$1/thr.txt
  LPGP_X_NOREFINE=1 LPGP_X_REFINE_THR=$thr python3 scratch/refine_thr.py poisson2d poisson1d heat c1 heatref 2>&1 | grep -v Warning | tee -a gpurun_out/$1/thr.txt
done
python3 - <<'PY' | tee -a gpurun_out/$1/thr.txt
import numpy as np
for name in ["poisson2d","poisson1d","heat","c1","heatref"]:
    m0=np.load(f"/tmp/x_{name}_0_m.npy"); v0=np.load(f"/tmp/x_{name}_0_v.npy")
    for thr in ["30","100","300","1000","1e30"]:
        m=np.load(f"/tmp/x_{name}_{thr}_m.npy"); v=np.load(f"/tmp/x_{name}_{thr}_v.npy")
        print(f"{name:10s} thr {thr:>5s}: mean {np.abs(m-m0).max()/np.abs(m0).max():.2e}  var {np.abs(v-v0).max()/np.abs(v0).max():.2e}")
PY
