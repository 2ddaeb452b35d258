#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r5_20; rm -rf $D; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_kernels.py tests/test_gpu_dist.py -q -m gpu -x > $D/pytest.log 2>&1; echo "pytest rc=$?" >> $D/pytest.log; tail -3 $D/pytest.log
python scratch/small_sizes.py 2>&1 | head -5 | cut -c1-140
