#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r5_07; rm -rf $D; mkdir -p $D
timeout 1500 python -m pytest tests/test_gpu_matrix_free.py tests/test_gpu_notebooks.py tests/test_gpu_bench_line.py tests/test_gpu_fused.py -q -m gpu -s > $D/pytest.log 2>&1; echo "pytest rc=$?" >> $D/pytest.log; grep -E "passed|failed|Error|CG |N = " $D/pytest.log | tail -12
