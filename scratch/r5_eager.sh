#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python3 scratch/small_sizes.py 2>&1 | head -5
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py -x -q -m gpu 2>&1 | tail -2
