#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r05_suite; mkdir -p $D
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > $D/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -6 $D/gpu_suite.log
