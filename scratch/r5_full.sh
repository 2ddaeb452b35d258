#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r5_14; rm -rf $D; mkdir -p $D
( time timeout 1700 python -m pytest tests -q -m gpu ) > $D/pytest_full.log 2>&1; tail -8 $D/pytest_full.log
