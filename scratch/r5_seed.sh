#!/bin/bash
cd "$GRAFT_REPO_ROOT"
S=${1:-2098}
run() { echo "== $*"; env "$@" LPGP_RANDOM_SEEDS=$S:$((S+1)) timeout 300 python -m pytest tests/test_gpu_random.py -q -m gpu -x -k "test_random_problem" 2>&1 | grep -E "^E  +Assertion|passed|failed" | head -3; }
run A=1
run LPGP_CHAIN_RESIDENT=-1
run LPGP_RIDE_VCHAIN=0
run LPGP_ASM_BATCH=0
run LPGP_FUSED_SOLVE=0
run LPGP_NB=256
