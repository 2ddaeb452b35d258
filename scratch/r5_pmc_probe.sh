#!/bin/bash
# does a follower kernel survive serialised dispatches?  (c1-sized problem under rocprofv3 --pmc, with and without the follower)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r05_pmc_probe; rm -rf $D; mkdir -p $D
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/on -- python3 scratch/small_trace.py p32 5 > $D/on.log 2>&1; echo "follower on: rc=$?"; tail -2 $D/on.log
export LPGP_RIDE_VCHAIN=0
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/off -- python3 scratch/small_trace.py p32 5 > $D/off.log 2>&1; echo "follower off: rc=$?"; tail -2 $D/off.log
rm -rf $D/on $D/off
unset LPGP_RIDE_VCHAIN
python3 scratch/small_trace.py c1 300 | tail -1; python3 scratch/small_trace.py p32 200 | tail -1
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_chain.py -x -q -m gpu 2>&1 | tail -1
