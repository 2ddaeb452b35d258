#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r5_17; rm -rf $D; mkdir -p $D
b() { local name=$1; shift
  env "$@" LPGP_BENCH_NO_MODES=1 timeout 600 python bench.py --steps $STEPS --warmup 3 --no-cpu $WL > $D/$name.json 2> $D/$name.err
  python -c "
import json
try:
    d=json.loads(open('$D/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3))
except Exception as e: print('$name FAILED', e)"
}
STEPS=30; WL=""
b c3_a LPGP_X=1
b c3_v2 LPGP_RIDE_STREAM=33
b c3_b LPGP_X=1
b c3_v2b LPGP_RIDE_STREAM=33
b c3_v2_g75 LPGP_RIDE_STREAM=33 LPGP_RIDE_GATE_PCT=75
b c3_v2_g55 LPGP_RIDE_STREAM=33 LPGP_RIDE_GATE_PCT=55
WL="--workload poisson1d"
b c2_a LPGP_X=1
b c2_v2 LPGP_RIDE_STREAM=33
WL="--workload heat1d"; STEPS=10
b c5_a LPGP_X=1
b c5_v2 LPGP_RIDE_STREAM=33
