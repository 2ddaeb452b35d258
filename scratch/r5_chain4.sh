#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r5_12; rm -rf $D; mkdir -p $D
export LPGP_CHAIN_RESIDENT=32
b() { # name, env...
  local name=$1; shift
  env "$@" LPGP_BENCH_NO_MODES=1 timeout 600 python bench.py --steps $STEPS --warmup 3 --no-cpu $WL > $D/$name.json 2> $D/$name.err
  python - "$D/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms_per_step", round(d["ms_per_step"],3))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
STEPS=30
WL="--workload poisson1d"
for g in 50 65 80 100; do b c2_gate$g LPGP_RIDE_GATE_PCT=$g; done
WL="--n-side 64 --m-side 32"
for g in 50 65 80 100; do b p64_gate$g LPGP_RIDE_GATE_PCT=$g; done
WL="--workload heat_reference"
for g in 50 80 100; do b heatref_gate$g LPGP_RIDE_GATE_PCT=$g; done
WL="--n-side 32 --m-side 16"
for g in 50 100; do b p32_gate$g LPGP_RIDE_GATE_PCT=$g; done
b p32_gate100_same0 LPGP_RIDE_GATE_PCT=100 LPGP_RIDE_SAME_STREAM_MAX_TILES=0
WL=""
for g in 50 65; do b c3_gate$g LPGP_RIDE_GATE_PCT=$g; done
