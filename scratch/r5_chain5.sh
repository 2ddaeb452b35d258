#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r5_13; rm -rf $D; mkdir -p $D
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
STEPS=50
for wl in "--workload poisson1d_c1" "--n-side 32 --m-side 16" "--n-side 48 --m-side 24" "--workload heat_reference" "--n-side 64 --m-side 32"; do
  WL="$wl"; tag=$(echo $wl | tr -d ' -' | cut -c1-18)
  b ${tag}_default LPGP_X=1
  b ${tag}_same0_g100 LPGP_RIDE_GATE_PCT=100 LPGP_RIDE_SAME_STREAM_MAX_TILES=0
  b ${tag}_same0_g65 LPGP_RIDE_GATE_PCT=65 LPGP_RIDE_SAME_STREAM_MAX_TILES=0
done
STEPS=20
WL=""
for g in 50 60 65 70 75; do b c3_gate$g LPGP_RIDE_GATE_PCT=$g; done
WL="--workload heat1d"
STEPS=10
for g in 100 65; do b c5_gate$g LPGP_RIDE_GATE_PCT=$g; done
