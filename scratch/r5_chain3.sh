#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r5_10; rm -rf $D; mkdir -p $D
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
for r in -1 8 16 24 32 48; do b c2_res$r LPGP_CHAIN_RESIDENT=$r; done
WL=""
for r in -1 8 16 24; do b c3_res$r LPGP_CHAIN_RESIDENT=$r; done
for r in -1 8 16 32 64; do echo "small sizes, resident <= $r"; LPGP_CHAIN_RESIDENT=$r timeout 300 python scratch/small_sizes.py 2>&1 | head -5 | cut -c1-95; done
