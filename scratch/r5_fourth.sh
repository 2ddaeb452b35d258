#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r5_05; rm -rf $D; mkdir -p $D
b() { # name, env...
  local name=$1; shift
  env "$@" LPGP_BENCH_NO_MODES=1 python bench.py --steps $STEPS --warmup 2 --no-cpu $WL > $D/$name.json 2> $D/$name.err
  python - "$D/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms_per_step", round(d["ms_per_step"],3), "roof", round(d["roofline"]["frac"],3))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
STEPS=10
WL="--workload heat1d"
b c5_eager LPGP_BENCH_EAGER=1
for g in 100 85 70 50 35; do b c5_g$g LPGP_RIDE_GATE_PCT=$g; done
WL="--workload poisson1d"
b c2_eager LPGP_BENCH_EAGER=1
for g in 100 75 50 35; do b c2_g$g LPGP_RIDE_GATE_PCT=$g; done
WL="--workload scattered2d"
b sc_eager LPGP_BENCH_EAGER=1
for g in 100 50; do b sc_g$g LPGP_RIDE_GATE_PCT=$g; done
STEPS=3
WL="--n-side 256 --m-side 128"
b c4_eager LPGP_BENCH_EAGER=1
for g in 100 70 50; do b c4_g$g LPGP_RIDE_GATE_PCT=$g; done
python scratch/small_sizes.py 2>&1 | head -8 > $D/small_sizes.txt; cat $D/small_sizes.txt
( time timeout 1500 python -m pytest tests -q -m gpu -x ) > $D/pytest_full.log 2>&1; tail -6 $D/pytest_full.log
