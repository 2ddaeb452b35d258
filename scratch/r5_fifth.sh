#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r5_06; rm -rf $D; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_fused.py -q -m gpu > $D/pytest.log 2>&1; echo "pytest rc=$?" >> $D/pytest.log; tail -4 $D/pytest.log
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
STEPS=3
WL="--n-side 256 --m-side 128"
b c4_eager LPGP_BENCH_EAGER=1
b c4_auto LPGP_X=1
b c4_outer0 LPGP_RIDE_OUTER=0
b c4_auto_g50 LPGP_RIDE_GATE_PCT=50
STEPS=10
WL="--workload heat1d"
b c5_eager LPGP_BENCH_EAGER=1
b c5_auto LPGP_X=1
b c5_outer2048 LPGP_RIDE_OUTER=100 LPGP_NB_OUTER_SOLVE=2048
b c5_outer4096 LPGP_RIDE_OUTER=100
STEPS=20
WL=""
b c3_eager LPGP_BENCH_EAGER=1
b c3_auto LPGP_X=1
b c3_outer2048 LPGP_RIDE_OUTER=64 LPGP_NB_OUTER_SOLVE=2048
WL="--workload poisson1d"
b c2_auto LPGP_X=1
python scratch/small_sizes.py 2>&1 | head -5 > $D/small_sizes.txt; cat $D/small_sizes.txt
