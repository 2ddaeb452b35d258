#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r5_03; rm -rf $D; mkdir -p $D
timeout 1200 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -q -m gpu > $D/pytest.log 2>&1; echo "pytest rc=$?" >> $D/pytest.log
tail -4 $D/pytest.log
b() { # name, env...
  local name=$1; shift
  env "$@" LPGP_BENCH_NO_MODES=1 python bench.py --steps 20 --warmup 3 --no-cpu $WL > $D/$name.json 2> $D/$name.err
  python - "$D/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms_per_step", round(d["ms_per_step"],3), "roof", round(d["roofline"]["frac"],3))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
WL=""
b c3_eager LPGP_BENCH_EAGER=1
b c3_r1 LPGP_RIDE_OCC3=1
for g in 400 700 1000 1500 2200 3000; do b c3_r1_g$g LPGP_RIDE_OCC3=1 LPGP_RIDE_GATE_US=$g; done
for g in -1 700 1500; do b c3_r1r4_g$g LPGP_RIDE_STREAM=33 LPGP_RIDE_OCC3=1 LPGP_RIDE_GATE_US=$g; done     # 1 + 8*4
for g in -1 700 1500; do b c3_r1r0_g$g LPGP_RIDE_STREAM=1 LPGP_RIDE_OCC3=1 LPGP_RIDE_GATE_US=$g; done       # 1 + 8*0: second = s_outer
for g in 700 1500; do b c3_r1r2_g$g LPGP_RIDE_STREAM=17 LPGP_RIDE_OCC3=1 LPGP_RIDE_GATE_US=$g; done      # 1 + 8*2
b c3_r1_g1000_occ0 LPGP_RIDE_GATE_US=1000
b c3_eager2 LPGP_BENCH_EAGER=1
WL="--workload poisson1d"
b c2_eager LPGP_BENCH_EAGER=1
b c2_r1 LPGP_RIDE_OCC3=1
b c2_r1_g200 LPGP_RIDE_OCC3=1 LPGP_RIDE_GATE_US=200
b c2_r1r4 LPGP_RIDE_STREAM=33 LPGP_RIDE_OCC3=1
b c2_r1r2 LPGP_RIDE_STREAM=17 LPGP_RIDE_OCC3=1
WL="--workload heat1d"
b c5_eager LPGP_BENCH_EAGER=1
b c5_r1 LPGP_RIDE_OCC3=1
b c5_r1_g1500 LPGP_RIDE_OCC3=1 LPGP_RIDE_GATE_US=1500
b c5_r1r4_g1500 LPGP_RIDE_STREAM=33 LPGP_RIDE_OCC3=1 LPGP_RIDE_GATE_US=1500
