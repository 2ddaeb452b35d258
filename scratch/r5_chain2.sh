#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r5_09; rm -rf $D; mkdir -p $D
b() { # name, env...
  local name=$1; shift
  env "$@" LPGP_BENCH_NO_MODES=1 timeout 600 python bench.py --steps $STEPS --warmup 3 --no-cpu $WL > $D/$name.json 2> $D/$name.err
  python - "$D/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms_per_step", round(d["ms_per_step"],3), "roof", round(d["roofline"]["frac"],3))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
STEPS=20
WL="--workload poisson1d"
b c2_res64 LPGP_X=1
b c2_off LPGP_CHAIN_RESIDENT=-1
b c2_res64_eager LPGP_BENCH_EAGER=1
b c2_off_eager LPGP_CHAIN_RESIDENT=-1 LPGP_BENCH_EAGER=1
WL=""
b c3_res64 LPGP_X=1
b c3_off LPGP_CHAIN_RESIDENT=-1
b c3_res32 LPGP_CHAIN_RESIDENT=32
b c3_res96 LPGP_CHAIN_RESIDENT=96
b c3_res140 LPGP_CHAIN_RESIDENT=140
STEPS=10
WL="--workload heat1d"
b c5_res64 LPGP_X=1
b c5_off LPGP_CHAIN_RESIDENT=-1
timeout 300 python scratch/small_sizes.py 2>&1 | head -5 > $D/small_res.txt; cat $D/small_res.txt
LPGP_CHAIN_RESIDENT=-1 timeout 300 python scratch/small_sizes.py 2>&1 | head -5 > $D/small_off.txt; cat $D/small_off.txt
( time timeout 1200 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_kernels.py tests/test_gpu_configs.py tests/test_gpu_random.py tests/test_gpu_golden.py -q -m gpu -x ) > $D/pytest.log 2>&1; tail -5 $D/pytest.log
