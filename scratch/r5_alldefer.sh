#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r5_19; rm -rf $D; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_chain.py tests/test_gpu_configs.py -q -m gpu -k "not full_size" > $D/pytest.log 2>&1; echo "pytest rc=$?" >> $D/pytest.log; tail -4 $D/pytest.log
b() { local name=$1; shift
  env "$@" LPGP_BENCH_NO_MODES=1 timeout 600 python bench.py --steps $STEPS --warmup 3 --no-cpu $WL > $D/$name.json 2> $D/$name.err
  python -c "
import json
try:
    d=json.loads(open('$D/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3))
except Exception as e: print('$name FAILED', e)"
}
STEPS=30; WL=""
b c3 LPGP_X=1
b c3_again LPGP_X=1
WL="--workload poisson1d"; b c2 LPGP_X=1
WL="--workload heat1d"; STEPS=10; b c5 LPGP_X=1; b c5_gate65 LPGP_RIDE_GATE_PCT=65; b c5_gate85 LPGP_RIDE_GATE_PCT=85
STEPS=50
WL="--workload heat_reference"; b heatref LPGP_X=1
WL="--workload scattered2d"; STEPS=20; b scattered LPGP_X=1
python scratch/small_sizes.py 2>&1 | head -5 | cut -c1-100
