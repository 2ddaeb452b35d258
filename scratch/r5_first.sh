#!/bin/bash
# round 5, first contact of the fused factor-and-predict pipeline: tests, then bench A/B over the ride stream
set -u
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r5_01; rm -rf $D; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py -x -q -m gpu > $D/pytest.log 2>&1; echo "pytest rc=$?" >> $D/pytest.log
tail -5 $D/pytest.log
b() { # name, env..., -- args
  local name=$1; shift
  env "$@" python bench.py --steps 20 --warmup 3 --no-cpu > $D/$name.json 2> $D/$name.err
  python - "$D/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms_per_step", round(d["ms_per_step"],3), "phase", {k:round(v,2) for k,v in d["phase_ms"].items() if k!="note"}, "roof", round(d["roofline"]["frac"],3), "modes", {k:round(v["ms_per_step"],2) for k,v in d.get("modes",{}).items()}, "seq", {k:round(v,2) for k,v in d.get("reference_sequence",{}).items() if k.endswith("_ms")}, "e2e", d.get("e2e_with_h2d_ms"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
b c3_default LPGP_X=0
b c3_eager LPGP_BENCH_EAGER=1 LPGP_BENCH_NO_MODES=1
b c3_ride1 LPGP_RIDE_STREAM=1 LPGP_BENCH_NO_MODES=1
b c3_ride2 LPGP_RIDE_STREAM=2 LPGP_BENCH_NO_MODES=1
b c3_ride3 LPGP_RIDE_STREAM=3 LPGP_BENCH_NO_MODES=1
b c3_occ3 LPGP_RIDE_OCC3=1 LPGP_BENCH_NO_MODES=1
b c3_ride1_occ3 LPGP_RIDE_STREAM=1 LPGP_RIDE_OCC3=1 LPGP_BENCH_NO_MODES=1
for w in poisson1d heat1d; do
  for v in "LPGP_X=0" "LPGP_BENCH_EAGER=1" "LPGP_RIDE_STREAM=1" "LPGP_RIDE_STREAM=2"; do
    name=${w}_$(echo $v | tr '=' '_')
    env $v LPGP_BENCH_NO_MODES=1 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu > $D/$name.json 2> $D/$name.err
    python -c "
import json,sys
try:
    d=json.loads(open('$D/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['phase_ms'].items() if k!='note'})
except Exception as e: print('$name FAILED', e)
"
  done
done
