#!/bin/bash
mkdir -p gpurun_out
export LPGP_BENCH_NO_MODES=1
{
for rep in 1 2; do
for v in 55 60 65 70 75; do
  echo -n "rep=$rep gate_pct=$v poisson2d: "
  LPGP_RIDE_GATE_PCT=$v timeout 600 python bench.py --steps 30 --warmup 4 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_step %.3f frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"
done
done
for v in 55 65 75; do
  echo -n "gate_pct=$v poisson1d: "
  LPGP_RIDE_GATE_PCT=$v timeout 600 python bench.py --workload poisson1d --steps 30 --warmup 4 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms_per_step %.3f' % (d['ms_per_step']))"
done
} 2>&1 | tee gpurun_out/r6_gate.txt
