#!/bin/bash
# Round 6: the augmented form of the fused pipeline (LPGP_RIDE_AUG=1: V^T as rows of the matrix being factored -- one update grid)
# against the riding substitution (two claimants), same box, interleaved.  Output: gpurun_out/r6_aug.txt
mkdir -p gpurun_out
export LPGP_BENCH_NO_MODES=1
{
for rep in 1 2; do
  for aug in 0 1; do
    echo "== c3 aug=$aug rep=$rep"
    LPGP_RIDE_AUG=$aug timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu ${EXTRA:-} 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'step_frac', d['roofline'].get('step_frac_of_peak'), 'parity', d.get('parity'))
"
  done
done
for aug in 0 1; do
  echo "== c2 aug=$aug"
  LPGP_GRAM_CAPACITY_HINT=9600 LPGP_RIDE_AUG=$aug timeout 600 python bench.py --workload poisson1d --steps 50 --warmup 5 --no-cpu 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'parity', d.get('parity'))
"
done
} 2>&1 | tee gpurun_out/r6_aug.txt
