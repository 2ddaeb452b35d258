#!/bin/bash
OUT=$1; mkdir -p $OUT
for i in 1 2; do
for v in "X=0" "GPU_MAX_HW_QUEUES=8"; do
  for args in "--workload poisson2d --steps 20" "--n-side 256 --m-side 128 --steps 3" "--workload heat1d --steps 6" "--workload poisson1d --steps 30"; do
    env $v python3 bench.py --warmup 2 --no-cpu $args 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$v] $args', round(d['ms_per_step'],3), 'cond', round(d['phase_ms']['condition'],2), 'pred', round(d['phase_ms']['predict'],2))" | tee -a $OUT/ab.txt
  done
done
done
