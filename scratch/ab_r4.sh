#!/bin/bash
# A/B on one box: usage ab_r4.sh <outdir> <reps> "<ENV=.. ENV=..>" "<ENV..>" ...   -- c3 and c2 per variant, interleaved
OUT=$1; REPS=$2; shift 2
mkdir -p $OUT
for i in $(seq $REPS); do
  idx=0
  for v in "$@"; do
    for wl in poisson2d poisson1d; do
      env $v python3 bench.py --steps 30 --warmup 3 --no-cpu --workload $wl 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$v] $wl', round(d['ms_per_step'],3), 'cond', round(d['phase_ms']['condition'],2), 'pred', round(d['phase_ms']['predict'],2), 'potrf_us', round(1e3*d['kernels']['potrf_tile']['ms_per_step']/max(1,d['kernels']['potrf_tile']['launches_per_step']),1))" | tee -a $OUT/ab.txt
    done
  done
done
