#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export LPGP_BENCH_NO_MODES=1
run() { python3 bench.py "$@" --no-cpu 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); a=d['roofline_assembly']; print(round(d['ms_per_step'],3), 'asm', round(a['assemble']['frac'],3), round(a['assemble']['achieved']), 'grid', round(a.get('assemble_grid',{}).get('frac',0),3))"; }
for rep in 1 2; do
for v in 0 1000000000; do
  export LPGP_GRID_MIN_POINTS=$v
  echo "min_points=$v c3: $(run --steps 10 --warmup 3) | n192: $(run --n-side 192 --m-side 96 --steps 3 --warmup 1)"
done
done
