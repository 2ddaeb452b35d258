#!/bin/bash
# A/B: remainder update (b) beside a chain-bound panel chain on the 64x64-tile kernel up to LPGP_B_SMALL_TILES 128-tiles
cd "$GRAFT_REPO_ROOT"
for wl in poisson1d poisson2d; do
  for f in 0 600 1200 2400 100000; do
    for rep in 1 2; do
      LPGP_B_SMALL_TILES=$f python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu 2>/dev/null | tail -n 1 | \
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl b_small_tiles=$f', round(d['ms_per_step'],3), 'ms')"
    done
  done
done
