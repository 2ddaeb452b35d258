#!/bin/bash
# A/B: remainder update at ONE workgroup per CU (LPGP_SOLO_RATIO x chain estimate) on c2 / c3 / c5
cd "$GRAFT_REPO_ROOT"
for wl in ${WLS:-poisson1d poisson2d}; do
  for f in ${RATIOS:-0 0.5 1 1.5 2.5 100}; do
    for rep in 1 2; do
      LPGP_SOLO_RATIO=$f python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu 2>/dev/null | tail -n 1 | \
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl solo_ratio=$f', round(d['ms_per_step'],3), 'ms')"
    done
  done
done
