#!/bin/bash
# A/B of the panel factorisation order (LPGP_DIAG_FIRST = 0 rows ride along, 1 diagonal block first, 2 auto) on c2 / c3 / c5
cd "$GRAFT_REPO_ROOT"
for wl in poisson1d poisson2d heat1d; do
  for f in 0 1 2; do
    for rep in 1 2; do
      LPGP_DIAG_FIRST=$f python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu 2>/dev/null | tail -n 1 | \
        python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl diag_first=$f', round(d['ms_per_step'],3), 'ms')"
    done
  done
done
