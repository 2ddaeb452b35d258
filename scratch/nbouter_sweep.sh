#!/bin/bash
# re-sweep of the outer-panel schedule (far columns updated once per LPGP_NB_OUTER columns) on c3 with the round-2 chain
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do python bench.py --steps 10 --warmup 3 --no-cpu 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default', round(d['ms_per_step'],3))"; done
for nbo in 1024 2048; do for mt in 48 80 112; do
  LPGP_NB_OUTER=$nbo LPGP_NB_OUTER_MIN_TILES=$mt python bench.py --steps 10 --warmup 3 --no-cpu 2>/dev/null | tail -n 1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nb_outer=$nbo min_tiles=$mt', round(d['ms_per_step'],3))"
done; done
