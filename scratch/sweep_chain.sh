#!/bin/bash
# sweep of the chain-bound / update-bound crossover estimates after the tile solves got their refinement step
for c in 115 150 190; do for s in 115 150 190; do
  echo -n "chain_us_tile=$c solve_chain_us_tile=$s : "
  LPGP_CHAIN_US_TILE=$c LPGP_SOLVE_CHAIN_US_TILE=$s python bench.py --steps 5 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2))"
done; done
for nb in 512 768 1024; do echo -n "nb_solve=$nb : "; LPGP_NB_SOLVE=$nb python bench.py --steps 5 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2))"; done
echo -n "c2: "; python bench.py --workload poisson1d --steps 10 --warmup 3 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2))"
