#!/bin/bash
# same-box sweeps of the fused pipeline's policy knobs at c3 (ms per step, 10 steps, 3 warm-up)
set -u
cd "$GRAFT_REPO_ROOT"
export LPGP_BENCH_NO_MODES=1
run() { python3 bench.py "$@" --no-cpu 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
echo "default $(run --steps 10 --warmup 3) $(run --steps 10 --warmup 3)"
for g in 55 60 63 68 72; do echo "gate=$g $(LPGP_RIDE_GATE_PCT=$g run --steps 10 --warmup 3) $(LPGP_RIDE_GATE_PCT=$g run --steps 10 --warmup 3)"; done
for r in 16 24 40 48; do echo "resident=$r $(LPGP_CHAIN_RESIDENT=$r run --steps 10 --warmup 3) $(LPGP_CHAIN_RESIDENT=$r run --steps 10 --warmup 3)"; done
for o in 1536 3072 4096; do echo "outer_rows=$o $(LPGP_RIDE_OUTER_ROWS=$o run --steps 10 --warmup 3) $(LPGP_RIDE_OUTER_ROWS=$o run --steps 10 --warmup 3)"; done
echo "default $(run --steps 10 --warmup 3) $(run --steps 10 --warmup 3)"
