#!/bin/bash
# condition + predict on one GPU against the problem size (2-D Poisson, n x n collocation, (n/2)^2 prediction points)
cd "$GRAFT_REPO_ROOT"
for n in 32 48 64 96 128 160 192 224 256; do
  python bench.py --n-side $n --m-side $((n/2)) --steps 5 --warmup 2 --no-cpu 2>/dev/null | tail -n 1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('n_side=$n N_tot=%d M=%d  %.3f ms  %.1f TFLOP/s  (%.1f %% of 78.6)' % (c['n_total'], c['m_predict'], d['ms_per_step'], d['value']/1e3, d['value']/786.0))"
done
