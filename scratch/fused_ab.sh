#!/bin/bash
# A/B of the fused panel step of the forward substitution
for w in "" "--workload poisson1d" "--workload heat1d"; do for f in 0 1; do
  echo -n "fused_solve=$f $w : "
  LPGP_FUSED_SOLVE=$f python bench.py --steps 6 --warmup 2 --no-cpu $w 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), {k:(round(v['ms_per_step'],2),v['launches_per_step']) for k,v in d['kernels'].items() if k in ('panel_fused','trsm_gemm','gemm_small')})"
done; done
