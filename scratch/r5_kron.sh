#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r5_22; rm -rf $D; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_kernels.py tests/test_gpu_configs.py -q -m gpu -k "not full_size" > $D/pytest.log 2>&1; echo "pytest rc=$?" >> $D/pytest.log; tail -4 $D/pytest.log
for b in 1 0 1 0; do LPGP_KRON_WIDE=$b LPGP_BENCH_NO_MODES=1 python bench.py --steps 20 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 kron_wide=$b', round(d['ms_per_step'],3), {k:(round(v['achieved']),round(v['frac'],3),v['launches_per_step']) for k,v in d['roofline_assembly'].items()})"; done
for b in 1 0; do LPGP_KRON_WIDE=$b LPGP_BENCH_NO_MODES=1 python bench.py --steps 3 --no-cpu --n-side 256 --m-side 128 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 kron_wide=$b', round(d['ms_per_step'],1), {k:(round(v['achieved']),round(v['frac'],3)) for k,v in d['roofline_assembly'].items()})"; done
