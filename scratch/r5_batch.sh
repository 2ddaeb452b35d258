#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r5_21; rm -rf $D; mkdir -p $D
timeout 1200 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_kernels.py tests/test_gpu_configs.py tests/test_gpu_golden.py tests/test_gpu_reference_cases.py tests/test_gpu_matern_iso.py tests/test_gpu_spawn.py -q -m gpu -k "not full_size" > $D/pytest.log 2>&1; echo "pytest rc=$?" >> $D/pytest.log; tail -4 $D/pytest.log
for b in 1 0; do echo "asm_batch=$b"; LPGP_ASM_BATCH=$b python scratch/small_sizes.py 2>&1 | head -5 | cut -c1-100; done
for b in 1 0; do LPGP_ASM_BATCH=$b LPGP_BENCH_NO_MODES=1 python bench.py --steps 20 --no-cpu 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 asm_batch=$b', round(d['ms_per_step'],3), {k:(round(v['achieved']),round(v['frac'],3),v['launches_per_step']) for k,v in d['roofline_assembly'].items()})"; done
