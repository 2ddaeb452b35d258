#!/bin/bash
# round 6, final pass: the whole suite, then the profile collection and the bench lines (scratch/r6_collect.sh)
mkdir -p gpurun_out
( time python -m pytest tests -q -m gpu ) > gpurun_out/r06_gpu_suite_pytest.log 2>&1
tail -4 gpurun_out/r06_gpu_suite_pytest.log
bash scratch/r6_collect.sh
