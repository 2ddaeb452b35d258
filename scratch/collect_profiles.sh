#!/bin/bash
# run on the GPU box: rocprofv3 stats + PMC passes of the bench command, then bench itself
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r01_stats gpurun_out/r01_pmc_fetch gpurun_out/r01_pmc_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01_stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/r01_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r01_pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu > gpurun_out/r01_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r01_pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu > gpurun_out/r01_pmc_write.log 2>&1
python3 bench.py --steps 5 --warmup 2 2>&1 | tail -1 > gpurun_out/bench_r01_c3.json
tail -c 300 gpurun_out/r01_stats.log
