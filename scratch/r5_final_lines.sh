#!/bin/bash
# after the profiles of the final sources are in profiles/: the default bench line (now with roofline.traffic) and the small sizes on a fresh box
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r05_final; rm -rf $D; mkdir -p $D
python3 scratch/small_sizes.py 2>&1 | head -5 > $D/small_sizes.txt; cat $D/small_sizes.txt
python3 bench.py > $D/bench_line_default.json 2> $D/bench_line_default.err; tail -c 600 $D/bench_line_default.json
python3 bench.py --workload poisson1d_c1 --steps 200 > $D/bench_line_poisson1d_c1.json 2>/dev/null
python3 scratch/small_sizes.py 2>&1 | head -5 > $D/small_sizes2.txt; cat $D/small_sizes2.txt
