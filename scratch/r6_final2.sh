#!/bin/bash
# after r6_final.sh (summaries of c3 / c2 copied into profiles/): the new golden test, the default bench line with `traffic`, c5 / c4 collections
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_golden.py -q -m gpu 2>&1 | tail -3
python3 bench.py 2> gpurun_out/r06_bench_line_default.err | tail -1 > gpurun_out/r06_bench_line_default.json
python3 bench.py --workload poisson1d --steps 50 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06_bench_line_c2.json
bash scratch/r6_collect_c4c5.sh
python3 bench.py --workload heat1d --steps 10 --warmup 2 2>/dev/null | tail -1 > gpurun_out/r06_bench_line_c5.json
python3 - <<'PY'
import json
for n in ("default", "c2", "c5", "c4"):
    d = json.load(open(f"gpurun_out/r06_bench_line_{n}.json"))
    print(n, "ms", round(d["ms_per_step"], 3), "frac", round(d["roofline"]["frac"], 3), "traffic", d["roofline"].get("traffic"), "sha", d["config"]["csrc_sha16"])
PY
