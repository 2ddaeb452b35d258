#!/bin/bash
# round 6: the profile collections of c4 (one GPU) and c5 on the final sources, and c4's bench line
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash scratch/collect_profiles.sh r06 c5 --workload heat1d
bash scratch/collect_profiles.sh r06 c4 --n-side 256 --m-side 128
python3 bench.py --n-side 256 --m-side 128 --steps 3 --warmup 1 --no-cpu 2>/dev/null | tail -1 > gpurun_out/r06_bench_line_c4.json
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench_line_c4.json"))
print("c4 ms", round(d["ms_per_step"], 1), "two_pipeline", d.get("two_pipeline_ms_per_step"), "frac", round(d["roofline"]["frac"], 3), "step_frac", d["roofline"].get("step_frac_of_peak"))
PY
