#!/bin/bash
# round 6 profile collection: bench under rocprofv3 (c3, c2), the reference sequence under rocprofv3 --stats, bench lines
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash scratch/collect_profiles.sh r06 c3
bash scratch/collect_profiles.sh r06 c2 --workload poisson1d
D=gpurun_out/r06_refseq
rm -rf $D; mkdir -p $D
echo "rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 scratch/r6_refseq.py 5 c3" > $D/commands.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 scratch/r6_refseq.py 5 c3 > $D/stats.log 2>&1
find $D/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r06_refseq_c3_kernel_stats.csv
tail -2 $D/stats.log
python3 bench.py 2> gpurun_out/r06_bench_line_default.err | tail -1 > gpurun_out/r06_bench_line_default.json
python3 bench.py --workload poisson1d --steps 50 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06_bench_line_c2.json
python3 bench.py --workload heat1d --steps 10 --warmup 2 2>/dev/null | tail -1 > gpurun_out/r06_bench_line_c5.json
python3 bench.py --workload poisson1d_c1 --steps 50 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06_bench_line_poisson1d_c1.json
python3 bench.py --workload heat_reference --steps 50 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06_bench_line_heat_reference.json
python3 - <<'PY'
import json
for n in ("default", "c2", "c5", "poisson1d_c1", "heat_reference"):
    try:
        d = json.load(open(f"gpurun_out/r06_bench_line_{n}.json"))
        print(n, "ms", round(d["ms_per_step"], 3), "two_pipeline", round(d["two_pipeline_ms_per_step"], 3), "frac", round(d["roofline"]["frac"], 3),
              "refseq", None if "reference_sequence" not in d else round(d["reference_sequence"]["default_mode_ms"], 3), "parity", d.get("parity", {}).get("pass"))
    except Exception as e:
        print(n, "failed", e)
PY
