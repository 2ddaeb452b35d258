#!/bin/bash
# round 6, first full pass on the GPU box: the whole suite, then the bench lines and the reference sequence
mkdir -p gpurun_out
( time python -m pytest tests -x -q -m gpu ) > gpurun_out/r6_suite.log 2>&1
tail -5 gpurun_out/r6_suite.log
python bench.py --steps 20 --warmup 3 2> gpurun_out/r6_bench_c3.err | tail -1 > gpurun_out/r6_bench_c3.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6_bench_c3.json"))
print("c3 ms", d["ms_per_step"], "mode", d["mode"], "two_pipeline", d["two_pipeline_ms_per_step"], "frac", d["roofline"]["frac"], "parity", d["parity"].get("pass"))
print("refseq", {k: v for k, v in d["reference_sequence"].items() if k.endswith("_ms")})
print("modes", d["modes"]["eager_default"]["ms_per_step"])
PY
python scratch/r6_refseq.py 5 c3
LPGP_TRSV_RESIDENT=0 python scratch/r6_refseq.py 5 c3
python scratch/r6_refseq.py 10 c2
