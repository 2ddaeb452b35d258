#!/bin/bash
# does another shape of the tile block an XCD works on (band x 64/band tiles) cut the operand re-fetches
# of the rank-512 trailing update?  time + HIP-event rate of the SYRK per band height, then FETCH_SIZE / WRITE_SIZE per launch
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for b in 4 8 16 32; do
  echo -n "band=$b : "
  LPGP_GEMM_BAND=$b python3 bench.py --steps 5 --warmup 2 --no-cpu 2>/dev/null | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms', round(d['ms_per_step'],2), 'syrk TF/s', round(d['roofline']['achieved'],2), 'solve gemm TF/s', round(d['kernels']['gemm']['tflops'],2))"
done
for b in 4 8 16; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/band_$b_$c
    LPGP_GEMM_BAND=$b rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/band_${b}_$c -- python3 bench.py --steps 1 --warmup 0 --no-cpu > /dev/null 2>&1
  done
  python3 - <<PY
import csv, glob
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/band_${b}_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "gemm_f64_kernel<false, false, 1>" in r["Kernel_Name"]]
    tot[c] = (sum(float(r["Counter_Value"]) for r in rows), len(rows))
print("band=${b}: SYRK launches", tot["FETCH_SIZE"][1], "HBM bytes per launch (2*FETCH+WRITE)*1024 = %.3f GB" % ((2 * tot["FETCH_SIZE"][0] + tot["WRITE_SIZE"][0]) * 1024 / tot["FETCH_SIZE"][1] / 1e9))
PY
  rm -rf gpurun_out/band_${b}_FETCH_SIZE gpurun_out/band_${b}_WRITE_SIZE
done
