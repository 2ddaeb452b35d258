#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r5_02; rm -rf $D; mkdir -p $D
timeout 1200 python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_kernels.py tests/test_gpu_configs.py -q -m gpu -k "not full_size" > $D/pytest.log 2>&1; echo "pytest rc=$?" >> $D/pytest.log
tail -8 $D/pytest.log
( time python tests/golden/make_c4_golden.py --out $D/c4_posterior.npz ) > $D/c4_golden.log 2>&1; tail -4 $D/c4_golden.log
b() { # name, env...
  local name=$1; shift
  env "$@" LPGP_BENCH_NO_MODES=1 python bench.py --steps 20 --warmup 3 --no-cpu > $D/$name.json 2> $D/$name.err
  python - "$D/$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms_per_step", round(d["ms_per_step"],3), "phase", {k:round(v,2) for k,v in d["phase_ms"].items() if k!="note"}, "roof", round(d["roofline"]["frac"],3))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
b c3_eager LPGP_BENCH_EAGER=1
b c3_r1o3 LPGP_RIDE_STREAM=1 LPGP_RIDE_OCC3=1
b c3_r2o3 LPGP_RIDE_STREAM=2 LPGP_RIDE_OCC3=1
b c3_r1o3_res0 LPGP_RIDE_STREAM=1 LPGP_RIDE_OCC3=1 LPGP_RESERVE_CUS=0
b c3_r1o3_res0_n0 LPGP_RIDE_STREAM=1 LPGP_RIDE_OCC3=1 LPGP_RESERVE_CUS=0 LPGP_RESERVE_CUS_NARROW=0
b c3_r1o3_res8 LPGP_RIDE_STREAM=1 LPGP_RIDE_OCC3=1 LPGP_RESERVE_CUS=8
b c3_r1o3_g3f LPGP_RIDE_STREAM=1 LPGP_RIDE_OCC3=1 LPGP_GEMM3_FACT=1
b c3_r1o3_nb1024 LPGP_RIDE_STREAM=1 LPGP_RIDE_OCC3=1 LPGP_NB=1024
b c3_r1o3_chain0 LPGP_RIDE_STREAM=1 LPGP_RIDE_OCC3=1 LPGP_CHAIN_US_TILE=1 LPGP_CHAIN_US_FIXED=1
b c3_r1o3_chain600 LPGP_RIDE_STREAM=1 LPGP_RIDE_OCC3=1 LPGP_CHAIN_US_TILE=600
b c3_r1o3_again LPGP_RIDE_STREAM=1 LPGP_RIDE_OCC3=1
b c3_eager_again LPGP_BENCH_EAGER=1
export LPGP_RIDE_STREAM=1 LPGP_RIDE_OCC3=1 LPGP_BENCH_NO_MODES=1 LPGP_BENCH_PROF_STEPS=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d $D/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $D/trace.log 2>&1
f=$(ls $D/trace/*/*kernel_trace.csv | head -1); ls -la $f
python - "$f" > $D/trace_compact.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
for r in rows:
    n=r['Kernel_Name'].replace('void lpgp::','').replace('lpgp::','')
    n=n[:n.index('(')] if '(' in n else n
    print((int(r['Start_Timestamp'])-t0)//100, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))//100, r.get('Queue_Id','?'), int(r['Grid_Size_X'])//max(1,int(r['Workgroup_Size_X'])), n[:48])
PY
gzip -9 $D/trace_compact.txt; rm -rf $D/trace; ls -la $D | head -30
