#!/bin/bash
# round 5: the committed evidence -- profiles of the four BASELINE configurations, bench lines, small sizes, the full GPU suite
set -u
cd "$GRAFT_REPO_ROOT"
bash scratch/collect_profiles.sh r05 c3
bash scratch/collect_profiles.sh r05 c2 --workload poisson1d
bash scratch/collect_profiles.sh r05 c5 --workload heat1d
bash scratch/collect_profiles.sh r05 c4 --n-side 256 --m-side 128
D=gpurun_out/r05_lines; rm -rf $D; mkdir -p $D
python3 bench.py > $D/bench_line_default.json 2> $D/bench_line_default.err
python3 bench.py --workload poisson1d_c1 --steps 50 > $D/bench_line_poisson1d_c1.json 2>/dev/null
python3 bench.py --workload heat_reference --steps 50 > $D/bench_line_heat_reference.json 2>/dev/null
python3 bench.py --workload scattered2d --steps 20 > $D/bench_line_scattered2d.json 2>/dev/null
python3 bench.py --workload poisson1d --steps 50 > $D/bench_line_c2.json 2>/dev/null
python3 bench.py --workload heat1d --steps 10 --no-cpu > $D/bench_line_c5.json 2>/dev/null
python3 scratch/small_sizes.py 2>&1 | head -5 > $D/small_sizes.txt
# kernel traces of one step of the small problems (compact: start/100 ns, duration/100 ns, queue, workgroups, kernel)
for w in c1 p32 heat; do
  rocprofv3 --kernel-trace --output-format csv -d $D/trace_$w -- python3 scratch/small_trace.py $w 20 > $D/trace_$w.log 2>&1
  f=$(ls $D/trace_$w/*/*kernel_trace.csv | head -1)
  python3 - "$f" > $D/small_trace_${w}.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
for r in rows:
    n=r['Kernel_Name'].replace('void lpgp::','').replace('lpgp::','')
    n=n[:n.index('(')] if '(' in n else n
    print((int(r['Start_Timestamp'])-t0)//100, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))//100, r.get('Queue_Id','?'), int(r['Grid_Size_X'])//max(1,int(r['Workgroup_Size_X'])), n[:48])
PY
  gzip -9f $D/small_trace_${w}.txt; rm -rf $D/trace_$w $D/trace_$w.log
done
for v in 96 0; do echo "ride_vchain_max_wgs=$v: $(LPGP_RIDE_VCHAIN=$v python3 scratch/small_trace.py c1 300 | tail -1); $(LPGP_RIDE_VCHAIN=$v python3 scratch/small_trace.py p32 300 | tail -1); $(LPGP_RIDE_VCHAIN=$v python3 scratch/small_trace.py heat 100 | tail -1)" >> $D/vchain_ab.txt; done
cat $D/vchain_ab.txt
for band in 4 8 16; do
  LPGP_GEMM_BAND=$band LPGP_BENCH_NO_MODES=1 python3 bench.py --steps 20 --no-cpu 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('band $band: ms_per_step', round(d['ms_per_step'],3), 'syrk frac', round(d['roofline']['frac'],3))" >> $D/band_ab.txt
  LPGP_GEMM_BAND=$band LPGP_BENCH_EAGER=1 LPGP_BENCH_NO_MODES=1 python3 bench.py --steps 20 --no-cpu 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('band $band (two pipelines): ms_per_step', round(d['ms_per_step'],3), 'syrk frac', round(d['roofline']['frac'],3))" >> $D/band_ab.txt
done
cat $D/band_ab.txt
( time timeout 1700 python -m pytest tests -q -m gpu ) > $D/gpu_suite_pytest.log 2>&1; tail -5 $D/gpu_suite_pytest.log
