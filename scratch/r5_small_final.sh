#!/bin/bash
# the small-problem evidence after the grid threshold (Python-only change: kernel sources / csrc_sha16 unchanged), then the full GPU suite
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r05_small_final; rm -rf $D; mkdir -p $D
python3 scratch/small_sizes.py 2>&1 | head -5 > $D/small_sizes.txt; cat $D/small_sizes.txt
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
python3 bench.py --workload heat_reference --steps 50 > $D/bench_line_heat_reference.json 2>/dev/null
python3 bench.py --workload poisson1d_c1 --steps 50 > $D/bench_line_poisson1d_c1.json 2>/dev/null
python3 -c "
import json
for f in ('heat_reference','poisson1d_c1'):
    d=json.loads(open('$D/bench_line_'+f+'.json').read().strip().splitlines()[-1]); print(f, round(d['ms_per_step'],3), round(d['modes']['eager_default']['ms_per_step'],3), d['parity']['pass'])"
( time timeout 1700 python -m pytest tests -q -m gpu ) > $D/gpu_suite_pytest.log 2>&1; tail -5 $D/gpu_suite_pytest.log
