#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r05_heat; rm -rf $D; mkdir -p $D
python3 scratch/small_trace.py heat 100 | tail -1
w=heat
rocprofv3 --kernel-trace --output-format csv -d $D/trace_$w -- python3 scratch/small_trace.py $w 20 > $D/trace_$w.log 2>&1
f=$(ls $D/trace_$w/*/*kernel_trace.csv | head -1)
python - "$f" > $D/trace_${w}_compact.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
for r in rows:
    n=r['Kernel_Name'].replace('void lpgp::','').replace('lpgp::','')
    n=n[:n.index('(')] if '(' in n else n
    print((int(r['Start_Timestamp'])-t0)//100, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))//100, r.get('Queue_Id','?'), int(r['Grid_Size_X'])//max(1,int(r['Workgroup_Size_X'])), n[:48])
PY
gzip -9f $D/trace_${w}_compact.txt; rm -rf $D/trace_$w
