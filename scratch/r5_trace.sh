#!/bin/bash
# kernel trace of a bench run: r5_trace.sh <outdir> <bench args...>   (env vars of the variant exported by the caller)
set -u
cd "$GRAFT_REPO_ROOT"
D=$1; shift
mkdir -p $D
export LPGP_BENCH_NO_MODES=1 LPGP_BENCH_PROF_STEPS=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d $D/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu "$@" > $D/trace.log 2>&1
f=$(ls $D/trace/*/*kernel_trace.csv | head -1)
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
gzip -9f $D/trace_compact.txt; rm -rf $D/trace
tail -c 300 $D/trace.log | tr '\n' ' ' | cut -c1-200; echo
