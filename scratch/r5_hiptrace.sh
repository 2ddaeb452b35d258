#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/r05_hiptrace; mkdir -p $D
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in c1 p32; do
  rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d $D/t_$w -- python3 scratch/small_trace.py $w 50 > $D/t_$w.log 2>&1
  tail -1 $D/t_$w.log
  f=$(ls $D/t_$w/*/*hip_api_trace.csv | head -1)
  python3 - "$f" 55 > $D/hip_api_$w.txt <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
nsteps=int(sys.argv[2])
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# steady state: the last 40 % of the calls
rows=rows[int(len(rows)*0.6):]
agg=collections.defaultdict(lambda:[0,0])
for r in rows:
    a=agg[r['Function']]; a[0]+=1; a[1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
span=(int(rows[-1]['End_Timestamp'])-int(rows[0]['Start_Timestamp']))/1e3
tot=sum(a[1] for a in agg.values())/1e3
print(f"window {span:.0f} us, inside HIP API calls {tot:.0f} us ({100*tot/span:.0f} %)")
for k,(n,t) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:25]:
    print(f"{k:40s} n={n:6d} total {t/1e3:9.1f} us avg {t/n/1e3:7.2f} us  share of window {100*t/1e3/span:5.1f} %")
PY
  cat $D/hip_api_$w.txt
  rm -rf $D/t_$w
done
