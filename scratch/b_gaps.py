# consecutive remainder updates (b) of the LAST potrf in a trace: duration and the gap before each
import csv, glob, os, sys
d = sys.argv[1]
f = sorted(glob.glob(f'{d}/**/*kernel_trace.csv', recursive=True), key=lambda p: -os.path.getmtime(p))[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    r['b'] = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])
rows.sort(key=lambda r: r['s'])
pt = [r for r in rows if 'potrf_tile' in r['Kernel_Name']]
last = pt[-132:]
t0 = last[0]['s']; t1 = last[-1]['e']
bs = [r for r in rows if t0 <= r['s'] <= t1 and 'gemm_f64_kernel<false, false, 1>' in r['Kernel_Name']]
prev = None
tot_gap = 0
for r in bs:
    gap = (r['s'] - prev['e']) / 1e3 if prev else 0.0
    tot_gap += max(gap, 0)
    print(f"start {(r['s']-t0)/1e3:9.1f}  dur {(r['e']-r['s'])/1e3:8.1f}  wgs {r['b']:6d}  gap_before {gap:8.1f} us  q {r.get('Queue_Id')}")
    prev = r
print("potrf window ms", (t1 - t0) / 1e6, " sum of (b) durations ms", sum(r['e'] - r['s'] for r in bs) / 1e6, " sum of gaps ms", tot_gap / 1e3)
