# kernels of the LAST potrf in a rocprofv3 kernel trace: everything longer than a threshold, with queue
import csv, glob, os, sys
d = sys.argv[1]; thr = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
ntiles = int(sys.argv[3]) if len(sys.argv) > 3 else 132
f = sorted(glob.glob(f'{d}/**/*kernel_trace.csv', recursive=True), key=lambda p: -os.path.getmtime(p))[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    r['b'] = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])
rows.sort(key=lambda r: r['s'])
pt = [r for r in rows if 'potrf_tile' in r['Kernel_Name']]
last = pt[-ntiles:]
t0 = last[0]['s']; t1 = last[-1]['e']
print("potrf window ms", (t1 - t0) / 1e6)
def short(n):
    return n.replace('void lpgp::', '').replace('(lpgp::GemmArgs)', '')[:40]
k = 0
for r in rows:
    if r['s'] < t0 - 2e6 or r['s'] > t1: continue
    if 'potrf_tile' in r['Kernel_Name']:
        k += 1
        continue
    if (r['e'] - r['s']) / 1e3 >= thr:
        print(f"{(r['s']-t0)/1e3:9.1f} -> {(r['e']-t0)/1e3:9.1f} us  dur {(r['e']-r['s'])/1e3:8.1f}  wgs {r['b']:6d}  q {r.get('Queue_Id','?'):>3}  tiles_done {k:3d}  {short(r['Kernel_Name'])}")
