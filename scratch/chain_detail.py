# kernel-by-kernel timeline of the panel chain in the chain-bound phase of the LAST potrf of a trace
import csv, glob, os, sys
d = sys.argv[1]
lo, hi = int(sys.argv[2]), int(sys.argv[3])          # tile range to print
f = sorted(glob.glob(f'{d}/**/*kernel_trace.csv', recursive=True), key=lambda p: -os.path.getmtime(p))[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    r['b'] = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])
rows.sort(key=lambda r: r['s'])
pt = [r for r in rows if 'potrf_tile' in r['Kernel_Name']]
ntiles = int(sys.argv[4]) if len(sys.argv) > 4 else 132
last = pt[-ntiles:]
t0 = last[lo]['s']; t1 = last[hi]['s'] if hi < ntiles else last[-1]['e'] + 100000
def short(n):
    n = n.replace('void lpgp::', '').replace('(lpgp::GemmArgs)', '')
    return n[:44]
prev_e = None
for r in rows:
    if r['s'] < t0 or r['s'] >= t1: continue
    print(f"{(r['s']-t0)/1e3:9.1f} us  dur {(r['e']-r['s'])/1e3:7.1f}  wgs {r['b']:6d}  q {r.get('Queue_Id','?'):>3}  {short(r['Kernel_Name'])}")
print("window us", (t1 - t0) / 1e3, "per tile", (t1 - t0) / 1e3 / (hi - lo))
