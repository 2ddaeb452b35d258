import csv, glob, os, sys
d = sys.argv[1]; panels = [int(x) for x in sys.argv[2:]] or [20, 28]
f = sorted(glob.glob(f'{d}/*/*kernel_trace.csv'), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
for r in rows: r['s']=int(r['Start_Timestamp']); r['e']=int(r['End_Timestamp']); r['b']=int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])
rows.sort(key=lambda r:r['s'])
pt=[r for r in rows if 'potrf_tile' in r['Kernel_Name']]
last=pt[-132:]
def short(n):
    if 'potrf_tile' in n: return 'TILE'
    if 'gemm' in n: return n[n.index('gemm'):n.index('>')+1].replace('_f64_kernel','')
    return n[:20]
for p in panels:
    s0 = last[4*p]['s']; s1 = last[4*p+4]['s'] if 4*p+4 < len(last) else last[-1]['e'] + 300000
    sel=[r for r in rows if s0 <= r['s'] < s1]
    print("panel", p)
    for r in sel:
        print(f"  t={(r['s']-s0)/1e3:8.1f} dur {(r['e']-r['s'])/1e3:7.1f} q={r['Queue_Id']} blocks={r['b']:5d} {short(r['Kernel_Name'])}")
