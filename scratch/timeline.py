import csv, glob, sys
d = sys.argv[1]; skip = int(sys.argv[2]) if len(sys.argv) > 2 else 10; cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 32
f = glob.glob(f'{d}/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
for r in rows: r['s']=int(r['Start_Timestamp']); r['e']=int(r['End_Timestamp'])
rows.sort(key=lambda r:r['s'])
pt=[r for r in rows if 'potrf_tile' in r['Kernel_Name']]
last=pt[-132:]
ws = last[4]['s']; we = last[-1]['e']
print("big potrf window ms", (we-ws)/1e6)
sel=[r for r in rows if r['s']>=ws and r['s']<=we]
def short(n):
    if 'potrf_tile' in n: return 'TILE'
    if 'gemm' in n: return 'GEMM'+n[n.index('<'):n.index('>')+1]
    return n[:20]
for r in sel[skip:skip+cnt]:
    print(f"{(r['s']-ws)/1e3:10.1f} us  dur {(r['e']-r['s'])/1e3:8.1f} us  q={r['Queue_Id']} blocks={int(r['Grid_Size_X'])//256:>6} {short(r['Kernel_Name'])}")
