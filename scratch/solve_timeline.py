import csv, glob, os, sys
d = sys.argv[1]
f = sorted(glob.glob(f'{d}/*/*kernel_trace.csv'), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
for r in rows: r['s']=int(r['Start_Timestamp']); r['e']=int(r['End_Timestamp']); r['b']=int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])
rows.sort(key=lambda r:r['s'])
# last predict: from the last cross assemble kernels to the final col_reduce2
cr=[i for i,r in enumerate(rows) if 'col_reduce2' in r['Kernel_Name']]
end=cr[-1]; start=cr[-2]+1
sel=rows[start:end+1]
t0=sel[0]['s']
print("predict window ms", (sel[-1]['e']-t0)/1e6, "kernels", len(sel))
def short(n):
    if 'gemm' in n: return n[n.index('gemm'):n.index('>')+1].replace('_f64_kernel','')
    return n[:24]
big=[r for r in sel if 'gemm<false, true, 0>' in short(r['Kernel_Name']) and r['b']>2000]
print("big updates:", len(big))
tot=0
for r in big[:40]:
    fl = r['b']*2*128*128*512  # approx (blocks incl. padding of the super-tile grid)
    print(f"t={(r['s']-t0)/1e3:9.1f} dur {(r['e']-r['s'])/1e3:8.1f} q={r['Queue_Id']} blocks={r['b']}")
q1=[r for r in sel if r['Queue_Id']==sel[0]['Queue_Id']]
busy=sum(r['e']-r['s'] for r in q1)
print("panel-stream busy ms", busy/1e6)
# last 3 panels detail
for r in sel[-45:]:
    print(f"  t={(r['s']-t0)/1e3:9.1f} dur {(r['e']-r['s'])/1e3:7.1f} q={r['Queue_Id']} blocks={r['b']:5d} {short(r['Kernel_Name'])}")
