# per-panel timeline of the last potrf in a rocprofv3 kernel trace
import csv, glob, sys
d = sys.argv[1]
f = sorted(glob.glob(f'{d}/*/*kernel_trace.csv'), key=lambda p: -__import__("os").path.getmtime(p))[0]
rows = list(csv.DictReader(open(f)))
for r in rows: r['s']=int(r['Start_Timestamp']); r['e']=int(r['End_Timestamp']); r['b']=int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])
rows.sort(key=lambda r:r['s'])
pt=[r for r in rows if 'potrf_tile' in r['Kernel_Name']]
last=pt[-132:]
ws = last[0]['s']; we = last[-1]['e']
print("potrf window ms", (we-ws)/1e6)
sel=[r for r in rows if r['s']>=ws and r['s']<=we+2000000]
tiles=[r for r in sel if 'potrf_tile' in r['Kernel_Name']]
syrk=[r for r in sel if 'false, false, 1>' in r['Kernel_Name']]
print("panel  chain_us   tiles_us  nsyrk  big_syrk_us big_blocks  syrk_TF   gap_to_next")
for p in range(33):
    t4 = tiles[4*p:4*p+4]
    s0 = t4[0]['s']; s1 = tiles[4*p+4]['s'] if 4*p+4 < len(tiles) else we
    inwin=[r for r in syrk if s0 <= r['s'] < s1]
    big = max(inwin, key=lambda r:r['b']) if inwin else None
    n = 16896 - 512*(p+1)
    fl = n*(n+1.0)*512
    print(f"{p:3d} {(s1-s0)/1e3:9.1f} {sum(r['e']-r['s'] for r in t4)/1e3:9.1f} {len(inwin):5d} "
          + (f"{(big['e']-big['s'])/1e3:11.1f} {big['b']:9d} {fl/(big['e']-big['s'])/1e3:8.1f}" if big else ""))
