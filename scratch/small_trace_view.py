"""One step of a compact small-problem trace: python scratch/small_trace_view.py <trace_compact.txt.gz>"""
import gzip, sys
rows=[]
for ln in gzip.open(sys.argv[1],"rt"):
    p=ln.split(None,4); rows.append((int(p[0])/10,int(p[1])/10,p[2],int(p[3]),p[4].strip()))
ends=[i for i,r in enumerate(rows) if r[4].startswith("col_reduce")]
k=len(ends)-3
lo=ends[k-1]+1; hi=ends[k]
t0=rows[lo][0]
print("launches", hi-lo+1, "span %.1f us"%(rows[hi][0]+rows[hi][1]-t0), "period %.1f"%(rows[ends[k]][0]-rows[ends[k-1]][0]))
last=t0
for r in rows[lo:hi+1]:
    print(f"  {r[0]-t0:8.1f} {r[1]:7.1f} gap {r[0]-last:6.1f} q{r[2]} {r[3]:5d} {r[4]}")
    last=max(last,r[0]+r[1])
