# Analysis of a compact kernel trace (scratch/r5_second.sh: "start/100ns dur/100ns queue wgs name") of the fused pipeline.
#   python scratch/fused_trace.py <trace_compact.txt.gz> [step index] [-v from_us to_us]
import gzip, sys, collections
rows = []
for ln in gzip.open(sys.argv[1], 'rt'):
    p = ln.split(None, 4)
    rows.append(dict(s=int(p[0]) / 10.0, d=int(p[1]) / 10.0, q=p[2], b=int(p[3]), n=p[4].strip()))
for r in rows: r['e'] = r['s'] + r['d']
# steps end with the column reduction of the prediction (col_reduce2_kernel / col_reduce_kernel): step k = the launches after
# the end of step k - 1 up to and including that kernel
ends = [i for i, r in enumerate(rows) if r['n'].startswith('col_reduce')]
nstep = len(ends)
k = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].lstrip('-').isdigit() else 2
i_lo = ends[k - 1] + 1 if k > 0 else 0
t0 = rows[i_lo]['s']
t1 = rows[ends[k]]['s'] + rows[ends[k]]['d'] + 0.05
sel = [r for r in rows if t0 <= r['s'] < t1]
tend = max(r['e'] for r in sel)
print(f"steps in trace {nstep}; step {k}: {len(sel)} launches, {(tend - t0) / 1e3:.3f} ms")
def union(iv):
    iv = sorted(iv); tot = 0; cs = ce = None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    if cs is not None: tot += ce - cs
    return tot
byq = collections.defaultdict(list)
for r in sel: byq[r['q']].append(r)
for q, rs in sorted(byq.items()):
    print(f"  queue {q}: {len(rs):5d} launches, busy {union([(r['s'], r['e']) for r in rs]) / 1e3:7.3f} ms, first {(min(r['s'] for r in rs) - t0) / 1e3:7.3f} last end {(max(r['e'] for r in rs) - t0) / 1e3:7.3f}")
agg = collections.defaultdict(list)
for r in sel: agg[(r['q'], r['n'])].append(r)
print(f"{'q':>2s} {'kernel':50s} {'n':>5s} {'sum ms':>8s} {'avg us':>8s} {'min':>7s} {'max':>7s} {'wgs':>6s}")
for (q, n), rs in sorted(agg.items(), key=lambda kv: -sum(r['d'] for r in kv[1])):
    du = [r['d'] for r in rs]
    print(f"{q:>2s} {n[:50]:50s} {len(rs):5d} {sum(du) / 1e3:8.3f} {sum(du) / len(du):8.1f} {min(du):7.1f} {max(du):7.1f} {sum(r['b'] for r in rs) // len(rs):6d}")
pts = [r for r in sel if 'potrf_tile' in r['n']]
print("potrf tile start times by panel (ms): " + " ".join(f"{(pts[i]['s'] - t0) / 1e3:.2f}" for i in range(0, len(pts), 4)))
print(f"last potrf ends {(pts[-1]['e'] - t0) / 1e3:.3f} ms; step ends {(tend - t0) / 1e3:.3f} ms")
if '-v' in sys.argv:
    i = sys.argv.index('-v'); a, b = float(sys.argv[i + 1]), float(sys.argv[i + 2])
    for r in sel:
        if a <= r['s'] - t0 < b:
            print(f"{r['s'] - t0:10.1f} {r['d']:8.1f} q{r['q']} {r['b']:6d} {r['n'][:60]}")
