# Timeline of ONE c3 step from a rocprofv3 --kernel-trace CSV: how long each phase takes, how busy the update queue is
# inside it, and what the chain kernels beside it cost (avg vs min durations).
#   usage: step_util.py <dir with *kernel_trace.csv> [step index from the end, default 2]
import csv, glob, os, sys, collections
d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = sorted(glob.glob(f'{d}/**/*kernel_trace.csv', recursive=True), key=lambda p: -os.path.getmtime(p))[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp']); r['n'] = r['Kernel_Name']
    r['b'] = int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))
    r['q'] = r.get('Queue_Id', '?')
rows.sort(key=lambda r: r['s'])
# a step starts with the first kron2 launch (PDE block assembly is late in the step) -> use potrf_tile count: 132 per step
pt = [i for i, r in enumerate(rows) if 'potrf_tile' in r['n']]
import os as _os
NP = int(_os.environ.get('NPOTRF', '132'))
nstep = len(pt) // NP
print("steps in trace:", nstep)
k = nstep - back
first, last = pt[NP * k], pt[NP * (k + 1)] if NP * (k + 1) < len(pt) else len(rows) - 1
t0, t1 = rows[first]['s'], rows[last]['s']
sel = [r for r in rows if t0 <= r['s'] < t1]
print(f"step window {(t1 - t0) / 1e6:.3f} ms, {len(sel)} launches")
def short(n):
    n = n.replace('lpgp::', '').replace('void ', '')
    return n[:n.index('(')] if '(' in n else n
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = None, None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    if cs is not None: tot += ce - cs
    return tot
byq = collections.defaultdict(list)
for r in sel: byq[r['q']].append(r)
for q, rs in sorted(byq.items()):
    print(f"queue {q}: {len(rs)} launches, busy {union([(r['s'], r['e']) for r in rs]) / 1e6:.3f} ms, sum {sum(r['e'] - r['s'] for r in rs) / 1e6:.3f} ms")
print(f"any queue busy: {union([(r['s'], r['e']) for r in sel]) / 1e6:.3f} ms")
agg = collections.defaultdict(list)
for r in sel: agg[short(r['n'])].append(r)
print(f"{'kernel':70s} {'n':>5s} {'sum ms':>8s} {'avg us':>8s} {'min us':>8s} {'max us':>8s} {'blocks':>8s}")
for n, rs in sorted(agg.items(), key=lambda kv: -sum(r['e'] - r['s'] for r in kv[1])):
    du = [(r['e'] - r['s']) / 1e3 for r in rs]
    print(f"{n[:70]:70s} {len(rs):5d} {sum(du) / 1e3:8.3f} {sum(du) / len(du):8.1f} {min(du):8.1f} {max(du):8.1f} {sum(r['b'] for r in rs) // len(rs):8d}")
# the big updates: busy union and gaps
big = [r for r in sel if ('gemm_f64_kernel<false, false, 1>' in r['n'] or 'gemm3_f64_kernel' in r['n'] or 'gemm_f64_kernel<false, true, 0>' in r['n']) and r['b'] >= 256]
bu = union([(r['s'], r['e']) for r in big])
print(f"big update launches: {len(big)}, busy {bu / 1e6:.3f} ms of {(t1 - t0) / 1e6:.3f}")
if '-v' in sys.argv:
    for r in sel:
        print(f"{(r['s'] - t0) / 1e3:10.1f} {(r['e'] - r['s']) / 1e3:8.1f} q{r['q']} {r['b']:6d} {short(r['n'])[:60]}")

upd = sorted([r for r in sel if r['b'] >= 200 and ('gemm_f64_kernel' in r['n'] or 'gemm3_f64' in r['n']) and r['q'] != sel[0]['q']], key=lambda r: r['s'])
pe = None; tot_gap = 0
for r in upd:
    if pe is not None and r['s'] > pe: tot_gap += r['s'] - pe
    pe = max(pe or 0, r['e'])
print(f"update queues: {len(upd)} launches, gaps between consecutive ones sum to {tot_gap / 1e6:.3f} ms")
