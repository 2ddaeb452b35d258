# Whole-step view of the LAST bench step in a rocprofv3 kernel trace: per-kernel totals, time with nothing running,
# tile-to-tile period of the panel chain, and (optionally) the kernel-by-kernel list of a window.
#   python scratch/step_timeline.py <trace dir> <step index (0-based, counts tensor-grid assembly launches; warmup + steps - 1
#   = the last timed step, later ones belong to bench.py's per-kernel event pass)> <tile factorisations per step> [list_from_us list_to_us]
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]; step_idx = int(sys.argv[2])
f = sorted(glob.glob(f'{d}/**/*kernel_trace.csv', recursive=True), key=lambda p: -os.path.getmtime(p))[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    r['b'] = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])
rows.sort(key=lambda r: r['s'])
def short(n):
    return n.replace('void lpgp::', '').replace('(lpgp::GemmArgs)', '').replace('lpgp::', '')[:60]
pt = [r for r in rows if 'potrf_tile' in r['Kernel_Name']]
per_step = int(sys.argv[3])                            # tile factorisations per step (c2: 65, c3: 136)
last = pt[step_idx * per_step:(step_idx + 1) * per_step]
ntiles = len(last)
# step boundaries: the widest kernel-free gap between the previous step's last tile factorisation and this step's first one
# (the host reads the results back there), and likewise towards the next step
def widest_gap(lo_t, hi_t):
    ks = [r for r in rows if lo_t <= r['s'] <= hi_t]
    best, cut, end = -1, None, lo_t
    for r in ks:
        if r['s'] - end > best and end > lo_t: best, cut = r['s'] - end, r['s']
        end = max(end, r['e'])
    return cut
prev_last = pt[step_idx * per_step - 1]['e'] if step_idx > 0 else rows[0]['s']
nxt_first = pt[(step_idx + 1) * per_step]['s'] if (step_idx + 1) * per_step < len(pt) else rows[-1]['e']
t0 = widest_gap(prev_last, last[0]['s']) or rows[0]['s']
tn = widest_gap(last[-1]['e'], nxt_first) or rows[-1]['e'] + 1
i0 = next(i for i, r in enumerate(rows) if r['s'] >= t0)
i1 = max(i for i, r in enumerate(rows) if r['s'] < tn)
t1 = max(r['e'] for r in rows[i0:i1 + 1])
sel = rows[i0:i1 + 1]
print(f"step: {len(sel)} kernels, {(t1 - t0) / 1e6:.3f} ms; potrf window {(last[-1]['e'] - last[0]['s']) / 1e6:.3f} ms "
      f"(starts at {(last[0]['s'] - t0) / 1e3:.0f} us), after potrf {(t1 - last[-1]['e']) / 1e6:.3f} ms")
tot = defaultdict(lambda: [0, 0.0])
for r in sel:
    k = short(r['Kernel_Name']); tot[k][0] += 1; tot[k][1] += (r['e'] - r['s']) / 1e3
for k, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {us:9.1f} us  {n:5d} x {us / n:7.1f}  {k}")
# idle time (no kernel running)
ev = sorted([(r['s'], 1) for r in sel] + [(r['e'], -1) for r in sel])
run = 0; idle = 0; lastt = t0
for t, dlt in ev:
    if run == 0: idle += t - lastt
    run += dlt; lastt = t
print(f"time with no kernel running: {idle / 1e3:.1f} us")
per = [(last[i + 1]['s'] - last[i]['s']) / 1e3 for i in range(ntiles - 1)]
print("tile-to-tile period (us), by panel of 4:")
for p in range(0, ntiles - 1, 4):
    print(f"  tiles {p:3d}-{p + 3:3d}: " + " ".join(f"{x:7.1f}" for x in per[p:p + 4]))
if len(sys.argv) > 5:
    a, b = float(sys.argv[4]) * 1e3 + t0, float(sys.argv[5]) * 1e3 + t0
    for r in sel:
        if a <= r['s'] < b:
            print(f"{(r['s'] - t0) / 1e3:9.1f} us  dur {(r['e'] - r['s']) / 1e3:7.1f}  wgs {r['b']:6d}  q {r.get('Queue_Id', '?'):>3}  {short(r['Kernel_Name'])}")
