"""Tile-step period (start of one potrf_tile_kernel to the next) and kernel durations along ONE factorisation, from a rocprofv3
--kernel-trace CSV (scratch/collect_profiles.sh leaves it under gpurun_out/<round>_<tag>/stats/).  usage: tile_step_periods.py <trace.csv> <tiles per step>"""
import collections, csv, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
T = int(sys.argv[2])
l = sorted([(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']) for r in rows])
short = lambda n: n.split('<')[0].split('::')[-1].split('(')[0]
pot = [i for i, x in enumerate(l) if 'potrf_tile' in x[2]]
# the last complete run of T consecutive tile Choleskys whose gaps stay below 3 ms
runs, cur = [], [pot[0]]
for a, b in zip(pot, pot[1:]):
    if l[b][0] - l[a][0] > 3e6:
        runs.append(cur); cur = []
    cur.append(b)
runs.append(cur)
st = [r for r in runs if len(r) >= T][-2][-T:]
per = [(l[st[k + 1]][0] - l[st[k]][0]) / 1e3 for k in range(len(st) - 1)]
print(f"# {sys.argv[1]}: {len(st)} tile steps; period in us (profiler attached: ~15 % slower than a plain run)")
for a in range(0, len(per), 8):
    print(f"tile {a:3d}:", " ".join(f"{x:5.0f}" for x in per[a:a + 8]))
gaps = [(l[i + 1][0] - l[i][1]) / 1e3 for i in range(st[0], st[-1]) if l[i][3] == l[i + 1][3] == l[st[0]][3]]
print(f"gap between consecutive kernels of the panel queue: median {statistics.median(gaps):.2f} us, p90 {sorted(gaps)[9 * len(gaps) // 10]:.2f} us")
d = collections.defaultdict(list)
for s, e, n, q in l[st[0]:st[-1] + 1]:
    d[short(n) + ' @queue ' + q].append((e - s) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:10]:
    print(f"{k:42s} {len(v):4d} launches, median {statistics.median(v):7.1f} us, sum {sum(v) / 1e3:6.2f} ms")
print(f"wall time of the run of tile steps: {(l[st[-1]][1] - l[st[0]][0]) / 1e6:.2f} ms")
