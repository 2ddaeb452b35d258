import csv, glob, os, sys, collections
d = sys.argv[1]
f = sorted(glob.glob(f'{d}/*/*kernel_trace.csv'), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    n = r['Kernel_Name'][:90]; agg[n][0] += 1; agg[n][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{t/1e6:10.2f} ms {c:6d}  {n}")
