import csv, glob, os, collections, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "kron" in n or "assemble" in n:
        agg[(n[:60], r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:6]:
    print(k, len(v), "avg us", round(sum(v) / len(v), 1))
