"""Turn gpurun_out/r01_* (rocprofv3 output) into the committed files under profiles/."""
import collections, csv, glob, json, os, shutil
os.makedirs('profiles', exist_ok=True)
SYRK = 'gemm_f64_kernel<false, false, 1>'
st = max(glob.glob('gpurun_out/r01_stats/*/*kernel_stats.csv'), key=os.path.getmtime)
shutil.copy(st, 'profiles/r01_bench_c3_kernel_stats.csv')
rows = list(csv.DictReader(open(st)))
syrk_row = [r for r in rows if SYRK in r['Name']][0]
summary = {
    "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu",
    "syrk_kernel": syrk_row['Name'],
    "syrk_calls": int(syrk_row['Calls']),
    "syrk_avg_us": float(syrk_row['AverageNs']) / 1e3,
    "syrk_total_ms": float(syrk_row['TotalDurationNs']) / 1e6,
}
for name in ("fetch", "write"):
    f = max(glob.glob(f'gpurun_out/r01_pmc_{name}/*/*counter_collection.csv'), key=os.path.getmtime)
    rr = [r for r in csv.DictReader(open(f)) if SYRK in r['Kernel_Name']]
    # keep the committed file small: per-kernel totals instead of one row per dispatch
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][0] += 1
        agg[r['Kernel_Name']][1] += float(r['Counter_Value'])
    with open(f'profiles/r01_bench_c3_pmc_{name}_size.csv', 'w') as out:
        out.write(f"Kernel_Name,Dispatches,{name.upper()}_SIZE_KB_total\n")
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            out.write(f"\"{k}\",{n},{v}\n")
    summary[f"syrk_{name}_size_kb_total"] = sum(float(r['Counter_Value']) for r in rr)
    summary[f"syrk_{name}_dispatches"] = len(rr)
fetch_b = summary["syrk_fetch_size_kb_total"] * 1024 * 2   # gfx950: FETCH_SIZE reports 1/2 of a wide coalesced stream
write_b = summary["syrk_write_size_kb_total"] * 1024
summary["syrk_hbm_bytes_per_launch"] = (fetch_b + write_b) / summary["syrk_fetch_dispatches"]
summary["pmc_command"] = "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --no-cpu (separate passes)"
summary["traffic_formula"] = "(2*FETCH_SIZE + WRITE_SIZE) * 1024 B per dispatch, MI355X_MICROARCH.md section HBM"
b = json.load(open('gpurun_out/bench_r01_c3.json'))
summary["bench_line"] = b
json.dump(summary, open('profiles/r01_bench_c3_summary.json', 'w'), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != 'bench_line'}, indent=1))
print("bench:", b["value"], b["ms_per_step"], b["roofline"])
