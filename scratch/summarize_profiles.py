"""gpurun_out/<round>_<tag>/ (rocprofv3 output of scratch/collect_profiles.sh) -> small per-kernel tables written to
gpurun_out/<round>_<tag>/profiles/ (copied by hand into profiles/, which is tracked).  Runs on the GPU box right after
the collection (the raw per-dispatch CSVs are too large to travel) or locally on the merged directory."""
import collections, csv, glob, json, os, sys

rnd, tag = sys.argv[1], sys.argv[2]
D = f"gpurun_out/{rnd}_{tag}"
OUT = f"{D}/profiles"
os.makedirs(OUT, exist_ok=True)
SYRK_CANDIDATES = ("gemm_f64_kernel<false, false, 1>", "gemm3_f64_kernel<false, 1>")   # rank-nb trailing update, remainder half: two- / three-resident kernel
SYRK = SYRK_CANDIDATES[0]
commands = open(f"{D}/commands.txt").read().strip().split("\n")


def newest(pattern):
    files = glob.glob(pattern, recursive=True)
    return max(files, key=os.path.getmtime) if files else None


summary = {"round": rnd, "config": tag, "commands": commands}
commands = [c for c in commands if not c.startswith("#")]        # (notes about the environment stay in summary["commands"])
st = newest(f"{D}/stats/**/*kernel_stats.csv")
if st:
    rows = list(csv.DictReader(open(st)))
    with open(f"{OUT}/{rnd}_bench_{tag}_kernel_stats.csv", "w") as f:
        f.write("# " + commands[0] + "\n")
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in rows[:24]:
            w.writerow(r)
    summary["stats_command"] = commands[0]
    summary["kernels"] = [{"name": r["Name"][:110], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                           "total_ms": float(r["TotalDurationNs"]) / 1e6, "pct": float(r["Percentage"])} for r in rows[:14]]
    cand = [r for r in rows if any(c in r["Name"] for c in SYRK_CANDIDATES)]
    if cand:                       # the one the run spent more time in is the roofline kernel of that run
        best = max(cand, key=lambda r: float(r["TotalDurationNs"]))
        SYRK = next(c for c in SYRK_CANDIDATES if c in best["Name"])
    syrk = [r for r in rows if SYRK in r["Name"]]
    if syrk:
        summary.update(syrk_kernel=syrk[0]["Name"], syrk_calls=int(syrk[0]["Calls"]), syrk_avg_us=float(syrk[0]["AverageNs"]) / 1e3,
                       syrk_total_ms=float(syrk[0]["TotalDurationNs"]) / 1e6)
per_kernel = {}
for i, name in enumerate(("fetch", "write")):
    f = newest(f"{D}/pmc_{name}/**/*counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][0] += 1
        agg[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    with open(f"{OUT}/{rnd}_bench_{tag}_pmc_{name}_size.csv", "w") as out:
        out.write("# " + commands[1 + i] + "\n")
        out.write(f"Kernel_Name,Dispatches,{name.upper()}_SIZE_KB_total\n")
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
            out.write(f"\"{k}\",{n},{v}\n")
            per_kernel.setdefault(k, {})[name] = (n, v)
    summary[f"pmc_{name}_command"] = commands[1 + i]
# HBM bytes per launch of the hot kernels: (2 * FETCH_SIZE + WRITE_SIZE) * 1024 B (MI355X_MICROARCH.md section HBM: on gfx950
# FETCH_SIZE reports half of a wide coalesced stream)
traffic = {}
for k, d in per_kernel.items():
    if "fetch" in d and "write" in d and d["fetch"][0] > 0:
        traffic[k[:110]] = {"dispatches": d["fetch"][0], "hbm_bytes_per_launch": (2 * d["fetch"][1] + d["write"][1]) * 1024 / d["fetch"][0],
                            "write_bytes_per_launch": d["write"][1] * 1024 / d["write"][0]}
summary["traffic_formula"] = "(2*FETCH_SIZE + WRITE_SIZE) * 1024 B per dispatch, MI355X_MICROARCH.md section HBM"
summary["traffic_per_kernel"] = dict(sorted(traffic.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["dispatches"])[:10])
for k, v in traffic.items():
    if SYRK in k:
        summary["syrk_hbm_bytes_per_launch"] = v["hbm_bytes_per_launch"]
f = newest(f"{D}/pmc_mfma/**/*counter_collection.csv")
if f:
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[r["Kernel_Name"]] += 1
    busy = {}
    for k, d in agg.items():
        if d.get("GRBM_GUI_ACTIVE", 0) > 0 and d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs (/ 8 = kernel duration in core cycles), the SQ counter over the 1024 SIMDs
            busy[k[:110]] = {"dispatches": cnt[k] // 2, "mfma_busy_frac": d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8 * 1024)}
    summary["pmc_mfma_command"] = commands[3]
    summary["mfma_busy_note"] = ("SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs): GRBM_GUI_ACTIVE is summed over the 8 XCDs, the SQ counter over "
                                 "the SIMDs; counter collection serialises the kernels (each runs alone)")
    summary["mfma_busy"] = dict(sorted(busy.items(), key=lambda kv: -kv[1]["mfma_busy_frac"])[:8])
try:
    summary["bench_command"] = commands[-1]
    summary["bench_line"] = json.load(open(f"{D}/bench.json"))
except Exception as exc:  # noqa: BLE001
    summary["bench_line_error"] = str(exc)
json.dump(summary, open(f"{OUT}/{rnd}_bench_{tag}_summary.json", "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k not in ("bench_line", "kernels", "traffic_per_kernel")}, indent=1)[:1500])
