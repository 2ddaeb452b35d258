"""Chain kernels beside the remainder update, from a rocprofv3 kernel trace of bench.py: for the tile steps that run WHILE a
rank-512 update (gemm_f64_kernel<false,false,1> on another queue) is in flight -- duration of each chain kernel and the gap
in front of it on the panel queue -- against the steps that run with no update in flight.
usage: python scratch/chain_under_update.py <kernel_trace.csv>"""
import csv, sys, statistics as st
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
upd = [(r["s"], r["e"]) for r in rows if "gemm_f64_kernel<false, false, 1>" in r["Kernel_Name"]]
chainq = max(set(r["Queue_Id"] for r in rows if "tile_solve" in r["Kernel_Name"]), key=lambda q: sum(1 for r in rows if r["Queue_Id"] == q and "tile_solve" in r["Kernel_Name"]))
chain = [r for r in rows if r["Queue_Id"] == chainq]
def under(r):
    return any(s < r["s"] and r["e"] < e for s, e in upd)
acc = {}
for prev, r in zip(chain, chain[1:]):
    name = r["Kernel_Name"].split("(")[0].replace("void lpgp::", "").replace("lpgp::", "")
    if not any(k in name for k in ("potrf_tile", "tile_solve", "gemm64_f64_kernel<false, false, 2>", "streamOps", "gemm_f64_kernel<false, false, 3>", "gemm64_f64_kernel<false, false, 1>")):
        continue
    key = (name[:40], under(r))
    acc.setdefault(key, []).append(((r["e"] - r["s"]) / 1e3, (r["s"] - prev["e"]) / 1e3))
for (name, u), v in sorted(acc.items()):
    d = [x[0] for x in v]; g = [x[1] for x in v]
    print(f"{name:42s} {'beside an update' if u else 'alone           '} n={len(v):4d}  duration median {st.median(d):7.1f} p90 {sorted(d)[int(0.9*len(d))]:7.1f} us   gap before: median {st.median(g):6.1f} p90 {sorted(g)[int(0.9*len(g))]:7.1f} us")

d = [(e - s_) / 1e3 for s_, e in upd]
print(f"rank-512 updates: n={len(d)} total {sum(d)/1e3:.1f} ms, median {st.median(d):.0f} us")
srv = [r for r in rows if "potrf_server" in r["Kernel_Name"]]
if srv:
    print(f"potrf_server_kernel: n={len(srv)}, median residency {st.median([(r['e'] - r['s']) / 1e3 for r in srv]):.0f} us")
