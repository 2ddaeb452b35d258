#!/bin/bash
# kernel trace of c2 steps with the two-kernel chain: per panel the chain kernel (A) and the rows kernel (B)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
D=gpurun_out/r6_chain_trace; rm -rf $D; mkdir -p $D
export LPGP_BENCH_NO_MODES=1 LPGP_BENCH_PROF_STEPS=1
LPGP_CHAIN_RESIDENT2=${R2:-0} LPGP_CHAIN_AHEAD=${AH:-1} rocprofv3 --kernel-trace --output-format csv -d $D -- python3 bench.py --workload ${WL:-poisson1d} --steps 3 --warmup 2 --no-cpu > $D/log.txt 2>&1
python3 - <<'PY'
import csv, glob
f = max(glob.glob("gpurun_out/r6_chain_trace/**/*kernel_trace.csv", recursive=True))
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# the last complete step: find the last kron/assemble? simply print the chain kernels of the last ~40 launches of panel_chain_kernel
ch = [r for r in rows if "panel_chain" in r["Kernel_Name"] or "potrf_tile" in r["Kernel_Name"]]
out = []
for r in ch[-60:]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    out.append(f"{r['Kernel_Name'][:40]:40s} q{r.get('Queue_Id','?'):>3s} wgs {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):5d} start {s:12.1f} us  dur {e - s:8.1f}")
open("gpurun_out/r6_chain_trace.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[-40:]))
PY
