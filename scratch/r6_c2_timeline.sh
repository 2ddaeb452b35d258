#!/bin/bash
# kernel trace of ONE c2 step in the product's schedule: every kernel of the last step with queue, start, duration
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
D=gpurun_out/r6_${TAG:-c2}_timeline; rm -rf $D; mkdir -p $D
export LPGP_BENCH_NO_MODES=1 LPGP_BENCH_PROF_STEPS=1
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 bench.py --workload ${WL:-poisson1d} --steps 3 --warmup 2 --no-cpu > $D/log.txt 2>&1
python3 - "${TAG:-c2}" <<'PY'
import csv, glob, sys
TAG = sys.argv[1]
f = max(glob.glob(f"gpurun_out/r6_{TAG}_timeline/**/*kernel_trace.csv", recursive=True))
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps start with the Gram assembly: find starts of 'assemble' bursts; take the last timed step = the last-but-(prof passes) ... simply: split at assemble_fast launches separated by > 2 ms
idx = [i for i, r in enumerate(rows) if "assemble" in r["Kernel_Name"]]
starts = [idx[0]]
for a, b in zip(idx, idx[1:]):
    if int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]) > 2_000_000: starts.append(b)
print("steps found", len(starts))
# choose the 4th step (after 2 warm-ups: steps 3..5 are timed)
s0 = starts[min(3, len(starts) - 2)]; s1 = starts[min(4, len(starts) - 1)]
t0 = int(rows[s0]["Start_Timestamp"])
out = []
short = lambda n: n.replace("void lpgp::", "").replace("lpgp::", "").split("(")[0][:44]
for r in rows[s0:s1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    out.append(f"{short(r['Kernel_Name']):44s} q{r.get('Queue_Id','?'):>3s} wgs {int(r['Grid_Size_X'])//max(1,int(r['Workgroup_Size_X'])):6d} start {s:9.1f} end {e:9.1f} dur {e - s:8.1f}")
open(f"gpurun_out/r6_{TAG}_timeline.txt", "w").write("\n".join(out) + "\n")
print(len(out), "kernels in the step;", "step length", out[-1].split("end")[1].split()[0], "us")
PY
