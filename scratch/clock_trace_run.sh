#!/bin/bash
# clock trace beside (1) bench.py c3 steps, (2) the sustained SYRK loop
cd "$GRAFT_REPO_ROOT"; D=gpurun_out/r4w; mkdir -p $D
python3 bench.py --steps 60 --warmup 3 --no-cpu > $D/bench.json 2> $D/bench.err &
BP=$!
sleep 9          # imports, uploads, warm-up, into the timed steps
./scratch/clock_trace 1.0 2 > $D/clock_bench.txt
wait $BP
python3 - <<'PY'
import numpy as np, json
a = np.loadtxt("gpurun_out/r4w/clock_bench.txt")
print("clock beside bench.py steps: samples", len(a), "MHz min/median/max", a[:,1].min(), np.median(a[:,1]), a[:,1].max())
# histogram
h, e = np.histogram(a[:,1], bins=[0,1500,1700,1800,1900,2000,2100,2200,2300,2450])
print("histogram", dict(zip([f"<{int(x)}" for x in e[1:]], h.tolist())))
d = json.loads(open("gpurun_out/r4w/bench.json").read().strip().splitlines()[-1]); print("bench", d["ms_per_step"])
# print a 120-ms window at 1-ms resolution
w = a[(a[:,0] > 200) & (a[:,0] < 320)]
for t in range(200, 320, 2):
    s = w[(w[:,0] >= t) & (w[:,0] < t + 2)]
    if len(s): print(f"  t={t:4d} ms  {s[:,1].mean():6.0f} MHz")
PY
