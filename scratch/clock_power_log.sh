#!/bin/bash
# Core clock and socket power while bench.py runs c3 steps back to back (rocm-smi, read-only): the evidence behind "the clock
# under load is the ceiling" (MEASUREMENTS.md).  Samples every 0.25 s for the duration of the bench process.
cd "$GRAFT_REPO_ROOT"
python3 bench.py --steps 150 --warmup 3 --no-cpu > gpurun_out/clock_bench.json 2> gpurun_out/clock_bench.err &
BP=$!
echo "# t_s  sclk  power  (rocm-smi --showclocks --showpower, 0.25 s apart; bench.py --steps 150 running)"
T0=$(date +%s.%N)
while kill -0 $BP 2>/dev/null; do
  OUT=$(rocm-smi --showclocks --showpower 2>/dev/null)
  S=$(echo "$OUT" | grep -i "sclk" | head -1 | sed 's/.*(\([0-9]*Mhz\)).*/\1/I')
  P=$(echo "$OUT" | grep -i -E "power \(W\)|Socket Power|Average Graphics Package Power" | head -1 | sed 's/.*: *//')
  NOW=$(date +%s.%N)
  echo "$(echo "$NOW - $T0" | bc | cut -c1-6)  $S  $P"
  sleep 0.25
done
wait $BP
tail -n 1 gpurun_out/clock_bench.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('# bench:', round(d['ms_per_step'],2), 'ms per step,', round(d['roofline']['achieved'],1), 'TFLOP/s in the update')"
rocm-smi --showclocks --showpower 2>/dev/null | grep -i -E "sclk|power" | head -4 | sed 's/^/# idle afterwards: /'
