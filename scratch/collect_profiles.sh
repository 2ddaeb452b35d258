#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel stats + PMC passes (separate runs, no other trace domains) of ONE
# bench command per configuration, then the bench line itself.  Every command that ran is written, verbatim, into
# gpurun_out/<round>_<tag>/commands.txt, which scratch/summarize_profiles.py copies into the committed summary.
#   usage: collect_profiles.sh <round, e.g. r02> <tag> <bench args ...>
set -u
ROUND=$1; TAG=$2; shift 2
ARGS="$*"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export LPGP_BENCH_NO_MODES=1      # the profiled runs contain the timed mode only (the default-mode / reference-sequence passes of the bench line would mix two schedules into one table)
export LPGP_BENCH_PROF_STEPS=3      # bench.py's per-kernel HIP-event passes: short under the profiler (as in the committed r03 collections)
D=gpurun_out/${ROUND}_${TAG}
rm -rf "$D"; mkdir -p "$D"
run() {   # run <subdir> <rocprofv3 options ...> -- <program ...>; logs the command line exactly as executed
  local sub=$1; shift
  echo "$*" >> "$D/commands.txt"
  "$@" > "$D/$sub.log" 2>&1
}
STATS_ARGS="--steps 5 --warmup 2 --no-cpu $ARGS"
PMC_ARGS="--steps 1 --warmup 0 --no-cpu $ARGS"
run stats rocprofv3 --kernel-trace --stats --output-format csv -d "$D/stats" -- python3 bench.py $STATS_ARGS
# counter collection SERIALISES kernel dispatches: a kernel that follows another kernel's device flags (panel_chain_v_kernel, the
# substitution's panel step behind a resident chain) must not be picked before it -- off for the counter passes (it is not the
# roofline kernel; the stats pass and the bench line run with it)
export LPGP_RIDE_VCHAIN=0
echo "# the three --pmc passes run with LPGP_RIDE_VCHAIN=0 in the environment (serialised dispatches)" >> "$D/commands.txt"
run pmc_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$D/pmc_fetch" -- python3 bench.py $PMC_ARGS
run pmc_write rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$D/pmc_write" -- python3 bench.py $PMC_ARGS
run pmc_mfma rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$D/pmc_mfma" -- python3 bench.py $PMC_ARGS
unset LPGP_RIDE_VCHAIN
echo "python3 bench.py --steps 5 --warmup 2 --no-cpu $ARGS" >> "$D/commands.txt"
unset LPGP_BENCH_NO_MODES
python3 bench.py --steps 5 --warmup 2 --no-cpu $ARGS 2> "$D/bench.err" | tail -1 > "$D/bench.json"
python3 scratch/summarize_profiles.py "$ROUND" "$TAG"
