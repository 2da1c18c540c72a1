#!/bin/bash
# One GPU-box visit: rocprofv3 kernel stats + PMC passes of the commands bench.py times, summarised into profiles/.
# usage (from the repo root on the GPU box): bash tools/gpu_round.sh <tag> [configs...]   e.g.  r02_v3 uniform_fixed zipf_fixed keys_short
# Each rocprofv3 run is wrapped in its own timeout (a profiler that does not come back must not eat the visit).
set -u
TAG=${1:-r02}
shift
CONFIGS=${@:-uniform_fixed uniform_wide zipf_fixed string_keys string_keys_long}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TC_GIT_SHA=${TC_GIT_SHA:-$(cat $R/.git_sha 2>/dev/null || echo unknown)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
for C in $CONFIGS; do
  PMCENV=""
  case $C in
    # PMC passes serialise the dispatches: the engine's probe then finds no concurrent grouping stream and would run
    # its batches in order, on other kernel variants (4 items per lane, no preset decision bytes, bucket path) than
    # the timed, pipelined run.  TCGPU_ASSUME_CONCURRENT=1 skips the probe, so the counter passes see exactly the
    # kernels of the timed configuration (ordering is by events either way).  The *_bucket configs profile the
    # in-order bucket path instead.
    uniform_fixed) CMD="python $R/bench.py --profile-run --steps 20 --warmup 3 --layout fixed"; PMCENV="TCGPU_ASSUME_CONCURRENT=1" ;;
    uniform_fixed_bucket) CMD="python $R/bench.py --profile-run --steps 20 --warmup 3 --layout fixed --in-order" ;;
    uniform_wide) CMD="python $R/bench.py --profile-run --steps 20 --warmup 3 --layout wide"; PMCENV="TCGPU_ASSUME_CONCURRENT=1" ;;
    uniform_fixed_tiers4) CMD="python $R/bench.py --profile-run --steps 20 --warmup 3 --layout fixed --plans tiers4"; PMCENV="TCGPU_ASSUME_CONCURRENT=1" ;;
    uniform_fixed_tiers1000) CMD="python $R/bench.py --profile-run --steps 20 --warmup 3 --layout fixed --plans tiers1000"; PMCENV="TCGPU_ASSUME_CONCURRENT=1" ;;
    zipf_fixed_tiers1000) CMD="python $R/bench.py --profile-run --steps 20 --warmup 3 --layout fixed --workload zipf --plans tiers1000"; PMCENV="TCGPU_ASSUME_CONCURRENT=1" ;;
    zipf_fixed) CMD="python $R/bench.py --profile-run --steps 20 --warmup 3 --layout fixed --workload zipf"; PMCENV="TCGPU_ASSUME_CONCURRENT=1" ;;
    zipf_wide) CMD="python $R/bench.py --profile-run --steps 20 --warmup 3 --layout wide --workload zipf"; PMCENV="TCGPU_ASSUME_CONCURRENT=1" ;;
    general_uniform_fixed) CMD="python $R/bench.py --profile-run --steps 20 --warmup 3 --layout fixed --workload general"; PMCENV="TCGPU_ASSUME_CONCURRENT=1" ;;
    general_zipf_fixed) CMD="python $R/bench.py --profile-run --steps 20 --warmup 3 --layout fixed --workload general_zipf"; PMCENV="TCGPU_ASSUME_CONCURRENT=1" ;;
    string_keys) CMD="python $R/tools/profile_keys.py short 8"; PMCENV="TCGPU_ASSUME_CONCURRENT=1" ;;
    string_keys_long) CMD="python $R/tools/profile_keys.py long 8"; PMCENV="TCGPU_ASSUME_CONCURRENT=1" ;;
    *) echo "unknown config $C"; continue ;;
  esac
  cd /tmp
  timeout 240 rocprofv3 --kernel-trace --stats -d $O/${C}_stats -o s -- $CMD > $O/${C}_stats.log 2>&1; echo "$C stats rc=$?"
  env $PMCENV timeout 240 rocprofv3 --pmc FETCH_SIZE -d $O/${C}_fetch -o f -- $CMD > $O/${C}_fetch.log 2>&1; echo "$C fetch rc=$?"
  env $PMCENV timeout 240 rocprofv3 --pmc WRITE_SIZE -d $O/${C}_write -o w -- $CMD > $O/${C}_write.log 2>&1; echo "$C write rc=$?"
  cd $R
  python tools/summarize_prof.py ${TAG}_${C} $O/${C}_stats $O/${C}_fetch $O/${C}_write "$CMD  [PMC passes: $PMCENV, streams serialised by the profiler]" > $O/${C}_summary.log 2>&1; tail -1 $O/${C}_summary.log
  mkdir -p $O/profiles; cp profiles/${TAG}_${C}* $O/profiles/ 2>/dev/null
  rm -rf $O/${C}_stats $O/${C}_fetch $O/${C}_write   # (the rocpd databases are tens of MB each: gpurun brings back 64 MiB at most)
done
