#!/bin/bash
# One GPU-box visit of round 4 (run from the repo root).  usage: bash tools/gpu_visit4.sh <tag> [steps...]
# The caller writes the commit into .git_sha first (the box has no .git): git rev-parse --short=12 HEAD > .git_sha
set -u
TAG=${1:-r04_v1}; shift
STEPS=${@:-timed bench}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
export TC_GIT_SHA=$(cat $R/.git_sha 2>/dev/null || echo unknown)
summ() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print($1)"; }
for S in $STEPS; do
  case $S in
    tests) timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 > $O/pytest_gpu.log 2>&1; echo "tests rc=$?"; tail -5 $O/pytest_gpu.log ;;
    timed) timeout 900 python -m pytest tests/test_gpu_timed_config.py -m gpu -x -q --timeout 300 --durations=5 > $O/pytest_timed.log 2>&1; echo "timed-config tests rc=$?"; tail -12 $O/pytest_timed.log ;;
    quick) timeout 600 python -m pytest tests/test_gpu_prefill.py tests/test_gpu_fixed.py tests/test_gpu_slots.py tests/test_gpu_bucket.py -m gpu -x -q > $O/pytest_quick.log 2>&1; echo "quick rc=$?"; tail -6 $O/pytest_quick.log ;;
    bench) TC_BENCH_VERBOSE=1 timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_stdout.txt 2> $O/bench_stderr.txt; echo "bench rc=$?"
           tail -c 4200 $O/bench_stdout.txt; echo; wc -c $O/bench_stdout.txt; cp gpurun_out/bench_detail.json $O/ 2>/dev/null; grep -i "verified\|failed" $O/bench_stderr.txt | head -20 ;;
    bench200) for WL in uniform zipf; do timeout 300 python bench.py --steps 200 --warmup 10 --no-also --no-cpu --no-verify --workload $WL 2>/dev/null | tail -1 > $O/bench200_$WL.json; summ "d['config']['stream'], round(d['ms_per_step']*1e3,1), 'us/step', round(d['value']/1e9,2), 'G/s'" < $O/bench200_$WL.json; done ;;
    perkey) bash tools/gpu_round.sh $TAG uniform_fixed_tiers1000 zipf_fixed_tiers1000 ;;
    perkey4) bash tools/gpu_round.sh $TAG uniform_fixed_tiers4 ;;
    prof) bash tools/gpu_round.sh $TAG uniform_fixed zipf_fixed ;;
    profgen) bash tools/gpu_round.sh $TAG general_uniform_fixed general_zipf_fixed ;;
    ab) timeout 600 python tools/ab_step.py 100 ${AB_CONFIGS:-default} > $O/ab_step.txt 2>&1; echo "ab rc=$?"; cat $O/ab_step.txt ;;
    timeline) cd /tmp; export TMPDIR=/tmp
           for WL in uniform zipf; do
             timeout 240 rocprofv3 --kernel-trace -d $O/tl_$WL -o t -- python $R/bench.py --profile-run --steps 40 --warmup 5 --layout fixed --workload $WL > $O/tl_$WL.log 2>&1
             python $R/tools/timeline.py $O/tl_$WL > $O/timeline_$WL.txt 2>&1; tail -30 $O/timeline_$WL.txt; rm -rf $O/tl_$WL
           done; cd $R ;;
    *) if [ -f "$R/tools/visit_steps/$S.sh" ]; then bash $R/tools/visit_steps/$S.sh $TAG; else echo "unknown step $S"; fi ;;
  esac
done
